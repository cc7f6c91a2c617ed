"""FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh -> profiles/<name>_hbm_counters.json (+ the bench lines and rocprof summaries):
python tools/pmc_to_json.py gpurun_out/r01_j r01_j "state description" """
import csv, json, collections, shutil, sys, os
O, name, what = sys.argv[1].rstrip("/") + "/", sys.argv[2], sys.argv[3]
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
def load(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(O + d + "/pmc_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f, w = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py (4 frames forward+backward, cfg3: 500k Gaussians, "
               "1080p, SH3, fused accumulation through the one-view kernels, one stream; " + what + "); KB per launch as reported. FETCH_SIZE under-reports 16-B-per-lane "
               "streaming reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): hbm_bytes_est = 2*FETCH + WRITE.", "kernels": {}}
for k in sorted(set(f) | set(w)):
    if "tgs" in k:
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(f.get(k, 0.0), 1), "WRITE_SIZE_KB_per_launch": round(w.get(k, 0.0), 1),
                             "hbm_bytes_est": int((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)}
json.dump(out, open(f"{P}/{name}_hbm_counters.json", "w"), indent=1)
for src, dst in (("bench_default.json", "bench_default.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                 ("bench_under_rocprof_one_stream.json", "bench_under_rocprof_one_stream.json"), ("rp4/rp_kernel_stats.csv", "kernel_stats.csv"),
                 ("rp1/rp_kernel_stats.csv", "kernel_stats_one_stream.csv")):
    shutil.copy(O + src, f"{P}/{name}_{dst}")
for k, v in out["kernels"].items():
    print(k, v)
