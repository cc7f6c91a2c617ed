"""Counter passes of tools/profile_round.sh -> profiles/<name>_hbm_counters.json, profiles/<name>_sq_counters.json (+ the bench lines, rocprof
summaries and the VALU microbenchmark):   python tools/pmc_to_json.py gpurun_out/r02_a r02_a "state description" """
import csv, json, collections, glob, shutil, sys, os
O, name, what = sys.argv[1].rstrip("/") + "/", sys.argv[2], sys.argv[3]
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def rows(d):
    f = glob.glob(O + d + "/**/*counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def kname(r):
    return r["Kernel_Name"].replace("void ", "").split("(")[0]


def load(d, counter):
    agg = collections.defaultdict(list)
    for r in rows(d):
        if r["Counter_Name"] == counter:
            agg[kname(r)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd.build import source_hash
# bench.py quotes these counters only while the kernel sources still hash to this: the hash recorded ON THE BOX when the passes ran
# (tools/profile_round.sh: csrc_sha16.txt), or -- for older sets without the file -- of the tree this script runs in
SRC = open(O + "csrc_sha16.txt").read().strip() if os.path.exists(O + "csrc_sha16.txt") else source_hash()

# Per-kernel FETCH_SIZE calibration (round 3; round 2 doubled every kernel's FETCH): the dominant READ pattern of each kernel, and the factor
# tools/microbench/fetch_calibration.hip measured for that pattern on this chip (profiles/<name>_fetch_calibration.json when the round has
# one, else the guide's 2.0 for 16-B-per-lane streams and 1.0 for everything else -- stated per kernel in the output).
READ_PATTERN = {
    "k_preprocess_fwd": "k_stream16", "k_preprocess_fwd_pair": "k_stream16", "k_preprocess_fwd_batch": "k_stream16",   # SH rows, 16 B per lane
    "k_preprocess_bwd": "k_stream16", "k_preprocess_bwd_batch": "k_stream16", "k_preprocess_bwd_batch_split": "k_stream16",   # SH rows (streams, staged through LDS) outweigh the slab rows (gathers): see hbm_bytes_bounds
    "k_render_fwd": "k_stream16", "k_render_bwd": "k_stream16", "k_render_bwd_det": "k_stream16",                    # records, 16 B per lane
    "k_finalize": "k_gather64", "k_tile_sort": "k_stream8", "k_scatter": "k_stream4", "k_bin_count": "k_stream4", "k_bin_colscan": "k_stream4",
    "k_scan": "k_stream4", "k_fill_empty": "k_stream4", "k_ssim_stats": "k_stream4", "k_ssim_grad": "k_stream4", "k_sh_rgb": "k_stream16",
}
cal_path = f"{P}/{name}_fetch_calibration.json"
cal = json.load(open(cal_path))["patterns"] if os.path.exists(cal_path) else None
DEFAULT_FACTOR = {"k_stream16": 2.0}


def fetch_factor(kernel):
    """(pattern, bytes moved per reported byte, source).  The calibration divides KNOWN USEFUL bytes by the counter: 2.0 for coalesced streams of
    4, 8 and 16 B per lane (the counter tallies 128-B requests at 64 B), 1.01 for random 64-B lines (the counter is exact), 0.63 for three 16-B
    pieces of random 48-B rows -- there the counter EXCEEDS the useful bytes because whole lines move: a ratio below 1 means the counter
    already is the traffic, so the traffic factor is max(ratio, 1)."""
    base = kernel.split("::")[-1].split("<")[0]
    pat = READ_PATTERN.get(base, "k_stream4")
    if cal and pat in cal and "fetch_factor" in cal[pat]:
        return pat, max(1.0, round(float(cal[pat]["fetch_factor"]), 2)), os.path.basename(cal_path)
    return pat, DEFAULT_FACTOR.get(pat, 1.0), "MI355X_MICROARCH.md (2.0 for 16-B-per-lane streams, uncalibrated 1.0 otherwise)"


f, w = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py (4 frames forward+backward, cfg3: 500k Gaussians, "
               "1080p, SH3, fused accumulation through the one-view kernels, one stream; " + what + "); KB per launch as reported.  hbm_bytes_est = fetch_factor * FETCH + WRITE "
               "with the factor of the kernel's dominant read pattern (read_pattern, calibration source stated per kernel).", "csrc_sha16": SRC, "kernels": {}}
for k in sorted(set(f) | set(w)):
    if "tgs" in k:
        pat, fac, src = fetch_factor(k)
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(f.get(k, 0.0), 1), "WRITE_SIZE_KB_per_launch": round(w.get(k, 0.0), 1),
                             "read_pattern": pat, "fetch_factor": fac, "calibration": src,
                             "hbm_bytes_est": int((fac * f.get(k, 0.0) + w.get(k, 0.0)) * 1024),
                             # a kernel that mixes streams and gathers lies between the two: every read counted once / every read counted twice
                             "hbm_bytes_bounds": [int((f.get(k, 0.0) + w.get(k, 0.0)) * 1024), int((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)]}
json.dump(out, open(f"{P}/{name}_hbm_counters.json", "w"), indent=1)

# SQ passes: per kernel, per launch averages + the launch duration seen in the same pass (dispatch timestamps of the counter CSV)
sq = {"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes: pmc_sq, pmc_sq2) over tools/pmc_workload.py (4 frames forward+backward of cfg3 through the "
              "one-view kernels, one stream: every kernel alone on the GPU; " + what + "). Per launch averages. SQ_INSTS_* count wave-instructions; SQ_WAVE_CYCLES / "
              "SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (4 shader cycles) summed over waves (MI355X_MICROARCH.md, cycle constants). "
              "us_in_pass = average dispatch duration inside the counter pass (profiled passes run at a lower clock). valu_rate_G_per_s = SQ_INSTS_VALU / duration.",
      "csrc_sha16": SRC, "kernels": {}}
for d in ("pmc_sq", "pmc_sq2", "pmc_sq3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in rows(d):
        k = kname(r)
        if "tgs" not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, cs in agg.items():
        e = sq["kernels"].setdefault(k, {})
        for c, v in cs.items():
            e[c] = round(sum(v) / len(v), 1)
        if not dur[k]:
            continue
        e["us_in_pass_" + d] = round(sum(dur[k].values()) / len(dur[k]), 2)
        e["launches_" + d] = len(dur[k])
for k, e in sq["kernels"].items():
    if "SQ_INSTS_VALU" in e and e.get("us_in_pass_pmc_sq"):
        e["valu_rate_G_per_s"] = round(e["SQ_INSTS_VALU"] / e["us_in_pass_pmc_sq"] / 1e3, 1)
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac_of_lds_active"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 3)
    if e.get("SQ_THREAD_CYCLES_VALU") and e.get("SQ_ACTIVE_INST_VALU"):
        e["valu_lanes_active_frac"] = round(e["SQ_THREAD_CYCLES_VALU"] / (64.0 * e["SQ_ACTIVE_INST_VALU"]), 3)
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
json.dump(sq, open(f"{P}/{name}_sq_counters.json", "w"), indent=1)

for src, dst in (("bench_default.json", "bench_default.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                 ("bench_under_rocprof_one_stream.json", "bench_under_rocprof_one_stream.json"), ("rp4/rp_kernel_stats.csv", "kernel_stats.csv"),
                 ("rp1/rp_kernel_stats.csv", "kernel_stats_one_stream.csv"), ("valu_issue_rate.txt", "valu_issue_rate.txt"),
                 ("rp_cfg2/rp_kernel_stats.csv", "kernel_stats_cfg2.csv"), ("rp_cfg5/rp_kernel_stats.csv", "kernel_stats_cfg5.csv"),
                 ("rp_trainer/rp_kernel_stats.csv", "kernel_stats_trainer_protocol.csv"), ("bench_cfg2.json", "bench_cfg2.json"), ("bench_cfg5.json", "bench_cfg5.json"),
                 ("trainer_protocol.json", "trainer_protocol_under_rocprof.json"), ("rp_trainer3/rp_kernel_stats.csv", "kernel_stats_trainer_protocol_sh3.csv"),
                 ("trainer_protocol_sh3.json", "trainer_protocol_sh3_under_rocprof.json"), ("rp_dropin/rp_kernel_stats.csv", "kernel_stats_dropin.csv"),
                 ("dropin_under_rocprof.json", "dropin_under_rocprof.json")):
    if os.path.exists(O + src):
        shutil.copy(O + src, f"{P}/{name}_{dst}")
# ---- round 6: large splats (x4 / x8), the store-mode drop-in loop, the batch step -- one JSON with, per workload and kernel, FETCH / WRITE (KB per launch as
# reported + the estimate with the kernel's fetch factor), the SQ / LDS counters where a pass exists, and the kernel's average duration from the same round's kernel stats
def stats_us(path):
    try:
        return {kname({"Kernel_Name": r["Name"]}): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(path))}
    except OSError:
        return {}


extra = {"note": "round 6 (tools/profile_round.sh): the same counter passes over OTHER workloads than tools/pmc_workload.py -- 'x4' / 'x8': one forward + backward frame of config 3 with every "
                 "splat x 4 / x 8 (tools/stage_times.py; most splats on 5..64 tiles); 'dropin': one view per step through GaussianRasterizer (tools/dropin_loop.py: the STORE-mode "
                 "k_preprocess_bwd); 'batch': the 8-view step of bench.py on one stream (k_preprocess_fwd_pair, k_preprocess_bwd_batch_split).  hbm_bytes_est = fetch_factor * FETCH + WRITE; "
                 "tbps = hbm_bytes_est / avg_us of the same workload's kernel stats (a kernel-trace run, not the counter pass).", "csrc_sha16": SRC, "workloads": {}}
for wl, fetch_d, write_d, sq_ds, stats in (("x4", "pmc_fetch_x4", "pmc_write_x4", (), "rp_x4/rp_kernel_stats.csv"), ("x8", "pmc_fetch_x8", "pmc_write_x8", (), "rp_x8/rp_kernel_stats.csv"),
                                           ("dropin", "pmc_fetch_dropin", "pmc_write_dropin", ("pmc_sq_dropin", "pmc_lds_dropin"), "rp_dropin/rp_kernel_stats.csv"),
                                           ("batch", "pmc_fetch_batch", "pmc_write_batch", ("pmc_sq_batch",), "rp1/rp_kernel_stats.csv")):
    ff, ww, us = load(fetch_d, "FETCH_SIZE"), load(write_d, "WRITE_SIZE"), stats_us(O + stats)
    if not ff and not ww:
        continue
    ks = {}
    for k in sorted(set(ff) | set(ww)):
        if "tgs" not in k:
            continue
        pat, fac, _src = fetch_factor(k)
        e = {"FETCH_SIZE_KB": round(ff.get(k, 0.0), 1), "WRITE_SIZE_KB": round(ww.get(k, 0.0), 1), "fetch_factor": fac,
             "hbm_bytes_est": int((fac * ff.get(k, 0.0) + ww.get(k, 0.0)) * 1024)}
        if us.get(k):
            e["avg_us"] = round(us[k], 1)
            e["tbps"] = round(e["hbm_bytes_est"] / us[k] / 1e6, 2)
            e["frac_of_8TBps"] = round(e["tbps"] / 8.0, 3)
        ks[k] = e
    for d in sq_ds:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows(d):
            if "tgs" in kname(r):
                agg[kname(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                ks.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
    for k, e in ks.items():
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac_of_lds_active"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 3)
        if e.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac_of_wave_cycles"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
            if e.get("SQ_WAIT_INST_LDS") is not None:
                e["wait_lds_frac_of_wave_cycles"] = round(e["SQ_WAIT_INST_LDS"] / e["SQ_WAVE_CYCLES"], 3)
    extra["workloads"][wl] = ks
if extra["workloads"]:
    json.dump(extra, open(f"{P}/{name}_workload_counters.json", "w"), indent=1)
for src, dst in (("rp_x4/rp_kernel_stats.csv", "kernel_stats_x4.csv"), ("rp_x8/rp_kernel_stats.csv", "kernel_stats_x8.csv"), ("batch_stage_times.txt", "batch_stage_times.txt")):
    if os.path.exists(O + src):
        shutil.copy(O + src, f"{P}/{name}_{dst}")
for k, v in out["kernels"].items():
    print(k, v)
for k, v in sq["kernels"].items():
    print(k, v)
