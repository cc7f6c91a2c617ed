"""rocprofv3 counter CSVs of tools/microbench/fetch_calibration (FETCH_SIZE pass, WRITE_SIZE pass) + its printed byte counts ->
profiles/<name>_fetch_calibration.json: bytes per reported KB for every access pattern, i.e. the factor tools/pmc_to_json.py applies to a
kernel's FETCH_SIZE / WRITE_SIZE by the pattern of its dominant traffic.
    python tools/fetch_calibration.py gpurun_out/<dir> <name>     (expects <dir>/cal_fetch, <dir>/cal_write, <dir>/cal_bytes.txt)"""
import collections, csv, glob, json, os, sys

O, name = sys.argv[1].rstrip("/") + "/", sys.argv[2]
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
known = {}
for line in open(O + "cal_bytes.txt"):
    w = line.split()
    if len(w) >= 3 and w[0].startswith("k_"):
        known[w[0]] = {w[i]: int(w[i + 1]) for i in range(1, len(w) - 1, 2)}


def load(d, counter):
    f = glob.glob(O + d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for r in (csv.DictReader(open(f[0])) if f else []):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = load("cal_fetch", "FETCH_SIZE"), load("cal_write", "WRITE_SIZE")
out = {"note": "tools/microbench/fetch_calibration.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): counter value (KB) per launch "
               "against the bytes the kernel is known to touch once, cold (1 GiB buffers, 4 x the Infinity Cache).  factor = known bytes / (counter x 1024): "
               "what a kernel's FETCH_SIZE / WRITE_SIZE has to be multiplied by for that access pattern.", "patterns": {}}
for k, b in known.items():
    e = {"known": b}
    if "read" in b and fetch.get(k):
        tot = b["read"] + b.get("index", 0)
        e["FETCH_SIZE_KB"] = round(fetch[k], 1)
        e["fetch_factor"] = round(tot / (fetch[k] * 1024.0), 4)
    if "write" in b and write.get(k):
        e["WRITE_SIZE_KB"] = round(write[k], 1)
        e["write_factor"] = round(b["write"] / (write[k] * 1024.0), 4)
        if fetch.get(k):
            e["FETCH_SIZE_KB_of_a_store_kernel"] = round(fetch[k], 1)      # read-for-ownership of partially written lines shows up here
    out["patterns"][k] = e
json.dump(out, open(f"{P}/{name}_fetch_calibration.json", "w"), indent=1)
for k, e in out["patterns"].items():
    print(k, e)
