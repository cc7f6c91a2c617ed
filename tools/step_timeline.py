"""Where a step of the headline path spends its time ON THE GPU: reads a rocprofv3 kernel trace (per-dispatch start / end timestamps) of
`bench.py --steps N` and reports, for the timed steps, how many kernels run concurrently over time, the time no kernel runs, and the per-kernel
sums.   python tools/step_timeline.py <dir with *_kernel_trace.csv> [steps to analyse from the end]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]) for r in rows), key=lambda t: t[0])
# steps are separated by the per-Gaussian batch pass (one per step)
ends = [e for s, e, n in ev if "k_preprocess_bwd_batch" in n]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if len(ends) < nsteps + 1:
    raise SystemExit(f"only {len(ends)} steps in the trace")
t0, t1 = ends[-nsteps - 1], ends[-1]
sel = [(max(s, t0), min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
pts = sorted([(s, 1) for s, e, n in sel] + [(e, -1) for s, e, n in sel])
hist = collections.Counter(); cur = 0; last = t0
for t, dlt in pts:
    hist[cur] += t - last; last = t; cur += dlt
tot = t1 - t0
print(f"{nsteps} steps, {tot / nsteps / 1e3:.1f} us per step on the GPU's clock")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k] / tot * 100:5.1f} %  ({hist[k] / nsteps / 1e3:7.1f} us per step)")
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    per[n][0] += e - s; per[n][1] += 1
print("  kernel: launches per step, mean elapsed us (stretched by sharing), share of sum")
S = sum(v[0] for v in per.values())
for n, (ns, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n[:60]:60s} {c / nsteps:5.1f} {ns / c / 1e3:8.1f} {ns / S * 100:5.1f} %")
print(f"  sum of elapsed kernel times per step {S / nsteps / 1e3:.1f} us = {S / tot:.2f} x the step")
