R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq -o pmc -- $B > $O/pmc_sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -o pmc -- $B > $O/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $B > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $B > $O/pmc_write.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05_u"
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            if "batch" not in k and "pair" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(d, k, {c: round(v / n[(k, c)], 1) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
rm -rf $O/pmc_*/pmc_kernel_trace.csv; du -sh $O
