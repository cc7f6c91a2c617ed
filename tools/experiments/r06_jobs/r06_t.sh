# round 6, job t: why sort + gather take 172 us per frame in the 8-view batch at x 8 and 144 in the one-view path: kernel stats of both
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_t; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/batch8 -o rp -- python3 $R/tools/batch_stage_times.py 8 > $O/batch8.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single8 -o rp -- python3 $R/tools/stage_times.py 8 > $O/single8.log 2>&1
for d in batch8 single8; do echo "== $d"; f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
done
tail -2 $O/batch8.log | cut -c1-400
