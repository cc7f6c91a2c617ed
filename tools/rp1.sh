# one-stream rocprof kernel stats of the bench (every kernel alone on the GPU): tools/rp1.sh <name> [bench args]
N=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/rp1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp1 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-secondary --streams 1 "$@" > $O/bench_one_stream.json 2> $O/rp1.err
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("$O/rp1/**/rp_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
