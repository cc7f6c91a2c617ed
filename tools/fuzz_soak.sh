# fuzz soak at the frozen criterion (tests/util.compare: capped bar, recorded routes): bash tools/fuzz_soak.sh <name> <seeds...>   (96 scenes per seed)
R=$GRAFT_REPO_ROOT; NAME=$1; shift; O=$R/gpurun_out/$NAME; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for seed in "$@"; do
  timeout 1200 python tests/tools/fuzz_vs_oracle.py $seed 96 > $O/seed_$seed.txt 2>&1 < /dev/null
  tail -1 $O/seed_$seed.txt | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('seed', 'scenes', 'misses', 'over_1e4', 'scenes_beyond_the_oracle_route', 'routes', 'worst_rel_l2', 'largest_ok')}))
except Exception as e: print('seed $seed: no result', e)
" | tee -a $O/summary.txt
done
grep -h "MISS" $O/seed_*.txt | cut -c1-300 | head
