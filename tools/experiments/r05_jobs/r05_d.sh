# round 5, job d: the hand-written list builders (ql_append): parity of the render pair + kernel times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_d; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or seeded or light or overflow or random_small" > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt | cut -c1-300
for rep in 1 2 3; do
  echo "$(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"
done > $O/stage_times.txt 2>&1
cat $O/stage_times.txt
timeout 300 python bench.py --no-cpu --no-secondary > $O/bench.json 2> $O/bench.err < /dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
