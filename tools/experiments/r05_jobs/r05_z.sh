R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for rep in 1 2; do
  timeout 300 python bench.py --no-secondary --no-cpu 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("new", d['value'], d['ms_per_step'], d['kernels_ms'])
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -2
for seed in 81 82 83; do timeout 600 python tests/tools/fuzz_vs_oracle.py $seed 96 2>&1 | tail -1 | cut -c1-200; done
