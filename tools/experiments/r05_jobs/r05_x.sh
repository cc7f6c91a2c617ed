R=$GRAFT_REPO_ROOT; cd $R
for sm in 1 4 8; do
  echo "x$sm adaptive $(timeout 200 python tools/stage_times.py $sm 2>/dev/null < /dev/null | tail -1 | cut -c1-330)"
done
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 400 python bench.py --no-cpu 2>/dev/null | tail -1 > /tmp/b.json
python3 - <<PY
import json
d=json.load(open('/tmp/b.json')); s=d['secondary']
print(d['value'], d['ms_per_step'], d['kernels_ms'], 'dropin', s['dropin_api']['ms_per_frame'], s['trainer_protocol']['ms_per_step'], s['grown_splats'])
PY
