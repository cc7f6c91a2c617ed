R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for lib in default rows1 rows2; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 300 python bench.py --no-secondary --no-cpu 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$lib", d['value'], d['ms_per_step'], d['kernels_ms'])
PY
done; done
unset TGS_LIBRARY
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -2
