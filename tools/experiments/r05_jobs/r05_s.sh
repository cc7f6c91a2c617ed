R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for lib in default base; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 600 python bench.py 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
d=json.load(open('/tmp/b.json'))
s=d['secondary']
print("$lib", d['value'], d['ms_per_step'], 'bwd_us', d['kernels_ms'].get('render_bwd'), 'dropin', s['dropin_api']['ms_per_frame'], 'trainer', s.get('trainer_protocol',{}).get('sh0',{}), s.get('trainer_protocol',{}).get('sh3',{}))
PY
done; done
