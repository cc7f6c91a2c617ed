# round 6, job l: bench line after the side streams became shared by all SyncFreeBatch objects (grown splats / Morton order were measured by the 3rd / 4th batch object of the process)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_l; mkdir -p $O; cd $R
timeout 600 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_l/bench.json').read().strip().splitlines()[-1])
s=d['secondary']
print(d['value'], d['ms_per_step'], 'dropin', s['dropin_api']['ms_per_frame'], 'trainer', s['trainer_protocol']['ms_per_step'], 'grown', {k:(v['ms_per_frame'], v.get('dropin_ms_per_frame')) for k,v in s['grown_splats'].items() if k!='what'}, 'morton', s['spatially_ordered_gaussians']['ms_per_frame'])
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "known_miss or take_the_same" 2>&1 | tail -3
