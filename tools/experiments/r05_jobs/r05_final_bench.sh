R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_final
for i in 1 2; do
timeout 500 python bench.py --no-cpu > gpurun_out/r05_final/bench_$i.json 2> gpurun_out/r05_final/bench.err
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r05_final/bench_$i.json').read().strip().split('\n')[-1]); s=d['secondary']
print(d['value'], d['timing']['ms_per_step_blocks'], d['roofline']['frac'], 'dropin', s['dropin_api']['ms_per_frame'], s['trainer_protocol']['ms_per_step'])
PY
done
