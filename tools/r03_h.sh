# round 3, call 10: default bench line end to end (time it), full GPU suite, 256 binning chunks A/B on the drop-in path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_h; mkdir -p $O; cd $R
SECONDS=0; timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? in ${SECONDS}s"; python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_h")
try:
    d = json.loads(open(O + "/bench_default.json").read().strip().splitlines()[-1])
    print("default:", d["ms_per_step"], d["value"], d["config"]["dropin_ms_per_frame"], d["secondary"]["trainer_protocol"]["ms_per_step"], d["cpu_baseline"], d["roofline"]["frac"], d["roofline"]["traffic"], d["hbm_frac_of_8TBps"])
except Exception as e:
    print("ERR", e)
PY
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
cp gpurun_out/parity.json $O/parity.json 2>/dev/null
for rep in 1 2; do
  timeout 200 python tools/dropin_loop.py
  TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_bin256.so timeout 200 python tools/dropin_loop.py
done
timeout 200 python tools/trainer_protocol.py 0 40; TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_bin256.so timeout 200 python tools/trainer_protocol.py 0 40
timeout 300 python bench.py --no-cpu --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bin128', d['ms_per_step'], d['kernels_ms'])"
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_bin256.so timeout 300 python bench.py --no-cpu --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bin256', d['ms_per_step'], d['kernels_ms'])"
