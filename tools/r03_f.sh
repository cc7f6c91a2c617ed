# round 3, call 6: SLP vectoriser off for every kernel file (A/B by library), bind tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_f; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_bind.py -q -m gpu > $O/pytest_bind.txt 2>&1; tail -3 $O/pytest_bind.txt
for rep in 1 2; do
  python bench.py --no-cpu --no-secondary > $O/bench_slp_$rep.json 2>> $O/err.txt
  TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_noslp.so python bench.py --no-cpu --no-secondary > $O/bench_noslp_$rep.json 2>> $O/err.txt
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_f")
for f in sorted(glob.glob(O + "/bench_*slp_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["value"], d["kernels_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
python tools/dropin_loop.py 2>/dev/null | tail -2
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_noslp.so python tools/dropin_loop.py 2>/dev/null | tail -2
python tools/trainer_protocol.py 0 40; TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_noslp.so python tools/trainer_protocol.py 0 40
python tools/trainer_protocol.py 3 40; TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_noslp.so python tools/trainer_protocol.py 3 40
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_noslp.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_knn.py -q -m gpu -x > $O/pytest_noslp.txt 2>&1; tail -3 $O/pytest_noslp.txt
