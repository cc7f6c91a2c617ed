# round 3, second GPU call: whole GPU suite on the new API + light-tile kernels, A/B of the light kernels, timelines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
cp gpurun_out/parity.json $O/parity.json 2>/dev/null
for rep in 1 2; do
  TGS_LIGHT_TILES=0 python bench.py --no-cpu --no-secondary > $O/bench_light0_$rep.json 2>> $O/bench.err
  TGS_LIGHT_TILES=1 python bench.py --no-cpu --no-secondary > $O/bench_light1_$rep.json 2>> $O/bench.err
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_c")
for f in sorted(glob.glob(O + "/bench_light*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["value"], d["kernels_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps.so python tests/tools/timeline.py > $O/timeline_light1.txt 2>&1; tail -24 $O/timeline_light1.txt
python bench.py --no-cpu > $O/bench_full.json 2>> $O/bench.err; python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_c")
d = json.loads(open(O + "/bench_full.json").read().strip().splitlines()[-1])
print("full:", d["ms_per_step"], d["config"]["dropin_ms_per_frame"], d["secondary"]["trainer_protocol"], d["roofline"])
PY
