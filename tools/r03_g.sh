# round 3, call 8: RCCL at world size 1 as a child program, checkpoints / bind tests, k_scan<16>, light groups on the trainer protocol
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_g; mkdir -p $O; cd $R
timeout 240 python tools/rccl_world1.py 3 > $O/rccl_world1.json 2> $O/rccl_world1.err; echo "rccl rc=$?"; cat $O/rccl_world1.json; tail -3 $O/rccl_world1.err
timeout 300 python -m pytest tests/test_checkpoints.py tests/test_bind.py -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 200 python tools/trainer_protocol.py 0 40; TGS_LIGHT_TILES=1 timeout 200 python tools/trainer_protocol.py 0 40
timeout 200 python tools/trainer_protocol.py 3 40; TGS_LIGHT_TILES=1 timeout 200 python tools/trainer_protocol.py 3 40
timeout 200 python tools/dropin_loop.py; TGS_LIGHT_TILES=1 timeout 200 python tools/dropin_loop.py
timeout 400 python bench.py --no-cpu > $O/bench_full.json 2> $O/bench_full.err; python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_g")
d = json.loads(open(O + "/bench_full.json").read().strip().splitlines()[-1])
print("full:", d["ms_per_step"], d["value"], d["config"]["dropin_ms_per_frame"], d["secondary"]["trainer_protocol"]["ms_per_step"], d["secondary"].get("rccl_world1"), d["kernels_ms"])
PY
