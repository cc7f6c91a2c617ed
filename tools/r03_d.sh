# round 3, call 4: streaming SSIM kernels, dc/rest SH colours, light groups through the single-view API; trainer protocol A/B + rocprof
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_loss.py tests/test_gpu_api.py tests/test_gpu_parity.py -q -m gpu -k "loss or ssim or sh_color or light or example or training_style or golden" > $O/pytest_sel.txt 2>&1; tail -8 $O/pytest_sel.txt
for v in 0 1; do
  TGS_LOSS_TILED=$v python tools/trainer_protocol.py 0 40 > $O/trainer_sh0_tiled$v.json 2>> $O/err.txt
  TGS_LOSS_TILED=$v python tools/trainer_protocol.py 3 40 > $O/trainer_sh3_tiled$v.json 2>> $O/err.txt
done
cat $O/trainer_*.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer0 -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer_sh0_rocprof.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer3 -o rp -- python3 $R/tools/trainer_protocol.py 3 40 > $O/trainer_sh3_rocprof.json 2>> $O/err.txt
cd $R
head -16 $O/rp_trainer0/*/rp_kernel_stats.csv 2>/dev/null | cut -c1-150 || find $O/rp_trainer0 -name "*stats*" | head
python bench.py --no-cpu > $O/bench_full.json 2>> $O/err.txt; python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_d")
d = json.loads(open(O + "/bench_full.json").read().strip().splitlines()[-1])
print("full:", d["ms_per_step"], d["value"], d["config"]["dropin_ms_per_frame"], d["secondary"]["trainer_protocol"]["ms_per_step"], d["kernels_ms"])
PY
