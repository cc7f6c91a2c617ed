# round 3, call 5: trimmed streaming SSIM, fused binding-side op
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_loss.py tests/test_bind.py tests/test_gpu_api.py -q -m gpu -k "loss or ssim or bind or example" > $O/pytest_sel.txt 2>&1; tail -8 $O/pytest_sel.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer0 -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer_sh0_rocprof.json 2>> $O/err.txt
cd $R
cat $O/trainer_sh0_rocprof.json; python - <<'PY'
import csv, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_e")
rows = list(csv.DictReader(open(O + "/rp_trainer0/rp_kernel_stats.csv")))
for r in rows[:8]:
    print(f"{r['Name'][:60]:60s} {float(r['AverageNs'])/1e3:8.1f} us")
PY
python tools/trainer_protocol.py 0 40; python tools/trainer_protocol.py 3 40
