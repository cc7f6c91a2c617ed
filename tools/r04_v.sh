# round 4, step v3: SSIM streaming kernels with two columns per lane AND rows really in flight (TGS_LOSS_PX2=1) against the one-column kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v; mkdir -p $O; cd $R
for k in 1 0 1 0; do
  export TGS_LOSS_PX2=$k
  timeout 300 python -m pytest tests/test_loss.py -m gpu -x -q --timeout 200 2>&1 < /dev/null | tail -1 | cut -c1-80
  echo "trainer sh0 px2=$k $(timeout 200 python tools/trainer_protocol.py 0 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
done
cd /tmp && export TMPDIR=/tmp
export TGS_LOSS_PX2=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
