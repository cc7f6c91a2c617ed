# round 4, step v: SSIM streaming kernels with two columns per lane: loss tests + A/B by the TGS_LOSS_PX2 knob + kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_loss.py -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
for i in 1 2; do
  for k in 1 0; do
    export TGS_LOSS_PX2=$k
    echo "trainer sh3 px2=$k $(timeout 200 python tools/trainer_protocol.py 3 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
    echo "trainer sh0 px2=$k $(timeout 200 python tools/trainer_protocol.py 0 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
  done
done
cd /tmp && export TMPDIR=/tmp
export TGS_LOSS_PX2=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
