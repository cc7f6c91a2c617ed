# round 4, step v2: SSIM streaming kernels with unconditional clamped loads (rows really in flight): loss tests + trainer protocol + kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_loss.py -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
for i in 1 2; do
  echo "trainer sh3 $(timeout 200 python tools/trainer_protocol.py 3 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
  echo "trainer sh0 $(timeout 200 python tools/trainer_protocol.py 0 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
