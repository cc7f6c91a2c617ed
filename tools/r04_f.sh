# round 4, step f: GPU suite after the hygiene changes (always-valid light descriptors, options through SyncFreeBatch, 4-map SSIM), the fixed microbenchmark, trainer protocol
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_f; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -6 $O/pytest.txt | cut -c1-400
(cd tools/microbench && timeout 120 /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value valu_issue_rate.hip -o valu_issue_rate && timeout 200 ./valu_issue_rate) > $O/valu_issue_rate.txt 2>&1 < /dev/null; grep "render mix" $O/valu_issue_rate.txt | cut -c1-200
for i in 1 2; do
  echo "trainer sh0 $(timeout 200 python tools/trainer_protocol.py 0 60 2>/dev/null < /dev/null | tail -1 | cut -c1-300)"
done
echo "trainer sh3 $(timeout 200 python tools/trainer_protocol.py 3 60 2>/dev/null < /dev/null | tail -1 | cut -c1-300)"
echo "dropin $(timeout 120 python tools/dropin_loop.py 200 2>/dev/null < /dev/null | tail -1)"
