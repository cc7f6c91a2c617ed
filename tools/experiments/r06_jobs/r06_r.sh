# round 6, job r: the zero rows of a tile's never-visited tail written at the END of k_render_fwd's workgroups (nothing waits for them) instead of at the head of k_render_bwd's -- parity + API suites, then per-stage times and the batch step against the previous library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_r; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q --timeout 900 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
for sc in 1 4 8; do
for L in prev default prev default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
for sc in 1 4 8; do
for L in prev default prev default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== batch $L x$sc" | tee -a $O/batch_times.txt
  timeout 300 python tools/batch_stage_times.py $sc 2>&1 | tail -3 | cut -c1-400 | tee -a $O/batch_times.txt
done
done
