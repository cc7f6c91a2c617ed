# round 6, job p: the zero rows of a tile's never-visited tail written behind the first barrier of k_render_bwd (slot asked for with the prologue's loads) -- parity, then per-stage times against the previous library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_p; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 900 > $O/pytest_parity.txt 2>&1 < /dev/null; tail -4 $O/pytest_parity.txt | cut -c1-300
for sc in 1 4 8; do
for L in prev default prev default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
