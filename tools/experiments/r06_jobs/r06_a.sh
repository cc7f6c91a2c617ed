# round 6, job a: the 8-waves-per-tile backward (k_render_bwd_h, -DTGS_BWD_HALF=1) -- parity on the single-view paths, then per-stage times against the default library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_a; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for L in h176 default; do
  K="seeded or golden or fuzz or overflow or large or nonfinite or known_miss"; if [ $L = default ]; then K="nonfinite or known_miss"; fi
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 -k "$K" > $O/pytest_$L.txt 2>&1 < /dev/null; tail -5 $O/pytest_$L.txt | cut -c1-300
done
for sc in 1 4; do
for L in default h176 h128 h192 default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
