# round 6, job w: k_render_bwd with the tail's zero rows behind the first barrier AND barriers that order LDS only in the round loop (s_waitcnt lgkmcnt(0) + s_barrier: global stores are never waited for)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_w; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_ldsbar.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 900 -k "large or seeded or golden or overflow" > $O/pytest.txt 2>&1 < /dev/null; tail -2 $O/pytest.txt | cut -c1-300
for sc in 1 4 8; do
for L in default ldsbar default ldsbar; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
