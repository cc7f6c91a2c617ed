# round 6, job v: slab rows on a 64-byte pitch (one cache line per row; -DTGS_SLAB_ROW=4) against the 48-byte pitch: parity, per-stage times, the batch step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_v; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_pitch64.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 900 -k "large or seeded or golden or overflow" > $O/pytest.txt 2>&1 < /dev/null; tail -2 $O/pytest.txt | cut -c1-300
for sc in 1 4 8; do
for L in default pitch64 default pitch64; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
for sc in 1 4; do
for L in default pitch64 default pitch64; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== batch $L x$sc" | tee -a $O/batch_times.txt
  timeout 300 python tools/batch_stage_times.py $sc 2>&1 | tail -1 | cut -c1-400 | tee -a $O/batch_times.txt
done
done
