# round 6, job s: the tail's zero rows stored by four lanes per row (one 16-B piece each, one instruction) instead of one lane per row (three instructions) -- heavy path of k_render_bwd only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 900 -k "large or seeded or golden or overflow" > $O/pytest.txt 2>&1 < /dev/null; tail -2 $O/pytest.txt | cut -c1-300
for sc in 1 4 8; do
for L in prev default prev default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
