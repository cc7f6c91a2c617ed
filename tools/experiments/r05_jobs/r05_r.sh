R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do for lib in default base; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "$lib $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1 | cut -c40-330)"
done; done
unset TGS_LIBRARY
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "quadrant" 2>&1 | grep "quadrants named\|passed\|failed" | head -20
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_base.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "quadrant" 2>&1 | grep "quadrants named\|passed\|failed" | head -20
