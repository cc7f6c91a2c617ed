# round 5, job b: more of the decomposition of k_render_bwd: slab-row stores (e6), zero-fill of the tail rows (e7), LDS-only barriers (e8)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_b; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in default e6 e7 e8; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "$lib $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"
done
done > $O/stage_times.txt 2>&1
cat $O/stage_times.txt
export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_e8.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or seeded" > $O/pytest_e8.txt 2>&1 < /dev/null; tail -5 $O/pytest_e8.txt
