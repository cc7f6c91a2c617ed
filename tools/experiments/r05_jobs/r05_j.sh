R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j; mkdir -p $O; cd $R
for rep in 1 2; do for lib in default p16 p24 p32; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "$lib $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"
done; done
for lib in default p24; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 300 python bench.py --no-cpu --no-secondary 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['kernels_ms'])"
done
