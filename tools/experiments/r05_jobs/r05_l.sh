R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do for lib in default fp; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "$lib $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1 | cut -c40-260)"
done; done
export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_fp.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or seeded or light" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu --no-secondary 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp', d['value'], d['ms_per_step'], d['kernels_ms'])"
unset TGS_LIBRARY
timeout 300 python bench.py --no-cpu --no-secondary 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['kernels_ms'])"
