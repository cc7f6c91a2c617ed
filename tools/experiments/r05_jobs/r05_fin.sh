R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for lib in default fin1 fin3 fin4; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "$lib $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1 | cut -c40-330)"
done; done
