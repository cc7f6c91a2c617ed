R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for lib in default gt0 gt2; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "== $lib"; timeout 120 python tools/loss_times.py 2>&1 | tail -2
done; done
