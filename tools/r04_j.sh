# drop-in frame without the profiler: default library (env unset / named through TGS_LIBRARY) against the r04_a sources
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  unset TGS_LIBRARY
  echo "dropin unset   $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
  export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster.so
  echo "dropin named   $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
  export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_r04a.so
  echo "dropin r04a    $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
done
