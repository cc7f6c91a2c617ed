R=$GRAFT_REPO_ROOT; cd $R
for sm in 1 4 8; do for lib in default coop64 coop32 coop16; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "x$sm $lib $(timeout 200 python tools/stage_times.py $sm 2>/dev/null < /dev/null | tail -1 | cut -c1-330)"
done; done
