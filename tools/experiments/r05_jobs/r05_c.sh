# round 5, job c: light groups in the single-view kernels (TGS_LIGHT_TILES=1) -- kernel alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_c; mkdir -p $O; cd $R
for rep in 1 2; do
for lt in 0 1; do
  echo "light=$lt $(TGS_LIGHT_TILES=$lt timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"
done
done > $O/stage_times.txt 2>&1
cat $O/stage_times.txt
export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps.so
TGS_LIGHT_TILES=1 timeout 200 python tests/tools/timeline.py > $O/timeline_light.txt 2>&1 < /dev/null
grep -A30 "^bwd" $O/timeline_light.txt
