# round 6, job o: what the rows of the never-visited tail cost (timing only: -DTGS_EXP_NO_TAIL=1 writes no zero rows behind a tile's deepest contributor and sums 30 % of a splat's rows)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_o; mkdir -p $O; cd $R
for sc in 1 4 8; do
for L in default notail default notail; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
