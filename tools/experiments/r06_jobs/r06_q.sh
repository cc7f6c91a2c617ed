# round 6, job q: timing only -- a 1-byte flag per dead row instead of its 48-byte zero row in k_render_bwd (heavy path), against no write at all (notail) and the zero rows (default)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_q; mkdir -p $O; cd $R
for sc in 4 8; do
for L in default byteflag notail default byteflag notail; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
