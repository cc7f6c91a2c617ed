# round 6, job d: rectangles up to EMIT_RANK tiles walked by the splat's own lane in k_bin_count / k_scatter (4 = round 5) -- stage times x1 / x4 / x8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_d; mkdir -p $O; cd $R
for sc in 1 4 8; do
for L in default er6 er9 er12 default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
