# round 6, job e: which class of k_tile_sort sets its time?  timing-only builds (-DTGS_SORT_ONLY=1 / 2 / 3: only the heavy / mid / light class sorts) under the kernel trace, x1 and x4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_e; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for sc in 1 4; do
for L in default so1 so2 so3; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_${L}_x$sc -o rp -- python3 $R/tools/stage_times.py $sc > $O/rp_${L}_x$sc.log 2>&1
  echo "== $L x$sc: $(grep k_tile_sort $O/rp_${L}_x$sc/rp_kernel_stats.csv | cut -d, -f1-6 | cut -c1-120)" | tee -a $O/sort_classes.txt
done
done
find $O -name "*kernel_trace.csv" -delete
