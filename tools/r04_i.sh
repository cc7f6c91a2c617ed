# kernel stats of the drop-in loop, new sources against the r04_a sources
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_i; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for lib in default r04a; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_$lib -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin_$lib.json 2> $O/rp_$lib.err < /dev/null
done
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -d, -f1-4 | cut -c1-50,90-300; done
