# round 4, step q: forward per-Gaussian kernels ask for their view-independent inputs once, up front, with both halves of the SH rows; colour half of
# the split backward pass one view ahead: GPU suite + A/B against commit 6c7d1b2 (libtgs_raster_l.so) + kernel stats of the drop-in loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_q; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_l.so default libtgs_raster_l.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for i in 1 2; do
  for lib in default l; do
    if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
    echo "dropin $lib $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in default l; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_$lib -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin_$lib.json 2> $O/rp_$lib.err < /dev/null
done
unset TGS_LIBRARY
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
