# round 4, step ad: k_scatter asks for the next step's Gaussians behind its barrier: GPU suite + kernel stats of the drop-in loop for both libraries
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ad; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -2 $O/pytest.txt | cut -c1-300
timeout 600 bash tools/libs.sh "libtgs_raster_prev.so default libtgs_raster_prev.so default" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for lib in prev default; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_$lib -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin_$lib.json 2> $O/rp_$lib.err < /dev/null
done
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
