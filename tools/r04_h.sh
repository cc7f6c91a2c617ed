# round 4, step h: cov3D evaluated again in the backward (no copy in the geometry state), mean / cov3D hoisted out of the batch pass's view loop:
# GPU suite, A/B against the r04_a sources, and the two timing-only builds that walk the views in one half of the split pass only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_h; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_r04a.so default libtgs_raster_r04a.so libtgs_raster_skipc.so libtgs_raster_skipg.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for i in 1 2; do
  for lib in default r04a; do
    if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
    echo "dropin $lib $(timeout 120 python tools/dropin_loop.py 200 2>/dev/null < /dev/null | tail -1)"
  done
done
unset TGS_LIBRARY
