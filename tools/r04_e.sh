# round 4, step e: full GPU suite (per-test timeout), SALU microbenchmark, A/B by library on one box (r03 / f32 chain / lean masks), drop-in with and without the side stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -6 $O/pytest.txt | cut -c1-400
(cd tools/microbench && timeout 120 /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value valu_issue_rate.hip -o valu_issue_rate && timeout 200 ./valu_issue_rate) > $O/valu_issue_rate.txt 2>&1 < /dev/null; grep "W=8\|W=4" $O/valu_issue_rate.txt | cut -c1-200
timeout 900 bash tools/libs.sh "default libtgs_raster_r03.so libtgs_raster_f32chain.so libtgs_raster_lean.so default libtgs_raster_r03.so libtgs_raster_f32chain.so libtgs_raster_lean.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for i in 1 2; do
  for lib in default lean r03; do
    if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
    echo "dropin $lib $(timeout 120 python tools/dropin_loop.py 200 2>/dev/null < /dev/null | tail -1)"
  done
  unset TGS_LIBRARY
  echo "dropin side-stream on $(TGS_SIDE_STREAM=1 timeout 120 python tools/dropin_loop.py 200 2>/dev/null < /dev/null | tail -1)"
done
