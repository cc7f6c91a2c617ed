# round 4, step n: clamp bits / camera position asked for before the slab rows (colour half of the split pass); halves timed alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_n; mkdir -p $O; cd $R
timeout 900 bash tools/libs.sh "default libtgs_raster_l.so libtgs_raster_skipc.so libtgs_raster_skipg.so default libtgs_raster_l.so libtgs_raster_skipc.so libtgs_raster_skipg.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout 400 -k "batch" > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
