# round 4, step o: colour half of the split pass one view ahead; batch tests + A/B against commit 6c7d1b2 (libtgs_raster_l.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_o; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_multiview.py -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_l.so default libtgs_raster_l.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
