# round 4, step p: odd workgroups of the split pass swap the roles of their wave pairs; batch tests + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_p; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_noswap.so libtgs_raster_l.so default libtgs_raster_noswap.so libtgs_raster_l.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
