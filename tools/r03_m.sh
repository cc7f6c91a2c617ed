R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not fuzz" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
bash tools/libs.sh "default libtgs_raster_bs0.so default libtgs_raster_bs0.so" > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/libs.sh "default libtgs_raster_bs0.so" --streams 1 > $O/ab1.txt 2>&1; cat $O/ab1.txt
