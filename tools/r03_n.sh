R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "not fuzz" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for sl in 256 384 512 640 768 1024; do echo slots $sl; TGS_RENDER_SLOTS=$sl bash tools/libs.sh "default" --streams 1; done > $O/slots.txt 2>&1; cat $O/slots.txt
bash tools/libs.sh "libtgs_raster_bs0.so" --streams 1
