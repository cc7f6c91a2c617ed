# persistent forward render kernel: parity + A/B against the previous build (libtgs_raster_bs0.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "not fuzz" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
bash tools/libs.sh "default libtgs_raster_bs0.so default libtgs_raster_bs0.so" > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/libs.sh "default libtgs_raster_bs0.so" --streams 1 > $O/ab1.txt 2>&1; cat $O/ab1.txt
python tools/dropin_loop.py > $O/dropin.txt 2>&1; tail -2 $O/dropin.txt
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_bs0.so python tools/dropin_loop.py > $O/dropin0.txt 2>&1; tail -2 $O/dropin0.txt
