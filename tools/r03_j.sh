# block-serial forward kernel: parity, A/B against the one-wave-per-block kernel, wave-level utilisation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_j; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not fuzz" > $O/pytest_parity.txt 2>&1; tail -8 $O/pytest_parity.txt
bash tools/libs.sh "default libtgs_raster_bs0.so default libtgs_raster_bs0.so" > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/libs.sh "default libtgs_raster_bs0.so" --streams 1 > $O/ab1.txt 2>&1; cat $O/ab1.txt
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps.so timeout 600 python tests/tools/timeline.py > $O/timeline.txt 2>&1
head -12 $O/timeline.txt
