# round 4, step b: the f64 per-Gaussian chain -- parity (GPU suite subset + adjudication of the scenes that missed) and its cost (A/B by library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b; mkdir -p $O; cd $R
python -c "from oracle import oracle; oracle.build(force=True)"
timeout 1500 python -m pytest tests/test_bind.py tests/test_checkpoints.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for seed in 12 13; do timeout 900 python -m tests.adjudicate $seed 96 0 > $O/plain_$seed.txt 2>&1; grep -h "^seed\|^{" $O/plain_$seed.txt | cut -c1-330; done
timeout 900 python -m tests.adjudicate cfg 5 > $O/cfg.txt 2>&1; grep -h "dL_drot\|dL_dscales\|dL_dcov" $O/cfg.txt | cut -c1-300
bash tools/libs.sh "default libtgs_raster_f32chain.so default libtgs_raster_f32chain.so" > $O/ab.txt 2>&1; cat $O/ab.txt
for x in default f32chain; do
  if [ $x = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=youreditableavatar_amd/lib/libtgs_raster_$x.so; fi
  python tools/stage_times.py > $O/stage_$x.txt 2>&1; tail -25 $O/stage_$x.txt
done
