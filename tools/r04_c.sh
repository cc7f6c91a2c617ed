# round 4, step c: hi + lo conic slab rows, double slab sums, f64 chain: parity + cost
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_c; mkdir -p $O; cd $R
python -c "from oracle import oracle; oracle.build(force=True)"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for seed in 11 12 13 14; do timeout 900 python -m tests.adjudicate $seed 96 0 > $O/plain_$seed.txt 2>&1; grep -h "^seed\|^{" $O/plain_$seed.txt | cut -c1-250; done
timeout 900 python -m tests.adjudicate 2026 128 > $O/s2026.txt 2>&1; grep -h "^seed\|^{" $O/s2026.txt | cut -c1-250
bash tools/libs.sh "default libtgs_raster_f32chain.so libtgs_raster_split2.so default libtgs_raster_f32chain.so libtgs_raster_split2.so" > $O/ab.txt 2>&1; cat $O/ab.txt
