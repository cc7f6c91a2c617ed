# three-way adjudication (HIP / fp32 oracle / the oracle's text in double) of the round-3 fuzz seeds and of configs 3 and 5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_adjudicate; mkdir -p $O; cd $R
python -c "from oracle import oracle; oracle.build(); print(oracle.threads())"
for seed in 11 12 13 14; do
  timeout 1200 python -m tests.adjudicate $seed 96 1 > $O/light_$seed.txt 2>&1; tail -1 $O/light_$seed.txt | cut -c1-400
  timeout 1200 python -m tests.adjudicate $seed 96 0 > $O/plain_$seed.txt 2>&1; tail -1 $O/plain_$seed.txt | cut -c1-400
done
timeout 1200 python -m tests.adjudicate 2026 128 > $O/seed2026.txt 2>&1; tail -1 $O/seed2026.txt | cut -c1-400
timeout 1500 python -m tests.adjudicate cfg 3 5 > $O/cfg.txt 2>&1; grep -h "dL_drot\|dL_dmeans2D\|dL_dcov" $O/cfg.txt | cut -c1-300
grep -h "MISS" $O/*.txt | cut -c1-300 | head -40
