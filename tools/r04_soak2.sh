# end of round 4: sixteen more fuzz seeds at the final bar
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_soak2; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for seed in 31 32 33 34 35 36 37 38 39 40 41 42 43 44 45 46; do
  timeout 900 python -m tests.adjudicate $seed 96 > $O/seed_$seed.txt 2>&1 < /dev/null; tail -1 $O/seed_$seed.txt | cut -c1-150
done
grep -h "MISS" $O/*.txt | cut -c1-300 | head -20
