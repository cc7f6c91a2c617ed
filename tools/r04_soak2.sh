# end of round 4: ten more fuzz seeds at the final bar
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_soak3; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for seed in 51 52 53 54 55 56 57 58 59 60; do
  timeout 600 python -m tests.adjudicate $seed 96 > $O/seed_$seed.txt 2>&1 < /dev/null; tail -1 $O/seed_$seed.txt | cut -c1-150
done
grep -h "MISS" $O/*.txt | cut -c1-300 | head -20
