R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_soak3; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for seed in 54 58; do
  timeout 600 python -m tests.adjudicate $seed 96 > $O/seed_${seed}b.txt 2>&1 < /dev/null; grep "MISS\|own inputs" $O/seed_${seed}b.txt | cut -c1-330; tail -1 $O/seed_${seed}b.txt | cut -c1-150
done
