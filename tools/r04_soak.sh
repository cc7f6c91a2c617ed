# end of round 4: the GPU suite on a fresh box and the fuzz at the final bar (tests/adjudicate.py: five fp32 builds of the oracle against its double build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_soak; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 400 > $O/pytest_1.txt 2>&1 < /dev/null; tail -1 $O/pytest_1.txt | cut -c1-200
for seed in 11 12 13 14 21 22 23 24 25 26 27 28 31 32 33 34 35 36 37 38 39 40 41 42 43 44 45 46; do
  timeout 900 python -m tests.adjudicate $seed 96 > $O/seed_$seed.txt 2>&1 < /dev/null; tail -1 $O/seed_$seed.txt | cut -c1-130
done
timeout 900 python -m tests.adjudicate 2026 128 > $O/seed_2026.txt 2>&1 < /dev/null; tail -1 $O/seed_2026.txt | cut -c1-130
timeout 900 python -m tests.adjudicate cfg 3 5 > $O/cfg.txt 2>&1 < /dev/null; tail -1 $O/cfg.txt | cut -c1-200
grep -h "MISS" $O/*.txt | cut -c1-300 | head
