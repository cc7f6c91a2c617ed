# final check of round 4: the GPU suite, the smoke entry, the default bench line, and the fuzz at the final bar over ~900 scenes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 < /dev/null; tail -3 $O/smoke.txt | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -c 1500 $O/bench_default.json
for seed in 11 12 13 14; do
  timeout 900 python -m tests.adjudicate $seed 96 1 > $O/light_$seed.txt 2>&1 < /dev/null; tail -1 $O/light_$seed.txt | cut -c1-200
  timeout 900 python -m tests.adjudicate $seed 96 0 > $O/plain_$seed.txt 2>&1 < /dev/null; tail -1 $O/plain_$seed.txt | cut -c1-200
done
timeout 900 python -m tests.adjudicate 2026 128 > $O/seed2026.txt 2>&1 < /dev/null; tail -1 $O/seed2026.txt | cut -c1-200
timeout 900 python -m tests.adjudicate cfg 3 5 > $O/cfg.txt 2>&1 < /dev/null; grep -h "dL_drot\|dL_dmeans2D\|dL_dcov" $O/cfg.txt | cut -c1-200
grep -h "MISS" $O/*.txt | cut -c1-300 | head
