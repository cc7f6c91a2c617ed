# final check of round 4: the GPU suite, the smoke entry, the default bench line (the fuzz at the final bar: tools/r04_soak.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 < /dev/null; tail -3 $O/smoke.txt | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -c 1500 $O/bench_default.json
