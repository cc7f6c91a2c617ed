# final check of the round: the GPU suite, the smoke entry, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
