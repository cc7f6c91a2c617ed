# the GPU check of a milestone: oracle build, GPU suite, smoke(), the default bench line   (bash tools/gpu_suite.sh [name] on the MI355X box via gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-suite}; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest.txt 2>&1 < /dev/null; tail -6 $O/pytest.txt | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 < /dev/null; tail -2 $O/smoke.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench.err < /dev/null; tail -c 2500 $O/bench_default.json
