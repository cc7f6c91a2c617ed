R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_final
timeout 500 python bench.py > gpurun_out/r05_final/bench_default.json 2> gpurun_out/r05_final/bench.err; tail -c 600 gpurun_out/r05_final/bench_default.json
