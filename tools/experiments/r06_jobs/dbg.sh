cd $GRAFT_REPO_ROOT; timeout 300 python scratch/dbg_lm.py 2>&1 | tail -8 | cut -c1-300
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -m gpu -x -q --timeout 700 -k "level_major or run_views or batched or live_sh or range_wise" 2>&1 | tail -5 | cut -c1-400
