R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "test_vs_oracle_seeded" 2>&1 | tail -12 | cut -c1-400
