R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 400 -k "cutoff_flip" -s 2>&1 < /dev/null | grep -v Warning | tail -8 | cut -c1-250
