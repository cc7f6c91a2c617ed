R=$GRAFT_REPO_ROOT; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 800 python tests/tools/diag_pixels.py 23 93 2>&1 < /dev/null | grep -v Warning | cut -c1-200
