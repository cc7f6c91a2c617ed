R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 800 python tests/tools/diag_pixels.py 37 89 2>&1 < /dev/null | grep -v Warning | cut -c1-200
timeout 800 python tests/tools/diag_scene.py 37 89 2>&1 < /dev/null | grep -v Warning | cut -c1-300
