cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout 600 -k "discards" 2>&1 | grep -v "^$" | tail -40 | cut -c1-300
