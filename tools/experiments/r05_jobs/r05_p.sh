R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do timeout 120 python tools/loss_times.py 2>&1 | tail -2; done
timeout 300 python -m pytest tests/test_loss.py -x -q -m gpu 2>&1 | tail -2
