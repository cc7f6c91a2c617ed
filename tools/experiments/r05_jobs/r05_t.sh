R=$GRAFT_REPO_ROOT; cd $R
timeout 200 python tools/trainer_protocol.py 0 20 8 2>&1 | tail -8 | cut -c1-130
TP_NOGC=1 timeout 200 python tools/trainer_protocol.py 0 20 8 2>&1 | tail -8 | cut -c1-130
