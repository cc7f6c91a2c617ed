# debugging aid: the two multi-rank GPU tests by hand, with a hard limit and Python stacks on timeout (SIGABRT -> faulthandler)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mr; mkdir -p $O; cd $R
export PYTHONFAULTHANDLER=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s ABRT 200 python bench.py --gpus 2 --backend gloo --same-device --config 2 --steps 2 --warmup 2 --views-per-gpu 4 --no-cpu > $O/bench2.out 2> $O/bench2.err
echo "bench2 rc=$?"
python - <<'PY'
import re
src = open("tests/test_gpu_multirank.py").read()
w = src.split('_WORKER = r"""')[1].split('"""')[0]
open("gpurun_out/mr/worker.py", "w").write(w)
PY
export TGS_ROOT=$R MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 TGS_OUT=$O/res_w1 timeout -s ABRT 120 python $O/worker.py > $O/w1.out 2> $O/w1.err; echo "w1 rc=$?"
for r in 0 1; do RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r TGS_OUT=$O/res_w2 timeout -s ABRT 150 python $O/worker.py > $O/w2_$r.out 2> $O/w2_$r.err & done
wait; echo "w2 done"
tail -c 1500 $O/bench2.err; tail -c 600 $O/w1.err; tail -c 1500 $O/w2_0.err; tail -c 800 $O/w2_1.err
