# round 4, step w: views per launch of the shared per-Gaussian forward stage (TGS_FORWARD_GROUP) with the faster kernels
R=$GRAFT_REPO_ROOT; cd $R
for g in 2 1 4 8 2 4; do
  export TGS_FORWARD_GROUP=$g
  echo "group $g: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 20 --warmup 4 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['kernels_ms']['preprocess_fwd'])")"
done
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout 400 -k "batched_backward" 2>&1 < /dev/null | tail -3
