# like ab2.sh, also prints the per-Gaussian pass (preprocess_bwd per step) and the drop-in path
B=$1; N=$2; shift; shift
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export TGS_LIBRARY=$B; else unset TGS_LIBRARY; fi
    python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['ms_per_frame_per_gpu'], 'pre_bwd', d['kernels_ms']['preprocess_bwd'])"
  done
done
