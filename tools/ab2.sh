# A/B on one box: default lib vs TGS_LIBRARY variant, for a given set of bench args; prints ms/frame and preprocess_fwd alone time
B=$1; N=$2; shift; shift
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export TGS_LIBRARY=$B; else unset TGS_LIBRARY; fi
    python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['ms_per_frame_per_gpu'], d['kernels_ms']['preprocess_fwd'], d['kernels_ms']['scatter'], d['kernels_ms']['tile_sort'], d['kernels_ms']['render_fwd'], d['kernels_ms']['render_bwd'])"
  done
done
