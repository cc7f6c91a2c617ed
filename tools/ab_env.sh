# bench under several values of one environment variable: tools/ab_env.sh VAR "v1 v2 ..." [bench args]; prints ms/frame and the stage times alone
V=$1; VALS=$2; shift; shift
for x in $VALS; do
  export $V=$x
  python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$V=$x', d['config']['ms_per_frame_per_gpu'], ' '.join(f'{n}={k[n]}' for n in k))"
done
