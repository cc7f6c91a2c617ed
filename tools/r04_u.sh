# round 4, step u: stream count / forward group of the 8-view step with the faster per-Gaussian kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_u; mkdir -p $O; cd $R
for args in "--streams 4" "--streams 2" "--streams 3" "--streams 6" "--streams 8" "--streams 4"; do
  echo "$args: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 20 --warmup 4 $args 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'])")"
done
