# round 4, step x: staggered lanes in run_views (half of the lanes one forward ahead): run_views tests + A/B by TGS_STAGGER
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_x; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -m gpu -x -q --timeout 400 -k "run_views or sync_free or multirank or rank or range_wise" > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
for s in 1 0 1 0 1 0; do
  export TGS_STAGGER=$s
  echo "stagger $s: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['config']['frames_rerendered'])")"
done
for s in 1 0; do
  export TGS_STAGGER=$s
  echo "stagger $s loss: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 --loss 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'])")"
  echo "stagger $s streams 6: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 --streams 6 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'])")"
done
