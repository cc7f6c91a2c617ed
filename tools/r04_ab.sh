# round 4, step ab: split_pass on by default: run_views tests, multirank tests, bench A/B by --no-split-pass, rccl_world1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ab; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
for i in 1 2 3; do
  echo "default: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 45 --warmup 6 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['config']['native_calls'][:20])")"
  echo "no-split-pass: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 45 --warmup 6 --no-split-pass 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['config']['native_calls'][:20])")"
done
timeout 300 python tools/rccl_world1.py 2>/dev/null < /dev/null | tail -1 | cut -c1-600
