# round 4, step y: the per-Gaussian pass of the first round of views beside the remaining per-pixel backwards (SyncFreeBatch.split_pass), again, with the faster pass
R=$GRAFT_REPO_ROOT; cd $R
for s in 0 1 0 1 0 1 0 1 0 1; do
  export TGS_SPLIT_PASS=$s
  echo "split_pass $s: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 45 --warmup 6 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'])")"
done
