#!/bin/bash
# A/B of two builds of the library on ONE box (run-to-run noise of bench.py is ~1-2 %): bash tools/ab.sh <libB.so> [rounds] [bench args...]
# A = the default library; prints ms per frame of every run and the two means.
B=$1; N=${2:-5}; shift; shift
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export TGS_LIBRARY=$B; else unset TGS_LIBRARY; fi
    python bench.py --no-cpu --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['ms_per_frame_per_gpu'])"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import collections
m=collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    v,x=l.split(); m[v].append(float(x))
for v in sorted(m): print(v, 'mean', sum(m[v])/len(m[v]), 'min', min(m[v]), 'n', len(m[v]))
PY
