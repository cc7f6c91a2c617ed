R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for st in 4 3 5 6 8; do
  timeout 300 python bench.py --no-secondary --no-cpu --streams $st 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("streams $st", d['value'], d['ms_per_step'])
PY
done; done
