# round 6, job y: the default bench line after the last bench.py change (Morton order at x 4 in secondary.spatially_ordered_gaussians)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_y; mkdir -p $O; cd $R
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_y/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["secondary"]["spatially_ordered_gaussians"], {k: v.get("ms_per_frame") for k, v in d["secondary"]["grown_splats"].items() if isinstance(v, dict)})
PY
tail -3 $O/bench.err
