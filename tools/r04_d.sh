# round 4, step d: in-kernel Meta clear, side-stream colours, cov3D backward once per batch, exp2 oracle variant: full GPU suite + timings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_d; mkdir -p $O; cd $R
python -c "from oracle import oracle; oracle.build(force=True)"
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for seed in 14; do timeout 900 python -m tests.adjudicate $seed 96 0 > $O/plain_$seed.txt 2>&1; grep -h "^seed\|^{" $O/plain_$seed.txt | cut -c1-250; done
for i in 1 2; do
  TGS_SIDE_STREAM=1 python tools/dropin_loop.py 200 2>/dev/null | tail -1
  TGS_SIDE_STREAM=0 python tools/dropin_loop.py 200 2>/dev/null | tail -1
done
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernels_ms'), {k:(v.get('ms_per_frame') or v.get('ms_per_step')) for k,v in d.get('secondary',{}).items() if isinstance(v,dict)})"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_dropin -- python3 $R/tools/dropin_loop.py 80 > $O/prof_dropin.log 2>&1; cd $R
f=$(ls $O/prof_dropin/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_dropin.csv; head -16 $f | cut -c1-150
