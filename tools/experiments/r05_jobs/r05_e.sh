# round 5, job e: fp32 / double per-Gaussian chain chosen per wave: parity + the batch pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_e; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt | cut -c1-300
for i in 1 2; do timeout 300 python bench.py --no-cpu --no-secondary > $O/bench$i.json 2> $O/bench.err < /dev/null; python -c "
import json; d=json.load(open('$O/bench$i.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"; done
