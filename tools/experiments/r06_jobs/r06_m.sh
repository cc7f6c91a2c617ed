# round 6, job m: k_preprocess_fwd_pair with the two views on the two halves of a 512-thread workgroup -- parity of the batch paths, batch stage times, bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_m; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -m gpu -x -q --timeout 700 -k "run_views or batch or views or multirank or level_major or benchmark_configuration" 2>&1 | tail -3 | cut -c1-300 | tee $O/pytest.txt
for i in 1 2; do timeout 300 python tools/batch_stage_times.py 1 0 2>&1 | tail -1 | tee -a $O/batch.txt; done
timeout 300 python tools/batch_stage_times.py 4 0 2>&1 | tail -1 | tee -a $O/batch.txt
timeout 600 python bench.py --no-cpu --no-secondary > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['kernels_ms'])" | tee -a $O/batch.txt
