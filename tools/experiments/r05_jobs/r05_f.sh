R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_f; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for lib in k00 k00009765625; do
  export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so
  timeout 300 python bench.py --no-cpu --no-secondary > $O/bench_$lib.json 2> $O/bench.err < /dev/null; python -c "
import json; d=json.load(open('$O/bench_$lib.json')); print('$lib', d['value'], d['ms_per_step'], d['kernels_ms'])"
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fuzz or cutoff or random_small or autograd" > $O/pytest_$lib.txt 2>&1 < /dev/null; tail -3 $O/pytest_$lib.txt | cut -c1-400
done
