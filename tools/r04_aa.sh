# round 4, step aa: SSIM rows really in flight, k_scan's second pass with its loads in flight: GPU suite + kernel stats (drop-in loop, trainer protocol) + bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_aa; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
for i in 1 2; do echo "bench: $(timeout 200 python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['kernels_ms'])")"; done
echo "dropin $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
echo "trainer sh3 $(timeout 200 python tools/trainer_protocol.py 3 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin.json 2> $O/rp_dropin.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 3 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
