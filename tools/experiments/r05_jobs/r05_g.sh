# round 5, job g: k_scan over several workgroups: GPU suite subset + stage times + drop-in loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_g; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt | cut -c1-400
for rep in 1 2 3; do echo "$(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"; done > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
for rep in 1 2; do echo "dropin $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin.json 2> $O/rp_dropin.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null; find $O -name "*kernel_stats.csv" | head -2 | xargs -I{} sh -c 'head -16 {} | cut -c1-150'
