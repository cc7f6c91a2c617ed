R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_k; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --output-format csv -d $O/rp -o rp -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu --no-secondary > $O/bench.json 2> $O/rp.err < /dev/null
python3 $R/tools/step_timeline.py $O/rp 8 | tee $O/step_timeline.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
