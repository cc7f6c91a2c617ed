# round 4, step t: k_scan's length histogram with one counter per bucket and lane: GPU suite + kernel stats of drop-in loop and trainer protocol
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_t; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_l.so default" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin.json 2> $O/rp_dropin.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 3 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
cat $O/dropin.json $O/trainer.json | cut -c1-100
