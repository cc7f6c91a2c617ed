set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/rp4 $O/rp1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp4 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/bench_under_rocprof.json 2> $O/rp4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp1 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-secondary --streams 1 > $O/bench_under_rocprof_one_stream.json 2> $O/rp1.err
rm -rf $O/rp_trainer
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py > $O/trainer_protocol.json 2> $O/rp_trainer.err
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json
cat $O/trainer_protocol.json
