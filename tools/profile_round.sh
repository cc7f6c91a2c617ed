set -x
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh [name]  -> gpurun_out/<name>/{bench_default.json, rp4/, rp1/, pmc_fetch/, pmc_write/}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r01_j}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp4 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu > $O/bench_under_rocprof.json 2> $O/rp4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp1 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --streams 1 > $O/bench_under_rocprof_one_stream.json 2> $O/rp1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_write.log 2>&1
ls -R $O | head -40
tail -c 600 $O/bench_default.json
