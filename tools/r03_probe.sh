# round 3, first look: new API tests, render-kernel timelines (stamps build), the N > 1 rehearsal in a loop, baseline bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_probe; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -x -q -m gpu -k "speculative or two_threads or rccl or two_rank or sync_free_forward or training_style" > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps.so python tests/tools/timeline.py > $O/timeline.txt 2>&1
tail -30 $O/timeline.txt
python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
python tools/mr_loop.py $O/mr_plain 25 20 > $O/mr_plain.txt 2>&1; tail -3 $O/mr_plain.txt
python tools/mr_loop.py $O/mr_holder 25 20 --holder > $O/mr_holder.txt 2>&1; tail -3 $O/mr_holder.txt
