# round 6, job h: drop-in loop and trainers' protocol with / without the backward outputs the caller discards (TGS_KEEP_ALL_OUTPUTS=1: round 5's behaviour)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q --timeout 600 -k "discards or training_style or level_major" 2>&1 | tail -3 | cut -c1-300
for K in 1 0 1 0; do
  echo "== keep_all $K: dropin $(TGS_KEEP_ALL_OUTPUTS=$K timeout 300 python tools/dropin_loop.py 200 2>/dev/null | tail -1)  trainer sh0 $(TGS_KEEP_ALL_OUTPUTS=$K timeout 300 python tools/trainer_protocol.py 0 40 2>/dev/null | tail -1 | cut -c1-200)" | tee -a $O/keep_all.txt
done
cd /tmp; export TMPDIR=/tmp
for K in 1 0; do
TGS_KEEP_ALL_OUTPUTS=$K timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin_keep$K -o rp -- python3 $R/tools/dropin_loop.py > $O/rp$K.log 2>&1
echo "keep_all $K: $(grep 'k_preprocess_bwd' $O/rp_dropin_keep$K/rp_kernel_stats.csv | cut -d, -f2-4)" | tee -a $O/keep_all.txt
done
find $O -name "*kernel_trace.csv" -delete
