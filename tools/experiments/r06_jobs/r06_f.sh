# round 6, job f: per-stage times with the Gaussians numbered along a Morton curve (the mesh-bound order) against the generator's random order, x1 and x4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_f; mkdir -p $O; cd $R
for sc in 1 4; do
for M in 0 1 0 1; do
  echo "== morton $M x$sc" | tee -a $O/stage_times.txt
  STAGE_MORTON=$M timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
cd /tmp; export TMPDIR=/tmp
STAGE_MORTON=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_morton -o rp -- python3 $R/tools/stage_times.py 1 > $O/rp_morton.log 2>&1
grep "k_" $O/rp_morton/rp_kernel_stats.csv | cut -d, -f1-5 | cut -c1-60,140-200
find $O -name "*kernel_trace.csv" -delete
