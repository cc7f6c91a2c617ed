# round 6, job i: binning chunks 128 / 256 (default now) / 512 (builds with -DTGS_BIN_WGS_MAX), per kernel under the kernel trace, x1 and x4; then parity of the binning
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { # name lib scale
  if [ "$2" = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/$2; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_$1 -o rp -- python3 $R/tools/stage_times.py $3 > $O/rp_$1.log 2>&1
  echo "== $1 x$3: $(python3 - <<PY
import csv
out=[]; tot=0
for r in csv.DictReader(open('$O/rp_$1/rp_kernel_stats.csv')):
    n=r['Name'].replace('void ','').split('(')[0].replace('tgs::','')
    if n in ('k_bin_count','k_bin_colscan','k_scan','k_scatter'): out.append(f"{n} {float(r['AverageNs'])/1e3:.1f}"); tot+=float(r['AverageNs'])/1e3
print('  '.join(sorted(out)), ' sum %.1f' % tot)
PY
)" | tee -a $O/chunks2.txt
}
for sc in 1 4; do
run w128_$sc libtgs_raster_w128.so $sc
run w256_$sc default $sc
run w512_$sc libtgs_raster_w512.so $sc
run w256b_$sc default $sc
done
find $O -name "*kernel_trace.csv" -delete
unset TGS_LIBRARY; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q --timeout 600 -k "seeded or golden or pruning or grid_beyond or overflow or large or run_views" 2>&1 | tail -3 | cut -c1-300
