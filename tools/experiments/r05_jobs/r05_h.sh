# round 5, job h: speculation policy (decay 0.999) + 256 binning chunks variant: drop-in loop, trainer protocol, stage times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_h; mkdir -p $O; cd $R
for rep in 1 2 3; do echo "dropin $(timeout 120 python tools/dropin_loop.py 400 2>/dev/null < /dev/null | tail -1)"; done
for rep in 1 2; do echo "trainer sh0 $(timeout 200 python tools/trainer_protocol.py 0 80 2>/dev/null < /dev/null | tail -1 | cut -c1-80)"; done
export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_b256.so
for rep in 1 2; do echo "b256 $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"; done
for rep in 1 2; do echo "b256 dropin $(timeout 120 python tools/dropin_loop.py 400 2>/dev/null < /dev/null | tail -1)"; done
unset TGS_LIBRARY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin.json 2> $O/rp_dropin.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null; grep -i "copyBuffer\|k_scan\|k_scatter" $O/rp_dropin/rp_kernel_stats.csv | cut -c1-60,150-260
