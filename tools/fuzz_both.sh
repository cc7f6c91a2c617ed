# Fuzz against the CPU oracle with the default library and with the accurate-math build (-DTGS_FAST_MATH=0: libm-grade expf in the forward's
# reference expression, exact divisions, fixed-order backward kernel):  bash tools/fuzz_both.sh [seeds] [scenes per seed]
#   TGS_LIB_NAME=libtgs_raster_accurate.so TGS_DEFINES=-DTGS_FAST_MATH=0 python -m youreditableavatar_amd.build --force   (first; then rebuild the default)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fuzz
mkdir -p $O
S=${1:-10}; N=${2:-32}
for seed in $(seq 1 $S); do
  python $R/tests/tools/fuzz_vs_oracle.py $seed $N >> $O/fast.log 2>&1
  TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_accurate.so python $R/tests/tools/fuzz_vs_oracle.py $seed $N >> $O/accurate.log 2>&1
done
for v in fast accurate; do echo "$v: scenes $(grep -c ' ok \| FAIL ' $O/$v.log), failures $(grep -c ' FAIL ' $O/$v.log)"; grep ' FAIL ' $O/$v.log | cut -c1-400; done | tee $O/summary.txt
