# round 6, job b: wave-spread tile_reachable in k_preprocess_fwd (parity + stage times x1 / x4 / x8 against the round-5 library), kernel stats and counters of the
# large-splat frames (x4 / x8), and counters over the STORE-MODE launch of k_preprocess_bwd (the drop-in loop)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_b; mkdir -p $O; cd $R
python3 -c "import sys; sys.path.insert(0, '$R'); from youreditableavatar_amd.build import source_hash; print(source_hash())" > $O/csrc_sha16.txt
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q --timeout 600 -k "seeded or large or pruning or golden or spatially or nonfinite" > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt | cut -c1-300
for sc in 1 4 8; do
for L in r05 default r05 default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
done
unset TGS_LIBRARY
cd /tmp && export TMPDIR=/tmp
for sc in 4 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_x$sc -o rp -- python3 $R/tools/stage_times.py $sc > $O/rp_x$sc.log 2>&1
done
for sc in 4 8; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_x$sc -o pmc -- python3 $R/tools/stage_times.py $sc > $O/pmc_fetch_x$sc.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_x$sc -o pmc -- python3 $R/tools/stage_times.py $sc > $O/pmc_write_x$sc.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq_x4 -o pmc -- python3 $R/tools/stage_times.py 4 > $O/pmc_sq_x4.log 2>&1
# store-mode k_preprocess_bwd (and the whole drop-in frame): FETCH / WRITE / SQ / LDS passes over the drop-in loop
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_fetch_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_write_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_sq_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_lds_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_lds_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py > $O/dropin_under_rocprof.json 2> $O/rp_dropin.err
for d in pmc_fetch_x4 pmc_write_x4 pmc_fetch_x8 pmc_write_x8 pmc_sq_x4 pmc_fetch_dropin pmc_write_dropin pmc_sq_dropin pmc_lds_dropin; do echo "== $d"; python3 $R/tools/pmc_summary.py $O/$d; done > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +6M -delete 2>/dev/null
du -sh $O; tail -30 $O/pmc_summary.txt
