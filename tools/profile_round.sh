set -x
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh [name]      (every step under its own `timeout`: one hung step must not eat the call)
#   -> gpurun_out/<name>/{bench_default.json, rp4/, rp1/, pmc_fetch/, pmc_write/, pmc_sq/, pmc_sq2/, valu_issue_rate.txt}
# Counter passes never share a run with trace domains other than --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC slots: 8 SQ counters or
# FETCH_SIZE (3 TCC slots) or WRITE_SIZE (2) per pass).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r06_f}
mkdir -p $O
python3 -c "import sys; sys.path.insert(0, '$R'); from youreditableavatar_amd.build import source_hash; print(source_hash())" > $O/csrc_sha16.txt   # the sources these counters belong to
cd /tmp && export TMPDIR=/tmp
timeout 420 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp4 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/bench_under_rocprof.json 2> $O/rp4.err
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp1 -o rp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-secondary --streams 1 > $O/bench_under_rocprof_one_stream.json 2> $O/rp1.err
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_write.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_sq.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmc_sq2 -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_sq2.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT --output-format csv -d $O/pmc_sq3 -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_sq3.log 2>&1
# the probes are rebuilt from their sources first: a binary that travelled with the snapshot may be older than its .hip (rounds 4-6 ran a stale one)
(cd $R/tools/microbench && for p in valu_issue_rate fetch_calibration; do timeout 300 hipcc -O3 --offload-arch=gfx950 -w $p.hip -o $p; done) > $O/microbench_build.log 2>&1
(cd $R/tools/microbench && timeout 300 ./valu_issue_rate) > $O/valu_issue_rate.txt 2>&1
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts in this library's access patterns (tools/fetch_calibration.py -> profiles/<name>_fetch_calibration.json)
timeout 200 $R/tools/microbench/fetch_calibration > $O/cal_bytes.txt 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o pmc -- $R/tools/microbench/fetch_calibration > $O/cal_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o pmc -- $R/tools/microbench/fetch_calibration > $O/cal_write.log 2>&1
# drop-in loop and the trainers' protocol under the kernel trace
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py > $O/dropin_under_rocprof.json 2> $O/rp_dropin.err
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer3 -o rp -- python3 $R/tools/trainer_protocol.py 3 40 > $O/trainer_protocol_sh3.json 2> $O/rp_trainer3.err
# round 6: large splats as a first-class workload (kernel stats + FETCH / WRITE at x4 / x8), counters over the STORE-MODE per-Gaussian backward (the drop-in loop),
# and over the batch step (k_preprocess_bwd_batch_split, k_preprocess_fwd_pair)
for sc in 4 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_x$sc -o rp -- python3 $R/tools/stage_times.py $sc > $O/rp_x$sc.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_x$sc -o pmc -- python3 $R/tools/stage_times.py $sc > $O/pmc_fetch_x$sc.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_x$sc -o pmc -- python3 $R/tools/stage_times.py $sc > $O/pmc_write_x$sc.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_fetch_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_write_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_sq_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_lds_dropin -o pmc -- python3 $R/tools/dropin_loop.py 8 > $O/pmc_lds_dropin.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_batch -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --streams 1 > $O/pmc_fetch_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_batch -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --streams 1 > $O/pmc_write_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq_batch -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --streams 1 > $O/pmc_sq_batch.log 2>&1
timeout 300 python3 $R/tools/batch_stage_times.py 1 0 > $O/batch_stage_times.txt 2>/dev/null
timeout 300 python3 $R/tools/batch_stage_times.py 1 1 >> $O/batch_stage_times.txt 2>/dev/null
timeout 300 python3 $R/tools/batch_stage_times.py 4 0 >> $O/batch_stage_times.txt 2>/dev/null
timeout 300 python3 $R/tools/batch_stage_times.py 4 1 >> $O/batch_stage_times.txt 2>/dev/null
find $O -name "*counter_collection.csv" -size +6M -delete 2>/dev/null
# the big per-dispatch traces are not needed once the stats exist (gpurun merges at most 64 MiB back)
find $O -name "*kernel_trace.csv" -path "*rp*" -delete 2>/dev/null
ls -R $O | head -60
tail -c 600 $O/bench_default.json
# other configurations and the trainers' protocol (rocprof summaries only)
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_cfg2 -o rp -- python3 $R/bench.py --config 2 --steps 6 --warmup 2 --no-cpu --no-secondary > $O/bench_cfg2_under_rocprof.json 2> $O/rp_cfg2.err
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_cfg5 -o rp -- python3 $R/bench.py --config 5 --views-per-gpu 4 --steps 4 --warmup 2 --no-cpu --no-secondary > $O/bench_cfg5_under_rocprof.json 2> $O/rp_cfg5.err
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer_protocol.json 2> $O/rp_trainer.err
timeout 420 python3 $R/bench.py --config 2 --no-cpu --no-secondary > $O/bench_cfg2.json 2>/dev/null
timeout 420 python3 $R/bench.py --config 5 --views-per-gpu 4 --no-cpu --no-secondary > $O/bench_cfg5.json 2>/dev/null
du -sh $O; find $O -size +8M | head
