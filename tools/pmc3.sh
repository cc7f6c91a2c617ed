R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_c; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT --output-format csv -d $O/pmc_sq3 -o pmc -- python3 $R/tools/pmc_workload.py 4 > $O/pmc_sq3.log 2>&1
tail -2 $O/pmc_sq3.log; ls $O/pmc_sq3
