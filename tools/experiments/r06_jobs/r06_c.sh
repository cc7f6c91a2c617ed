# round 6, job c: level-major dL_dsh (parity of the batch pass, RCCL at world size 1 without staging), padded SH rows in k_preprocess_bwd (A/B against -DTGS_SH_ROW=12)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_c; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -m gpu -x -q --timeout 700 -k "level_major or run_views or batched or live_sh or range_wise" > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt | cut -c1-400
for L in sh12 default sh12 default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L" | tee -a $O/stage_times.txt
  timeout 300 python tools/stage_times.py 1 2>&1 | tail -1 | tee -a $O/stage_times.txt
done
unset TGS_LIBRARY
timeout 400 python tools/rccl_world1.py > $O/rccl_world1.json 2> $O/rccl_world1.err; tail -c 3000 $O/rccl_world1.json
