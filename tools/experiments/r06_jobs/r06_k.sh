# round 6, job k: one instruction sequence for the pair power in k_render_fwd and k_render_bwd (pair_power2) + the early global atomic in k_scan: the recorded misses, parity, stage times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_k; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "known_misses" -rxX 2>&1 | tail -8 | cut -c1-300 | tee $O/known.txt
timeout 300 python tests/tools/fuzz_diagnose.py 104 25 2>&1 | grep -v amdgpu | grep " rel \|scene" | tee -a $O/known.txt
timeout 300 python tests/tools/fuzz_diagnose.py 94 71 2>&1 | grep -v amdgpu | grep " rel \|scene" | tee -a $O/known.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q --timeout 600 --deselect tests/test_gpu_parity.py::test_known_misses_are_still_the_recorded_ones 2>&1 | tail -4 | cut -c1-300 | tee $O/pytest.txt
for L in r05 default r05 default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L $(timeout 300 python tools/stage_times.py 1 2>&1 | tail -1 | sed 's/.*us per stage//')" | tee -a $O/stage_times.txt
done
