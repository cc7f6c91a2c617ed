# round 4, step ac: k_render_bwd asks for the next round's records BEHIND the barrier (a __syncthreads() waits for outstanding loads): GPU suite + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ac; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -3 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "libtgs_raster_prev.so default libtgs_raster_prev.so default libtgs_raster_prev.so default" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for lib in prev default prev default; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "cfg5 $lib: $(timeout 300 python bench.py --config 5 --views-per-gpu 4 --no-cpu --no-secondary --steps 12 --warmup 3 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'])")"
  echo "trainer sh0 $lib $(timeout 200 python tools/trainer_protocol.py 0 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
done
unset TGS_LIBRARY
