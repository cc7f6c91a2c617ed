# round 5, job a: cost decomposition of k_render_bwd / k_render_fwd (timing-only variants built from tools/experiments/r05_render_decomposition.patch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_a; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in default e1 e2 e3 e4 e5 f3 f4; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  echo "$lib $(timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1)"
done
done > $O/stage_times.txt 2>&1
cat $O/stage_times.txt
export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps.so
timeout 200 python tests/tools/timeline.py > $O/timeline.txt 2>&1 < /dev/null
unset TGS_LIBRARY
tail -40 $O/timeline.txt
timeout 300 python bench.py --no-cpu > $O/bench_default.json 2> $O/bench.err < /dev/null; tail -c 1500 $O/bench_default.json
