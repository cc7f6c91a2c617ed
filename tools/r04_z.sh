# the two scenes of seed 23 that miss the bar: which build shows them?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_z; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for lib in default r04a accurate; do
  if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
  timeout 900 python -m tests.adjudicate 23 96 > $O/seed23_$lib.txt 2>&1 < /dev/null
  echo "== $lib"; grep "MISS" $O/seed23_$lib.txt | cut -c1-200; tail -1 $O/seed23_$lib.txt | cut -c1-160
done
