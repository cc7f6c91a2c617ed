# round 6, job j: re-tuning sweep of compile-time constants after the round's changes (stage times x1 and x4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j; mkdir -p $O; cd $R
for sc in 1 4; do
for L in default xcd0 fq384 fq512 bch320 default; do
  if [ $L = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$L.so; fi
  echo "== $L x$sc $(timeout 300 python tools/stage_times.py $sc 2>&1 | tail -1 | sed 's/.*us per stage//')" | tee -a $O/sweep.txt
done
done
