# round 4, step l: loads of the per-Gaussian backward issued together (SH staging, batch prologue, rejected flags, rows in groups): GPU suite + A/B
# against the previous commit (libtgs_raster_h.so) and the r04_a sources
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_l; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_h.so libtgs_raster_r04a.so default libtgs_raster_h.so libtgs_raster_r04a.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for i in 1 2; do
  for lib in default h r04a; do
    if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
    echo "dropin $lib $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
    echo "trainer sh3 $lib $(timeout 200 python tools/trainer_protocol.py 3 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
  done
done
unset TGS_LIBRARY
