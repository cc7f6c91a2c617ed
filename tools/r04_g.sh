# round 4, step g: the gather fused into the tile sort (k_finalize gone): GPU suite, A/B against the unfused build, kernel stats of the trainer protocol
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_g; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -6 $O/pytest.txt | cut -c1-400
timeout 600 bash tools/libs.sh "default libtgs_raster_unfused.so default libtgs_raster_unfused.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for i in 1 2; do
  for lib in default unfused; do
    if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
    echo "dropin $lib $(timeout 120 python tools/dropin_loop.py 200 2>/dev/null < /dev/null | tail -1)"
    echo "trainer sh0 $lib $(timeout 200 python tools/trainer_protocol.py 0 60 2>/dev/null < /dev/null | tail -1 | cut -c1-100)"
  done
done
unset TGS_LIBRARY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 0 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin.json 2> $O/rp_dropin.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -14 $f | cut -c1-140; done
