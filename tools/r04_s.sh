# round 4, step s: k_render_bwd asks for the first round's records with the pixel state; bind kernels load-all / store-all; SH colour kernels' 12-B inputs up front
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_s; mkdir -p $O; cd $R
timeout 60 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt | cut -c1-300
timeout 900 bash tools/libs.sh "default libtgs_raster_l.so default libtgs_raster_l.so" > $O/ab.txt 2>&1 < /dev/null; cat $O/ab.txt
for i in 1 2; do
  for lib in default l; do
    if [ $lib = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_$lib.so; fi
    echo "dropin $lib $(timeout 120 python tools/dropin_loop.py 300 2>/dev/null < /dev/null | tail -1)"
    echo "trainer sh3 $lib $(timeout 200 python tools/trainer_protocol.py 3 60 2>/dev/null < /dev/null | tail -1 | cut -c1-60)"
  done
done
cd /tmp && export TMPDIR=/tmp
unset TGS_LIBRARY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_dropin -o rp -- python3 $R/tools/dropin_loop.py 80 > $O/dropin.json 2> $O/rp_dropin.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_trainer -o rp -- python3 $R/tools/trainer_protocol.py 3 40 > $O/trainer.json 2> $O/rp_trainer.err < /dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
