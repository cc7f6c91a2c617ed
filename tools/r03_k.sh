R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_k; mkdir -p $O; cd $R
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps2.so timeout 600 python tests/tools/timeline.py phases > $O/phases.txt 2>&1
cat $O/phases.txt
