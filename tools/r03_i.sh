# wave-level utilisation inside the render workgroups (stamps build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_i; mkdir -p $O; cd $R
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_stamps.so timeout 600 python tests/tools/timeline.py > $O/timeline.txt 2>&1
cat $O/timeline.txt
