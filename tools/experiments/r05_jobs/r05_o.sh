R=$GRAFT_REPO_ROOT; cd $R/tools/microbench
hipcc -O3 -w --offload-arch=gfx950 valu_issue_rate.hip -o valu_issue_rate && timeout 300 ./valu_issue_rate | tee $R/gpurun_out/r05_o_valu_issue_rate.txt
