# round 6, job g: the 8-view batch step by stage, random order against Morton order, x1 and x4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_g; mkdir -p $O; cd $R
for sc in 1 4; do for M in 0 1; do timeout 300 python tools/batch_stage_times.py $sc $M 2>&1 | tail -1 | tee -a $O/batch_stage_times.txt; done; done
