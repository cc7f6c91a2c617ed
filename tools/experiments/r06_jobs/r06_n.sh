# the issue-rate microbenchmark from its current source (the binary the profile sets of rounds 4-6 ran was older than the source: modes 6 / 7 without the scc clobber, no modes 8-14)
mkdir -p gpurun_out/r06_n
cd tools/microbench && timeout 300 hipcc -O3 --offload-arch=gfx950 -w valu_issue_rate.hip -o valu_issue_rate && timeout 400 ./valu_issue_rate > ../../gpurun_out/r06_n/valu_issue_rate.txt 2>&1
tail -50 ../../gpurun_out/r06_n/valu_issue_rate.txt
