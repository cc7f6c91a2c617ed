# round 6, job u: the scattered-store rate of the GPU outside the library (tools/microbench/scatter_store_rate.hip)
mkdir -p gpurun_out/r06_u; cd tools/microbench && timeout 300 hipcc -O3 --offload-arch=gfx950 -w scatter_store_rate.hip -o scatter_store_rate && timeout 300 ./scatter_store_rate > ../../gpurun_out/r06_u/scatter_store_rate.txt 2>&1; cat ../../gpurun_out/r06_u/scatter_store_rate.txt
