# repeat the 2-rank bench (gloo, one device); on a hang dump the ranks' Python stacks
for i in $(seq ${1:-8}); do
  python tools/debug_cmd_tree.py 60 python bench.py --gpus 2 --backend gloo --same-device --config 2 --steps 2 --warmup 2 --views-per-gpu 4 --no-cpu > gpurun_out/b2_$i.out 2> gpurun_out/b2_$i.err
  rc=$?
  echo "run $i rc=$rc $(grep -c HUNG gpurun_out/b2_$i.out)"
  if [ $rc -ne 0 ]; then grep -v "^\[W\|amdgpu.ids\|OMP_NUM\|^\*\*\*\|^$" gpurun_out/b2_$i.err | tail -80 | cut -c1-220; cat gpurun_out/b2_$i.out | head -20 | cut -c1-220; break; fi
done
