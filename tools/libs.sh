# bench with each of the given library variants: tools/libs.sh "default lib1.so lib2.so" [bench args]
L=$1; shift
for x in $L; do
  if [ $x = default ]; then unset TGS_LIBRARY; else export TGS_LIBRARY=youreditableavatar_amd/lib/$x; fi
  python bench.py --no-cpu --no-secondary --steps 20 --warmup 4 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$x', d['config']['ms_per_frame_per_gpu'], ' '.join(f'{n}={k[n]}' for n in k))"
done
