#!/bin/bash
# Builds the cross-check harness OUTSIDE the repository (default /tmp/tgs_emu): the reference's
# kernel text is extracted by line range into that directory and never enters the repo or the GPU box.
set -euo pipefail
REF=${REF:-/root/reference/Edit_core/thirdparties/diff-gaussian-rasterization}
OUT=${OUT:-/tmp/tgs_emu}
HERE=$(cd "$(dirname "$0")" && pwd)
CR=$REF/cuda_rasterizer
mkdir -p "$OUT"
sed -n '18,374p'  "$CR/forward.cu"          > "$OUT/ref_fwd.inc"    # device fns + preprocessCUDA + renderCUDA
sed -n '18,557p'  "$CR/backward.cu"         > "$OUT/ref_bwd.inc"
sed -n '33,138p'  "$CR/rasterizer_impl.cu"  > "$OUT/ref_impl.inc"   # getHigherMsb, checkFrustum, duplicateWithKeys, identifyTileRanges
for v in fma nofma; do
  if [ $v = fma ]; then FL="-mfma -ffp-contract=fast"; else FL="-ffp-contract=off"; fi
  g++ -std=c++20 -O2 -fPIC -shared -pthread $FL -w -I"$OUT" -I"$HERE" -I"$CR" -I"$REF/third_party/glm" \
      "$HERE/driver.cpp" -o "$OUT/libtgs_emu_$v.so"
done
ls -la "$OUT"
