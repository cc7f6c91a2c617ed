# round 6: the miss of soak E (seed 207, scene 90): which oracle build reproduces the HIP result?  and the pixel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_miss207; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
echo > $O/diag2.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee -a $O/diag2.txt
import numpy as np, sys
sys.path.insert(0, '.')
from tests import fuzz, util, adjudicate
rng = np.random.default_rng(207)
for it in range(91):
    desc, inp, dL = fuzz.random_scene(rng, it)
mine = util.hip_run(inp, dL)
det = util.hip_run(inp, dL, deterministic=True)
print('scene', desc)
print('default vs fixed-order backward', {k: f"{util.rel_l2(mine[k], det[k]):.3e}" for k in ('dL_dconic', 'dL_dmeans2D', 'dL_dopacity')})
for variant in ('f32', 'f32_in', 'f32_out', 'f32_fma', 'f32_ex2', 'f64'):
    o = adjudicate.oracle_variant(inp, dL, variant)
    print(variant, {k: f"{util.rel_l2(np.asarray(mine[k]).reshape(np.asarray(o[k]).shape), o[k]):.3e}" for k in ('dL_dconic', 'dL_dmeans2D', 'dL_dopacity', 'dL_drotations', 'color')})
PY
