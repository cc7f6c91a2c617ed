# round 6: the miss of soak B (seed 104, scene 25) -- with the current library, the round-5 library, the fixed-order accurate-math backward, and light tiles forced on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_miss104; mkdir -p $O; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
echo "== current library" | tee $O/diag.txt
timeout 300 python tests/tools/fuzz_diagnose.py 104 25 2>&1 | tail -22 | tee -a $O/diag.txt
echo "== round-5 library (libtgs_raster_r05.so = HEAD of round 5)" | tee -a $O/diag.txt
TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_r05.so timeout 300 python tests/tools/fuzz_diagnose.py 104 25 2>&1 | tail -22 | tee -a $O/diag.txt
echo "== TGS_DETERMINISTIC=1 (k_render_bwd_det: the reference's order, accurate exp / division)" | tee -a $O/diag.txt
TGS_DETERMINISTIC=1 timeout 300 python tests/tools/fuzz_diagnose.py 104 25 2>&1 | tail -22 | tee -a $O/diag.txt
python - <<'PY' 2>&1 | tee -a $O/diag.txt
import numpy as np, sys
sys.path.insert(0, '.')
from tests import fuzz, util, adjudicate
rng = np.random.default_rng(104)
for it in range(26):
    desc, inp, dL = fuzz.random_scene(rng, it)
ref = util.oracle_run(inp, dL)
mine = util.hip_run(inp, dL)
noise = util.reference_noise_of(ref) if hasattr(util, 'reference_noise_of') else None
print('scene', desc)
for variant in ('f32_in', 'f32_out', 'f32_fma', 'f32_ex2', 'f64'):
    try:
        o = adjudicate.oracle_variant(inp, dL, variant)
        print(variant, {k: f"{util.rel_l2(np.asarray(mine[k]).reshape(np.asarray(o[k]).shape), o[k]):.3e}" for k in ('dL_dconic', 'dL_dmeans2D', 'dL_dopacity', 'color')})
    except Exception as ex:
        print(variant, 'error', repr(ex)[:200])
PY
