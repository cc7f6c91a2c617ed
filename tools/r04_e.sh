# round 4, step e: find what stalled step d -- every command under its own timeout, one small probe per suspect
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e; mkdir -p $O; cd $R
probe() { name=$1; shift; timeout 120 "$@" > $O/$name.txt 2>&1; echo "$name rc=$? $(tail -1 $O/$name.txt | cut -c1-200)"; }
cat > /tmp/p1.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from youreditableavatar_amd import scenes
from tests import util
cloud = scenes.make_cloud(3000, 3, seed=42, scale_mult=3.0)
cam = scenes.orbit_camera(160, 128, azimuth_deg=15.0)
inp = util.scene_input(cloud, cam)
dL = scenes.upstream_gradient(160, 128)
side = None if sys.argv[1] == "none" else bool(int(sys.argv[1]))
for i in range(3):
    out = util.hip_run(inp, dL, side_stream=side)
print("ok", sys.argv[1], float(out["color"].sum()), int(out["num_rendered"]))
PY
TGS_SIDE_STREAM=0 probe side_env_off python /tmp/p1.py none
probe side_off python /tmp/p1.py 0
probe side_on python /tmp/p1.py 1
probe side_default python /tmp/p1.py none
probe smoke python -c "import __graft_entry__ as g; g.smoke()"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 200 > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt | cut -c1-300
