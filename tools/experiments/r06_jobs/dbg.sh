cd $GRAFT_REPO_ROOT; timeout 600 python scratch/dbg_px.py 2>&1 | grep -v amdgpu | tail -60 | cut -c1-220
