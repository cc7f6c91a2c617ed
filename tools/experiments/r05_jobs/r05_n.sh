R=$GRAFT_REPO_ROOT; cd $R
export STAGE_INFO=1
for rep in 1 2; do
  for v in "0 0" "1 0" "1 1"; do set -- $v
    echo "light=$1 spread=$2 $(TGS_LIGHT_TILES=$1 TGS_LIGHT_SPREAD=$2 timeout 120 python tools/stage_times.py 2>/dev/null < /dev/null | tail -1 | cut -c40-330)"
  done
done
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
TGS_LIGHT_TILES=1 TGS_LIGHT_SPREAD=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -2
