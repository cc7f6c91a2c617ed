R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
echo "== deterministic backward (k_render_bwd_det: fixed order, accurate exp / divisions)"; TGS_DETERMINISTIC=1 timeout 900 python tests/tools/fuzz_vs_oracle.py 94 96 2>&1 | grep "MISS\|failures" | cut -c1-300
echo "== light tiles forced on"; timeout 900 python tests/tools/fuzz_vs_oracle.py 94 96 1 2>&1 | grep "MISS\|failures" | cut -c1-300
echo "== light tiles forced off"; timeout 900 python tests/tools/fuzz_vs_oracle.py 94 96 0 2>&1 | grep "MISS\|failures" | cut -c1-300
