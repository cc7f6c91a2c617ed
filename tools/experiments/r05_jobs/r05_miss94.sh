R=$GRAFT_REPO_ROOT; cd $R
timeout 120 python -c "from oracle import oracle; oracle.build(force=True)" < /dev/null
for c in 83ab7fc e148983; do
echo "== library at $c"; TGS_LIBRARY=$R/youreditableavatar_amd/lib/libtgs_raster_at_$c.so timeout 900 python tests/tools/fuzz_vs_oracle.py 94 96 2>&1 | grep "MISS\|failures" | cut -c1-300
done
