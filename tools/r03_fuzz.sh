R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_fuzz; mkdir -p $O; cd $R
for seed in 11 12 13 14; do
  timeout 900 python tests/tools/fuzz_vs_oracle.py $seed 96 1.0 1 > $O/light_$seed.txt 2>&1; tail -2 $O/light_$seed.txt | head -1
  timeout 900 python tests/tools/fuzz_vs_oracle.py $seed 96 1.0 0 > $O/plain_$seed.txt 2>&1; tail -2 $O/plain_$seed.txt | head -1
done
grep -h MISS $O/*.txt | cut -c1-260 | head -40
