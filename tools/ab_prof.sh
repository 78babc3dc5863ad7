cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-nobal base}; do
  GSR_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o out --output-format csv -- python $GRAFT_REPO_ROOT/tools/exp_chain.py $v 300 bwd > /tmp/log_$v 2>&1
  echo "== $v"; f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1); grep -E "k_blend_bwd|k_preprocess_bwd" $f | cut -d, -f1-4
done
