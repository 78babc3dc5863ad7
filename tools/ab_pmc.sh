#!/bin/bash
# Measurement aid (GPU box): L2 / fabric counters of k_blend_bwd for library variants.  usage: VARIANTS="a b" bash tools/ab_pmc.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "TCC_[A-Z0-9_]*(ATOMIC|RDREQ|WRREQ|HIT|MISS|REQ|WRITEBACK|EA0_RD|EA0_WR)[A-Z0-9_]*" | sort -u | tr '\n' ' ' | cut -c1-3000; echo
for v in ${VARIANTS:-base}; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_WRITE_sum TCC_WRITEBACK_sum"; do
    rm -rf /tmp/pmc_$v
    GSR_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$v -o t -- python $GRAFT_REPO_ROOT/tools/exp_chain.py $v 100 bwd > /tmp/pmclog 2>&1
    f=$(find /tmp/pmc_$v -name '*counter_collection.csv' | head -1)
    [ -z "$f" ] && { echo "$v: no counters for $set"; tail -3 /tmp/pmclog; continue; }
    python - "$f" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_blend_bwd" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[2], {k: round(v[0] / max(1, v[1])) for k, v in acc.items()})
PY
  done
done
