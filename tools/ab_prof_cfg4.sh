#!/bin/bash
# GPU box: per-kernel averages (rocprofv3) of the configs[3] training step for library variants.  usage: VARIANTS="base x" bash tools/ab_prof_cfg4.sh [traffic-child mode = cfg4_train]
cd /tmp && export TMPDIR=/tmp
w=${1:-cfg4_train}
for v in ${VARIANTS:-base}; do
  rm -rf /tmp/prof_${v}_$w
  GSR_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v.so timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_${v}_$w -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --traffic-child $w > /tmp/log_${v}_$w 2>&1
  echo "== $v $w"; f=$(find /tmp/prof_${v}_$w -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Calls"].isdigit() and int(r["Calls"]) >= 100:
        print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.2f} us')
PY
done
