#!/bin/bash
# Measurement aid (GPU box): rocprofv3 kernel trace of tools/exp_chain.py with the config-4 training step (EXP_CFG4=1: 3 views x
# 131 072 Gaussians, colour + depth); per kernel and GRID size: calls and average duration (the 3-view launches have 3x the grid)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cfg4
EXP_CFG4=1 GSR_LIB_PATH=${GSR_LIB_PATH:-} timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_cfg4 -o out --output-format csv -- python $GRAFT_REPO_ROOT/tools/exp_chain.py cfg4 60 bwd > /tmp/log_cfg4 2>&1
f=$(find /tmp/prof_cfg4 -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].replace("void gsr::", "").replace("(gsr::Params)", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    a = acc[k]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if n >= 20: print(f"{k[0][:60]:60s} grid {k[1]:>8s} x {k[2]:>3s}  calls {n:4d}  avg {t / n:8.2f} us")
PY
tail -1 /tmp/log_cfg4 | sed "s/image sum.*max_list.: [0-9]*}//"
