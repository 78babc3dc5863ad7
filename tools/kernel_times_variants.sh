#!/bin/bash
# Measurement aid (GPU box): per-kernel average durations (rocprofv3 --kernel-trace --stats) of tools/exp_all.py's shapes named in EXP_ONLY,
# once per library variant in tools/variants/.  usage: EXP_ONLY=fwd+bwd kernel_times_variants.sh outdir names...
out=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $out
for v in "$@"; do
  (cd /tmp && GSR_LIB_PATH=$R/tools/variants/$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/prof_$v -o t -- python $R/tools/exp_all.py $v 300 > $R/$out/$v.log 2>&1)
  db=$(find $out/prof_$v -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $out/kernel_stats_$v.md "$v" "rocprofv3 --kernel-trace --stats -- python tools/exp_all.py (EXP_ONLY=$EXP_ONLY)" > /dev/null
  echo "== $v: $(grep -v rocprofv3 $out/$v.log | tail -1)"
  grep "gsr::" $out/kernel_stats_$v.md | head -6
  rm -rf $out/prof_$v
done
