#!/bin/bash
# Measurement aid (GPU box): multi-view timings once per library variant.  usage: run_mv.sh names...
for v in "$@"; do
  export GSR_LIB_PATH=$PWD/tools/variants/$v.so
  a=$(timeout 100 python tools/multiview_prof.py 8 2>&1 | tail -1 | cut -c1-58)
  b=$(timeout 100 python tools/multiview_prof.py 48 2>&1 | tail -1 | cut -c1-60)
  c=$(timeout 100 python tools/multiview_prof.py 3 131072 40 2>&1 | tail -1 | cut -c1-60)
  echo "$v | $a | $b | $c"
done
