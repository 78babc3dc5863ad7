#!/bin/bash
# Measurement aid (GPU box): multi-view timings with the depth channel once per library variant.  usage: run_mvx.sh names...
for v in "$@"; do
  export GSR_LIB_PATH=$PWD/tools/variants/$v.so
  a=$(timeout 100 python tools/multiview_prof.py 8 300000 20 extra 2>&1 | tail -1 | cut -c1-64)
  b=$(timeout 100 python tools/multiview_prof.py 48 131072 20 extra 2>&1 | tail -1 | cut -c1-66)
  c=$(timeout 100 python tools/multiview_prof.py 3 131072 40 extra 2>&1 | tail -1 | cut -c1-66)
  echo "$v | $a | $b | $c"
done
