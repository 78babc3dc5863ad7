#!/bin/bash
# Measurement aid (GPU box): many-view timings (8 / 6 / 16 / 48 views) once per library variant.  usage: run_mv2.sh names...
for v in "$@"; do
  export GSR_LIB_PATH=$PWD/tools/variants/$v.so
  a=$(timeout 100 python tools/multiview_prof.py 8 2>&1 | tail -1 | cut -c1-58)
  b=$(timeout 100 python tools/multiview_prof.py 6 2>&1 | tail -1 | cut -c1-58)
  c=$(timeout 100 python tools/multiview_prof.py 16 131072 2>&1 | tail -1 | cut -c1-60)
  d=$(timeout 100 python tools/multiview_prof.py 48 2>&1 | tail -1 | cut -c1-60)
  echo "$v | $a | $b | $c | $d"
done
