#!/bin/bash
# Measurement aid (GPU box): tools/bwd_shapes_prof.py at several shapes once per library variant.  usage: run_bs.sh names...
for v in "$@"; do
  export GSR_LIB_PATH=$PWD/tools/variants/$v.so
  a=$(timeout 100 python tools/bwd_shapes_prof.py 256 8 2>&1 | tail -1 | cut -c30-90)
  b=$(timeout 100 python tools/bwd_shapes_prof.py 256 3 131072 40 2>&1 | tail -1 | cut -c30-90)
  c=$(timeout 100 python tools/bwd_shapes_prof.py 256 1 300000 100 2>&1 | tail -1 | cut -c30-90)
  d=$(timeout 100 python tools/bwd_shapes_prof.py 256 2 300000 40 2>&1 | tail -1 | cut -c30-90)
  echo "$v | V8: $a | V3/131k: $b | V1: $c | V2: $d"
done
