#!/bin/bash
# Measurement aid (GPU box): tools/large_image_prof.py once per library variant.  usage: run_li.sh side names...
side=$1; shift
for v in "$@"; do
  echo -n "$v: "; GSR_LIB_PATH=$PWD/tools/variants/$v.so timeout 120 python tools/large_image_prof.py $side 2>&1 | tail -1
done
