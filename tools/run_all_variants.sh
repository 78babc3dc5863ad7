#!/bin/bash
# Measurement aid (GPU box): tools/exp_all.py once per library variant in tools/variants/ (twice round-robin: box drift shows).  usage: run_all_variants.sh names...
for rep in 1 2; do
for v in "$@"; do
  GSR_LIB_PATH=$PWD/tools/variants/$v.so timeout 300 python tools/exp_all.py $v 300 2>&1 | tail -1
done
done
