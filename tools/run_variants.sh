#!/bin/bash
# Measurement aid (GPU box): tools/exp_chain.py once per library variant in tools/variants/.  usage: run_variants.sh [steps] [bwd] names...
steps=${1:-300}; bwd=$2; shift 2
for v in "$@"; do
  GSR_LIB_PATH=$PWD/tools/variants/$v.so timeout 120 python tools/exp_chain.py $v $steps $bwd 2>&1 | tail -1
done
