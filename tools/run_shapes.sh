for rep in 1 2; do for v in cur short; do GSR_LIB_PATH=$PWD/tools/variants/$v.so timeout 300 python tools/exp_shapes.py $v 2>&1 | tail -1; done; done
