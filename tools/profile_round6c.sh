#!/bin/bash
# GPU box: round 6, third call - the set binning launch (k_preprocess_bin_set).
L=${1:-r06_c}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20
export GSR_KEEP_LIB=1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 timeout 300 python tools/phase_stamps.py "configs[3] forward, random scene" > $O/stamps_config4.txt 2>&1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "configs[3] forward, pixel-aligned scene" > $O/stamps_config4_structured.txt 2>&1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 PS_TRAIN=1 timeout 300 python tools/phase_stamps.py "configs[3] training forward, random scene" > $O/stamps_config4_train.txt 2>&1
for f in $O/stamps_*.txt; do echo "=== $f"; grep -v amdgpu.ids $f | cut -c1-420 | head -12; done
unset GSR_KEEP_LIB
prof() {  # prof <name> <title> <bench.py args...>
  local name=$1 title=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o t -- python $R/bench.py "$@" > $R/$O/prof_$name.json 2> /dev/null)
  local db=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_$name.md "$L: $title" "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null
  sed -n 7,12p $O/kernel_stats_$name.md | cut -c1-160
}
prof config4_fwd "BASELINE configs[3] forward: 3 views x 131 072 Gaussians, colour + depth, 150 calls through the plan API" --traffic-child cfg4_fwd
prof config4_train "BASELINE configs[3] training step" --traffic-child cfg4_train
prof config4s_fwd "configs[3] forward on the pixel-aligned scene" --traffic-child cfg4s_fwd
prof config4s_train "configs[3] training step on the pixel-aligned scene" --traffic-child cfg4s_train
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
