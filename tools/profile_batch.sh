#!/bin/bash
# GPU box: kernel stats of PF3plat's training batch shape (tools/exp_batch.py, B = 4 scenes x 3 views x 131 072).  usage: profile_batch.sh [label]
L=${1:-r06_w}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
for st in random pixel_aligned; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_batch4_$st -o t -- python $R/tools/exp_batch.py $st 4 > $R/$O/batch4_$st.txt 2> /dev/null)
  db=$(find $O/prof_batch4_$st -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_batch4_$st.md "$L: training batch, 4 scenes x 3 views x 131 072 Gaussians, colour + depth ($st scenes): forward loop, then training-step loop" "rocprofv3 --kernel-trace --stats -- python tools/exp_batch.py $st 4" > /dev/null
  cat $O/batch4_$st.txt | grep -v amdgpu
  sed -n 7,16p $O/kernel_stats_batch4_$st.md | cut -c1-170
done
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
