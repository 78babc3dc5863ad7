#!/bin/bash
# Runs on the GPU box (through gpurun): GPU test suite, the bench line, a rocprofv3 kernel trace of the bench and the summaries
# that get copied into profiles/.  usage: bash tools/profile_round.sh <label>   (writes gpurun_out/<label>/)
L=${1:-r02}
O=gpurun_out/$L
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -1 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
CMD="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> /dev/null)
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/kernel_stats.md "$L: kernel statistics" "rocprofv3 --kernel-trace --stats -- $CMD" > /dev/null
python tools/rocprof_timeline.py $DB 40 | grep -v columns > $O/timeline.txt
cat $O/timeline.txt
timeout 200 python tools/exp_chain.py $L 300 bwd 2>/dev/null | grep -v amdgpu
