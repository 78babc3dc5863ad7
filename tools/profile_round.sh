#!/bin/bash
# Runs on the GPU box (through gpurun): the round's evidence in one call.  usage: bash tools/profile_round.sh <label>   (writes gpurun_out/<label>/)
#   pytest -m gpu (parity counts at the end of the log) | bench.py as the driver runs it (--steps 20 --warmup 5) and with its defaults |
#   rocprofv3 --kernel-trace --stats of the HEADLINE LOOP ALONE (bench.py --headline-only: nothing but the timed chain runs) ->
#   kernel_stats_headline.md + timeline.txt | the same of the training step (fwd + bwd) | SQ counters (three --pmc passes) | clock probe
L=${1:-r03}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -1 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
HL="python bench.py --steps 400 --warmup 20 --headline-only"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_headline -o t -- python $R/bench.py --steps 400 --warmup 20 --headline-only > $R/$O/prof_headline.json 2> /dev/null)
DB=$(find $O/prof_headline -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $O/kernel_stats_headline.md "$L: kernel statistics of the headline loop alone" "rocprofv3 --kernel-trace --stats -- $HL" > /dev/null
python tools/rocprof_timeline.py $DB 200 | grep -v columns > $O/timeline.txt
cat $O/timeline.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_train -o t -- python $R/bench.py --traffic-child train > /dev/null 2>&1)
DB2=$(find $O/prof_train -name "*.db" | head -1)
python tools/rocprof_summary.py $DB2 $O/kernel_stats_train.md "$L: kernel statistics of the training step (12 x forward with GSR_FLAG_BACKWARD_FOLLOWS + backward)" "rocprofv3 --kernel-trace --stats -- python bench.py --traffic-child train" > /dev/null
bash tools/sq_counters.sh $O/sq > /dev/null 2>&1
cp $O/sq/table.md $O/sq_counters_raw.md 2>/dev/null
timeout 120 python tools/clock_probe.py > $O/clock_probe.txt 2>&1
timeout 200 python tools/exp_chain.py $L 400 bwd 2>/dev/null | grep -v amdgpu
find $O -name "*.db" -size +20M -delete
