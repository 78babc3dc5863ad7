#!/bin/bash
# Runs on the GPU box (through gpurun): round 5's evidence in one call.  usage: bash tools/profile_round5.sh <label> [fuzz cases per seed]
#   pytest -m gpu | bench.py with the driver's flags and with its defaults | rocprofv3 --kernel-trace --stats of the headline loop alone, the
#   training step, BASELINE configs[3] (forward / training step), 8 views and 48 views per call -> kernel_stats_*.md | SQ counters of the
#   training step, the 8-view and the 48-view call | clock probe | HIP and fp32 oracle against the fp64 oracle | the fuzz survey
L=${1:-r05}
FUZZ=${2:-0}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$PERF_ONLY" ]; then
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -1 $O/pytest.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
prof() {  # prof <name> <title> <bench.py args...>
  local name=$1 title=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o t -- python $R/bench.py "$@" > $R/$O/prof_$name.json 2> /dev/null)
  local db=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_$name.md "$L: $title" "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null
}
prof headline "kernel statistics of the headline loop alone" --steps 400 --warmup 20 --headline-only
python tools/rocprof_timeline.py $(find $O/prof_headline -name "*.db" | head -1) 200 | grep -v columns > $O/timeline.txt
cat $O/timeline.txt
prof train "kernel statistics of the training step (150 x forward with GSR_FLAG_BACKWARD_FOLLOWS + backward)" --traffic-child train
prof config4_fwd "BASELINE configs[3] forward: 3 views x 131 072 Gaussians, colour + depth, 150 calls through the plan API" --traffic-child cfg4_fwd
prof config4_train "BASELINE configs[3] training step: 3 views x 131 072 Gaussians, colour + depth, 150 x (forward + backward)" --traffic-child cfg4_train
prof 8_views "8 views of the 300 000-Gaussian scene in one call, 150 calls" --traffic-child views8
prof 48_views "48 views of a 131 072-Gaussian scene in one call, 40 calls" --traffic-child views48
bash tools/sq_counters.sh $O/sq > /dev/null 2>&1
cp $O/sq/table.md $O/sq_counters_raw.md 2>/dev/null
bash tools/sq_counters.sh $O/sq8 bench.py --traffic-child views8 > /dev/null 2>&1
cp $O/sq8/table.md $O/sq_counters_8_views_raw.md 2>/dev/null
bash tools/sq_counters.sh $O/sq48 bench.py --traffic-child views48 > /dev/null 2>&1
cp $O/sq48/table.md $O/sq_counters_48_views_raw.md 2>/dev/null
timeout 120 python tools/clock_probe.py > $O/clock_probe.txt 2>&1
if [ -z "$PERF_ONLY" ]; then
timeout 600 python tools/parity_vs_fp64.py $O/parity_vs_fp64.md > $O/parity_vs_fp64.log 2>&1
tail -3 $O/parity_vs_fp64.log
fi
timeout 120 ./tools/micro/blend_mix_bench > $O/blend_mix_bench.txt 2>&1
cat $O/blend_mix_bench.txt
if [ "$FUZZ" != "0" ]; then
  timeout 1500 python tools/fuzz_parity.py --cases $FUZZ --seeds ${FUZZ_SEEDS:-11,12,13,14,15} --out $O/fuzz.json > $O/fuzz.log 2>&1
  tail -2 $O/fuzz.log
  python tools/fuzz_report.py $O/fuzz.json $O/fuzz_histogram.md > /dev/null 2>&1
fi
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
