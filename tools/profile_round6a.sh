#!/bin/bash
# Runs on the GPU box (through gpurun): round 6, first evidence call - what VERDICT r05 item 1(a) found missing.
#   phase stamps (K1 per workgroup / per colour unit, K2 per tile) + rocprofv3 kernel statistics + SQ counters of the shapes PF3plat
#   really runs: configs[3] (3 views x 131 072, colour + depth) and one 131 072-Gaussian view (configs[4]'s per-GPU shard), each on
#   the random scene of SURVEY 8d and on the pixel-aligned (encoder-structured) scene.
# usage: bash tools/profile_round6a.sh <label>      (tools/libgsr_hip_ablate.so cross-compiled beforehand: GSR_KEEP_LIB=1)
L=${1:-r06_a}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
tail -1 $O/pytest.log
fi
export GSR_KEEP_LIB=1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 timeout 300 python tools/phase_stamps.py "configs[3] forward, random scene" > $O/stamps_config4.txt 2>&1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "configs[3] forward, pixel-aligned scene" > $O/stamps_config4_structured.txt 2>&1
PS_N=131072 PS_V=1 PS_EXTRA=0 PS_SEED=50 timeout 300 python tools/phase_stamps.py "one 131 072-Gaussian view, random scene" > $O/stamps_shard.txt 2>&1
PS_N=131072 PS_V=1 PS_EXTRA=0 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "one 131 072-Gaussian view, pixel-aligned scene" > $O/stamps_shard_structured.txt 2>&1
PS_N=300000 PS_V=1 PS_EXTRA=0 PS_SEED=2 timeout 300 python tools/phase_stamps.py "headline: one 300 000-Gaussian view" > $O/stamps_headline.txt 2>&1
tail -n +1 $O/stamps_*.txt | cut -c1-400
unset GSR_LIB_PATH
prof() {  # prof <name> <title> <bench.py args...>
  local name=$1 title=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o t -- python $R/bench.py "$@" > $R/$O/prof_$name.json 2> /dev/null)
  local db=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_$name.md "$L: $title" "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null
  head -20 $O/kernel_stats_$name.md | cut -c1-220
}
prof config4_fwd "BASELINE configs[3] forward: 3 views x 131 072 Gaussians, colour + depth, 150 calls through the plan API" --traffic-child cfg4_fwd
prof config4_train "BASELINE configs[3] training step, 150 x (forward + backward)" --traffic-child cfg4_train
prof shard131k "one 131 072-Gaussian view (configs[4]'s per-GPU shard), 150 calls" --traffic-child shard131k
prof config4s_fwd "configs[3] forward on the pixel-aligned scene" --traffic-child cfg4s_fwd
prof config4s_train "configs[3] training step on the pixel-aligned scene" --traffic-child cfg4s_train
bash tools/sq_counters.sh $O/sq_cfg4 bench.py --traffic-child cfg4_fwd > /dev/null 2>&1
cp $O/sq_cfg4/table.md $O/sq_counters_config4_fwd_raw.md 2>/dev/null
bash tools/sq_counters.sh $O/sq_shard bench.py --traffic-child shard131k > /dev/null 2>&1
cp $O/sq_shard/table.md $O/sq_counters_shard131k_raw.md 2>/dev/null
bash tools/sq_counters.sh $O/sq_cfg4s bench.py --traffic-child cfg4s_fwd > /dev/null 2>&1
cp $O/sq_cfg4s/table.md $O/sq_counters_config4s_fwd_raw.md 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
tail -2 $O/bench_driver_flags.err
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
