#!/bin/bash
# GPU box: the final tree of round 6 (after the choose_chunk change) - smoke, bench.py with the driver's flags and with its defaults, kernel
# statistics of the nine workloads of profile_round6.sh and of the training batch.  usage: bash tools/profile_round6_final.sh <label>
L=${1:-r06_z}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
timeout 900 python bench.py > $O/bench_default_flags.json 2> $O/bench.err
prof() {  # prof <name> <title> <bench.py args...>
  local name=$1 title=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o t -- python $R/bench.py "$@" > $R/$O/prof_$name.json 2> /dev/null)
  local db=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_$name.md "$L: $title" "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null
  sed -n 7,11p $O/kernel_stats_$name.md | cut -c1-150
}
prof headline "kernel statistics of the headline loop alone" --steps 400 --warmup 20 --headline-only
prof train "kernel statistics of the training step (150 x forward with GSR_FLAG_BACKWARD_FOLLOWS + backward)" --traffic-child train
prof config4_fwd "BASELINE configs[3] forward: 3 views x 131 072 Gaussians, colour + depth, 150 calls through the plan API" --traffic-child cfg4_fwd
prof config4_train "BASELINE configs[3] training step: 3 views x 131 072 Gaussians, colour + depth, 150 x (forward + backward)" --traffic-child cfg4_train
prof config4s_fwd "configs[3] forward on the pixel-aligned (encoder-structured) scene, 150 calls" --traffic-child cfg4s_fwd
prof config4s_train "configs[3] training step on the pixel-aligned (encoder-structured) scene, 150 x (forward + backward)" --traffic-child cfg4s_train
prof shard131k "one 131 072-Gaussian view (BASELINE configs[4]'s share of one GPU), 150 calls" --traffic-child shard131k
prof 8_views "8 views of the 300 000-Gaussian scene in one call, 150 calls" --traffic-child views8
prof 48_views "48 views of a 131 072-Gaussian scene in one call, 40 calls" --traffic-child views48
for st in random pixel_aligned; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_batch4_$st -o t -- python $R/tools/exp_batch.py $st 4 > $R/$O/batch4_$st.txt 2> /dev/null)
  db=$(find $O/prof_batch4_$st -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_batch4_$st.md "$L: training batch, 4 scenes x 3 views x 131 072 Gaussians, colour + depth ($st scenes): forward loop, then training-step loop" "rocprofv3 --kernel-trace --stats -- python tools/exp_batch.py $st 4" > /dev/null
  sed -n 7,12p $O/kernel_stats_batch4_$st.md | cut -c1-150
done
timeout 300 python tools/decoder_call_profile.py > $O/decoder_call_profile.txt 2>&1
tail -6 $O/decoder_call_profile.txt
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
