#!/bin/bash
# Runs on the GPU box (through gpurun): round 6's evidence in one call.  usage: bash tools/profile_round6.sh <label>
#   pytest -m gpu | bench.py with the driver's flags and with its defaults | rocprofv3 --kernel-trace --stats of NINE workloads (headline loop,
#   training step, configs[3] forward / training step on the independently drawn AND on the pixel-aligned scene, one 131 072-Gaussian view,
#   8 views, 48 views) -> kernel_stats_*.md | SQ counters of the training step, configs[3] forward (both scenes) and the 131 072-Gaussian view |
#   phase stamps of both forward launches for five shapes (tools/phase_stamps.py, tools/libgsr_hip_ablate.so cross-compiled beforehand) |
#   the skip-rate study | clock probe | HIP and fp32 oracle against the fp64 oracle
L=${1:-r06}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -1 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -1 $O/smoke.log
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
prof config4s_fwd "configs[3] forward on the pixel-aligned (encoder-structured) scene, 150 calls" --traffic-child cfg4s_fwd
prof config4s_train "configs[3] training step on the pixel-aligned (encoder-structured) scene, 150 x (forward + backward)" --traffic-child cfg4s_train
prof shard131k "one 131 072-Gaussian view (BASELINE configs[4]'s share of one GPU), 150 calls" --traffic-child shard131k
prof 8_views "8 views of the 300 000-Gaussian scene in one call, 150 calls" --traffic-child views8
prof 48_views "48 views of a 131 072-Gaussian scene in one call, 40 calls" --traffic-child views48
for pair in "sq:train" "sq_cfg4:cfg4_fwd" "sq_cfg4s:cfg4s_fwd" "sq_shard:shard131k"; do
  d=${pair%%:*}; m=${pair##*:}
  bash tools/sq_counters.sh $O/$d bench.py --traffic-child $m > /dev/null 2>&1
  cp $O/$d/table.md $O/sq_counters_${m}_raw.md 2>/dev/null
  python tools/sq_derived.py $O/sq_counters_${m}_raw.md > $O/sq_counters_${m}.md 2>/dev/null
done
export GSR_KEEP_LIB=1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 timeout 300 python tools/phase_stamps.py "configs[3] forward, independently drawn scene" 2>&1 | grep -v amdgpu.ids > $O/stamps_config4.txt
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "configs[3] forward, pixel-aligned scene" 2>&1 | grep -v amdgpu.ids > $O/stamps_config4_structured.txt
PS_N=131072 PS_V=1 PS_EXTRA=0 PS_SEED=50 timeout 300 python tools/phase_stamps.py "one 131 072-Gaussian view, independently drawn scene" 2>&1 | grep -v amdgpu.ids > $O/stamps_shard.txt
PS_N=131072 PS_V=1 PS_EXTRA=0 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "one 131 072-Gaussian view, pixel-aligned scene" 2>&1 | grep -v amdgpu.ids > $O/stamps_shard_structured.txt
PS_N=300000 PS_V=1 PS_EXTRA=0 PS_SEED=2 timeout 300 python tools/phase_stamps.py "headline: one 300 000-Gaussian view" 2>&1 | grep -v amdgpu.ids > $O/stamps_headline.txt
unset GSR_KEEP_LIB
timeout 900 python tools/skip_rate.py 64 > $O/skip_rate.md 2> $O/skip_rate.err
timeout 120 python tools/clock_probe.py > $O/clock_probe.txt 2>&1
timeout 600 python tools/parity_vs_fp64.py $O/parity_vs_fp64.md > $O/parity_vs_fp64.log 2>&1
tail -3 $O/parity_vs_fp64.log
timeout 300 python tools/decoder_call_profile.py > $O/decoder_call_profile.txt 2>&1
tail -8 $O/decoder_call_profile.txt
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
