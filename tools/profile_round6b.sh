#!/bin/bash
# Runs on the GPU box: round 6, second call - after the page-counter fix, 512-Gaussian chunks and the cooperative long-run gather.
L=${1:-r06_b}
O=gpurun_out/$L
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20
fi
export GSR_KEEP_LIB=1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 timeout 300 python tools/phase_stamps.py "configs[3] forward, random scene" > $O/stamps_config4.txt 2>&1
PS_N=131072 PS_V=3 PS_EXTRA=1 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "configs[3] forward, pixel-aligned scene" > $O/stamps_config4_structured.txt 2>&1
PS_N=131072 PS_V=1 PS_EXTRA=0 PS_SEED=50 timeout 300 python tools/phase_stamps.py "one 131 072-Gaussian view, random scene" > $O/stamps_shard.txt 2>&1
PS_N=131072 PS_V=1 PS_EXTRA=0 PS_SEED=50 PS_STRUCT=pixel_aligned timeout 300 python tools/phase_stamps.py "one 131 072-Gaussian view, pixel-aligned scene" > $O/stamps_shard_structured.txt 2>&1
PS_N=300000 PS_V=1 PS_EXTRA=0 PS_SEED=2 timeout 300 python tools/phase_stamps.py "headline: one 300 000-Gaussian view" > $O/stamps_headline.txt 2>&1
for f in $O/stamps_*.txt; do echo "=== $f"; grep -v amdgpu.ids $f | cut -c1-420; done
unset GSR_KEEP_LIB
prof() {  # prof <name> <title> <bench.py args...>
  local name=$1 title=$2; shift 2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o t -- python $R/bench.py "$@" > $R/$O/prof_$name.json 2> /dev/null)
  local db=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $O/kernel_stats_$name.md "$L: $title" "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null
  sed -n 7,12p $O/kernel_stats_$name.md | cut -c1-160
}
prof config4_fwd "BASELINE configs[3] forward: 3 views x 131 072 Gaussians, colour + depth, 150 calls through the plan API" --traffic-child cfg4_fwd
prof shard131k "one 131 072-Gaussian view (configs[4]'s per-GPU shard), 150 calls" --traffic-child shard131k
prof config4s_fwd "configs[3] forward on the pixel-aligned scene" --traffic-child cfg4s_fwd
prof config4s_train "configs[3] training step on the pixel-aligned scene" --traffic-child cfg4s_train
prof headline "headline loop" --steps 400 --warmup 20 --headline-only
timeout 900 python tools/skip_rate.py 64 > $O/skip_rate.md 2> $O/skip_rate.err
cat $O/skip_rate.md; tail -3 $O/skip_rate.err
find $O -name "*.db" -size +20M -delete
find $O -name "*.csv" -size +2M -delete
