"""Runs bench.py (no CPU baseline) and prints the few numbers used when iterating on kernels.  usage: python tools/bench_brief.py [label]"""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300).stdout
d = json.loads(out.strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", "fwd us", round(d["eager_ms_per_step"] * 1e3, 2), "views/s", round(d["value"]),
      "stages", {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items() if not isinstance(v, str)}, "| bwd us", round(d.get("bwd_ms", 0) * 1e3, 1),
      {k: round(v * 1e3, 1) for k, v in d.get("bwd_stage_ms", {}).items()},
      "| 8 views/s", round(d["batched_8_views"]["views_per_s"]), "| decoder views/s", round(d["decoder_config4"]["views_per_s"]))
