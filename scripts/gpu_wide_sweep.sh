#!/bin/bash
# the opt-in persistent wide sweep (HMX_WIDE_SWEEP=1) against the default one-launch-per-block path on the configs[4] shard
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in 1 0; do
  HMX_WIDE_SWEEP=$m timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/ws_$m.json 2> gpurun_out/ws_$m.err
  grep "hmx\]" gpurun_out/ws_$m.err | head -3
  python - $m <<'PY'
import json, sys
m = sys.argv[1]
d = json.loads(open(f"gpurun_out/ws_{m}.json").read().splitlines()[0])
print("HMX_WIDE_SWEEP=" + m, round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 2), "ms", d.get("kernel_ms_per_step"), d.get("sweep_waits"), d["roofline"].get("kernel"), round(d["roofline"]["frac"], 3), d["roofline"].get("avg_launch_us"))
PY
done
