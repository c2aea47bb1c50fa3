#!/bin/bash
# The persistent wide sweep (k_sweep_wide3) against one launch per block (HMX_WIDE_SWEEP=0), same box, alternating
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config ${CFG:-c5} --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    r = d.get("roofline", {})
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", "dominant_us", round(r.get("avg_launch_us", 0), 1), d.get("kernel_ms_per_step"), d.get("sweep_waits"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, open("gpurun_out/w.err").read()[-1500:])
PY
}
for rep in 1 2 3; do
  run persistent_sweep X=1
  run launch_per_block HMX_WIDE_SWEEP=0
done 2>&1 | tee gpurun_out/r6_wsweep.txt
