#!/bin/bash
# process-to-process spread of the configs[4] shard: fresh processes, persistent sweep vs one launch per block, kernel families
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, open("gpurun_out/w.err").read()[-800:])
PY
}
for rep in 1 2 3; do
  run sweep X=1
  run blocks HMX_WIDE_SWEEP=0
  run sweep_lists_beside_sweep HMX_LISTS_BESIDE_RTZ=0
  run blocks_lists_beside_blocks HMX_WIDE_SWEEP=0 HMX_LISTS_BESIDE_RTZ=0
done 2>&1 | tee gpurun_out/r6_wsweep3.txt
