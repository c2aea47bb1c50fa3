#!/bin/bash
# wide regime: parity gate on the wide tests, then configs[4]-shard A/B (one box)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -k "edge_shapes or repeatable or c5 or wide or large[c5shape] or lloyd or kmeans_initialisation_wide or without_the_streaming" -rP > gpurun_out/wide_gate_full.log 2>&1
grep -E "passed|failed|error|Error|wide path|configs\[4\] shape|bench path|c5shape" gpurun_out/wide_gate_full.log | tail -30
grep -B3 -A25 "^E  " gpurun_out/wide_gate_full.log | head -60
run() {
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, open("gpurun_out/w.err").read()[-500:])
PY
}
for rep in 1 2; do
  run default X=1
  run table_separate HMX_WIDE_TABLE=separate
  run previous_build HMX_LIB=$PWD/build/libhmx_nors.so
done
