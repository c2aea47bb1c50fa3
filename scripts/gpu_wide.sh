#!/bin/bash
# wide regime (configs[4] shape): parity gate, then the shard bench with the old and the new streaming kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_large_golden.py -m gpu -q -x -k "c5 or wide or lloyd" -rP 2>&1 | grep -E "passed|failed|error|relF|Error" | tail -15 > gpurun_out/wide_gate.log
cat gpurun_out/wide_gate.log
if grep -qE "failed|error" gpurun_out/wide_gate.log; then echo "GATE FAILED"; exit 1; fi
for m in 1 2; do
  HMX_RTZW=$m timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/wide_rtzw$m.json 2> gpurun_out/wide_rtzw$m.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/wide_rtzw$m.json").read().splitlines()[0])
print("HMX_RTZW=$m", round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 2), "ms", d.get("kernel_ms_per_step"))
PY
done
