#!/bin/bash
# quick look at the wide regime: parity of the configs[4] shape, cycle stamps (profiling build if present), shard bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "c5" -rP 2>&1 | grep -E "passed|failed|error|relF|Error" | tail -6
if [ -f build/libhmx_prof.so ]; then
  HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]"
fi
timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/wide_quick.json 2> gpurun_out/wide_quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/wide_quick.json").read().splitlines()[0])
print(round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 2), "ms", d.get("kernel_ms_per_step"))
PY
