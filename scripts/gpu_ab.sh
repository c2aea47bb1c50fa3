#!/bin/bash
# A/B of two builds on the C3 loop and the configs[4] shard: gate (parity of both bench paths), then the benches
# usage: gpu_ab.sh <other-lib> ; the default library is harmonypy_amd/libhmx.so
export TMPDIR=/tmp
mkdir -p gpurun_out
OTHER=${1:-build/libhmx_nostag.so}
timeout 500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "bench_path" -rP 2>&1 | grep -E "passed|failed|error|relF|Error" | tail -8 > gpurun_out/ab_gate.log
cat gpurun_out/ab_gate.log
if grep -qE "failed|error" gpurun_out/ab_gate.log; then echo "GATE FAILED"; exit 1; fi
for lib in default $OTHER; do
  if [ "$lib" = default ]; then unset HMX_LIB; else export HMX_LIB=$PWD/$lib; fi
  for cfg in c3 c5; do
    timeout 300 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$lib" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().splitlines()[0])
    print(sys.argv[1], sys.argv[2], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex, open("gpurun_out/ab.err").read()[-400:])
PY
  done
done
