#!/bin/bash
# A/B of libraries on ONE box: usage gpu_r4_ab.sh "<gate -k expression>" lib1 lib2 ...   ("default" = harmonypy_amd/libhmx.so; "f32" = the same with HMX_ROUND_F32=1: the f32-input instances of k_round)
# the parity gate runs on the default library first; then `--config c3` and c2 per library, twice, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
GATE="$1"; shift
if [ -n "$GATE" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -k "$GATE" -rP > gpurun_out/ab_gate_full.log 2>&1
  grep -E "passed|failed|error|relF|Error|same schedule|diverged" gpurun_out/ab_gate_full.log | tail -40 > gpurun_out/ab_gate.log
  cat gpurun_out/ab_gate.log
  if grep -qE "failed|error" gpurun_out/ab_gate.log; then echo "GATE FAILED"; grep -B5 -A25 "Error" gpurun_out/ab_gate_full.log | head -80; fi
fi
for rep in 1 2; do
for lib in "$@"; do
  unset HMX_ROUND_F32
  if [ "$lib" = default ]; then unset HMX_LIB; elif [ "$lib" = f32 ]; then unset HMX_LIB; export HMX_ROUND_F32=1; else export HMX_LIB=$PWD/$lib; fi
  for cfg in ${CFGS:-c3 c2}; do
    timeout 300 python bench.py --config $cfg --steps 8 --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$lib" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().splitlines()[0])
    r = d.get("roofline", {})
    print(sys.argv[1], sys.argv[2], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", "k_round us", r.get("avg_launch_us"), d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex, open("gpurun_out/ab.err").read()[-400:])
PY
  done
done
done
