#!/bin/bash
# Round 5: k_rtz3c (eight waves, one pair buffer each) against k_rtz3b (HMX_RTZ3_WAVES=4), same box: parity gate of everything
# that runs the narrow streaming pass, then C3 / configs[1] timed twice each, then the kernel statistics of the default run.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "rtz or bench_path or large or matches_oracle or golden or config3 or config2 or edge_shapes" -rP > gpurun_out/rtz3c_gate_full.log 2>&1
grep -E "passed|failed|error|Error" gpurun_out/rtz3c_gate_full.log | tail -5
grep -B3 -A25 "^E  " gpurun_out/rtz3c_gate_full.log | head -60
run() {
  local label=$1 cfg=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $cfg --steps $steps --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    print(sys.argv[1], sys.argv[2], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex, open("gpurun_out/w.err").read()[-500:])
PY
}
for rep in 1 2; do
  for cs in c3:10 c2:40; do
    run eight_waves ${cs%%:*} ${cs##*:} X=1
    run four_waves ${cs%%:*} ${cs##*:} HMX_RTZ3_WAVES=4
  done
done 2>&1 | tee gpurun_out/ab_rtz3c.txt
