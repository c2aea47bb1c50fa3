#!/bin/bash
# Round 5: the three-term split with four single v_sub_f32 (default) against the packed form (build/libhmx_splitpk.so,
# -DHMX_SPLIT_PK=1), same box: parity gate of the bf16-pipe A/B tests, then C3 / configs[1] / configs[4] shard timed twice each.
export TMPDIR=/tmp
mkdir -p gpurun_out
[ -z "$SKIP_GATE" ] && timeout 900 python -m pytest tests -m gpu -q -k "bf16_pipe or split or c3_shape or c5_shape" -rP > gpurun_out/split_gate_full.log 2>&1
grep -E "passed|failed|error|Error" gpurun_out/split_gate_full.log | tail -5
grep -B3 -A25 "^E  " gpurun_out/split_gate_full.log | head -60
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
  for cs in c3:10 c2:40 c5:3; do
    run single_subs ${cs%%:*} ${cs##*:} X=1
    run packed_subs ${cs%%:*} ${cs##*:} HMX_LIB=$PWD/build/libhmx_splitpk.so
  done
done 2>&1 | tee gpurun_out/ab_split_subs.txt
