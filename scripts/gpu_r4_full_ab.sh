#!/bin/bash
# full GPU suite on the default library, then an A/B of libraries on the C3 / C2 / 10M-cell loops (one box)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rP > gpurun_out/r4_pytest_full.log 2>&1
grep -E "passed|failed|error|Error|wide path|configs\[4\] shape|same schedule|diverged|bench path|2 shards|time-out|1 vs 2|c3full|4 ranks" gpurun_out/r4_pytest_full.log | tail -80 > gpurun_out/r4_pytest.log
tail -6 gpurun_out/r4_pytest.log
grep -B3 -A25 "^E  " gpurun_out/r4_pytest_full.log | head -120
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset HMX_LIB; else export HMX_LIB=$PWD/$lib; fi
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
