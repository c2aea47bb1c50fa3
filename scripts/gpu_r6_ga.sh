#!/bin/bash
# Round 6: the GPU suite on the group-affine build, then same-box A/B of k_round's two tile maps (HMX_ROUND_GA=0: classic)
# on configs[2] (C3), configs[1], the configs[3] shard and all 10 M cells of configs[3] on one GPU.
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$1" != "noab" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q -rP > gpurun_out/ga_pytest_full.log 2>&1
grep -E "passed|failed|error|group-affine vs|free-running|library loop|C client" gpurun_out/ga_pytest_full.log | tail -40
fi
run() {
  local label=$1 cfg=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $cfg --steps $steps --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    r = d.get("roofline", {})
    print(sys.argv[1], sys.argv[2], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", "k_round_us", round(r.get("avg_launch_us", 0), 1), d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex, open("gpurun_out/w.err").read()[-800:])
PY
}
for rep in 1 2; do
  for cs in c3:10 c2:40 c4:5 c4x1:2; do
    run group_affine ${cs%%:*} ${cs##*:} X=1
    run classic ${cs%%:*} ${cs##*:} HMX_ROUND_GA=0
  done
done 2>&1 | tee gpurun_out/ab_group_affine_r6.txt
