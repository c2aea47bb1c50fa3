#!/bin/bash
# k_rtz3: where the time goes.  Ablation builds (no MFMAs / no requests), then SQ + MFMA + HBM counters of the C3 loop.
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in abl1 abl2; do
  HMX_LIB=$PWD/build/libhmx_$v.so timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 --steps 3 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$v.json").read().splitlines()[0]); k = d["kernel_ms_total"]; n = d["steps"] * 10
    print("$v", "rtz_round us/launch", round(1e3 * k["rtz_round"] / n, 1), "ridge_stats", round(1e3 * k["ridge_stats"] / d["steps"], 1), "k_round", round(1e3 * k["assign_block"] / n, 1))
except Exception as e:
    print("$v unreadable", e)
PY
done
bash scripts/gpu_pmc_sq.sh 2>&1 | tail -12
bash scripts/gpu_pmc_mfma.sh 2>&1 | tail -8
bash scripts/gpu_pmc.sh 2>&1 | tail -10
