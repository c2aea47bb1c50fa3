#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_large_golden.py -m gpu -q -x -k "bench_path_parity_c5 or edge_shapes or c5shape or lloyd or kmeans_initialisation_wide" > gpurun_out/pytest_gate.log 2>&1 || { tail -30 gpurun_out/pytest_gate.log; echo "GATE FAILED"; exit 1; }
tail -2 gpurun_out/pytest_gate.log
report() {
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$1.json").read().splitlines()[0]); k = d["kernel_ms_total"]; n = d["steps"] * 10
    print("$1", round(d["value"]/1e6, 2), "M", round(d["ms_per_step"], 3), "ms | rtz_round us/launch", round(1e3 * k["rtz_round"] / n, 1), "finish", round(1e3 * k["rtz_reduce"] / n, 1), "ridge_stats", round(1e3 * k["ridge_stats"] / d["steps"], 1), "assign per round", round(1e3 * k["assign_block"] / n, 1), "tables", round(1e3 * k["block_table"] / n, 1), "apply", round(1e3 * k["ridge_apply"] / d["steps"], 1))
except Exception as e:
    print("$1 unreadable", e)
PY
}
timeout 300 python bench.py --config c5 --no-lisi --no-convergence --cpu-sample 0 --steps 3 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; report c5
HMX_RTZ3_TASKS=contig timeout 300 python bench.py --config c5 --no-lisi --no-convergence --cpu-sample 0 --steps 3 > gpurun_out/bench_c5c.json 2> gpurun_out/bench_c5c.err; report c5c
