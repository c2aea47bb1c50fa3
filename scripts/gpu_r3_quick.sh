#!/bin/bash
# quick loop: gate tests, stamps of k_rtz3, C3 bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_large_golden.py -m gpu -q -x -k "bench_path_parity_c3 or forced_schedule or engine_vs_reference or edge_shapes or ridge_matches" > gpurun_out/pytest_gate.log 2>&1 || { tail -20 gpurun_out/pytest_gate.log; echo "GATE FAILED"; exit 1; }
tail -2 gpurun_out/pytest_gate.log
HMX_LIB=$PWD/build/libhmx_prof.so timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 --steps 4 2>&1 | grep -E "k_rtz3 prof" | cut -c1-400
timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_q.json").read().splitlines()[0]); k = d["kernel_ms_total"]; n = d["steps"] * 10
print("bench", round(d["value"]/1e6, 2), "M", round(d["ms_per_step"], 3), "ms | rtz_round us/launch", round(1e3 * k["rtz_round"] / n, 1), "finish", round(1e3 * k["rtz_reduce"] / n, 1), "ridge_stats", round(1e3 * k["ridge_stats"] / d["steps"], 1), "k_round", round(1e3 * k["assign_block"] / n, 1), "apply", round(1e3 * k["ridge_apply"] / d["steps"], 1))
PY
