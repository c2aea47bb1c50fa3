#!/bin/bash
# wide shapes: k_rtzw + device Lloyd.  Gate tests, then the C5 shard and C3 bench lines, then the whole GPU suite.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_large_golden.py -m gpu -q -x -k "bench_path_parity or forced_schedule or engine_vs_reference or edge_shapes or ridge_matches or lloyd or kmeans_initialisation" > gpurun_out/pytest_gate.log 2>&1 || { tail -30 gpurun_out/pytest_gate.log; echo "GATE FAILED"; exit 1; }
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
HMX_RTZ=2 timeout 300 python bench.py --config c5 --no-lisi --no-convergence --cpu-sample 0 --steps 3 > gpurun_out/bench_c5old.json 2> gpurun_out/bench_c5old.err; report c5old
timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; report c3
HMX_LIB=$PWD/build/libhmx_prof.so timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 --steps 4 2>&1 | grep -E "k_rtz3 prof" | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail=12 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|FAILED|ERROR" gpurun_out/pytest_gpu_full.log | tail -8
