#!/bin/bash
# Round-3 working check (3): k_rtz3 as a continuous k-step pipeline; interleaved vs contiguous tasks; ablations
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_large_golden.py -m gpu -q -x -k "bench_path_parity_c3 or forced_schedule or engine_vs_reference or edge_shapes" > gpurun_out/pytest_gate.log 2>&1 || { tail -20 gpurun_out/pytest_gate.log; echo "GATE FAILED"; exit 1; }
tail -2 gpurun_out/pytest_gate.log
report() {
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$1.json").read().splitlines()[0]); k = d["kernel_ms_total"]; n = d["steps"] * 10
    print("$1", round(d["value"]/1e6, 2), "M", round(d["ms_per_step"], 3), "ms | rtz_round us/launch", round(1e3 * k["rtz_round"] / n, 1), "finish", round(1e3 * k["rtz_reduce"] / n, 1), "ridge_stats", round(1e3 * k["ridge_stats"] / d["steps"], 1), "k_round", round(1e3 * k["assign_block"] / n, 1), "apply", round(1e3 * k["ridge_apply"] / d["steps"], 1))
except Exception as e:
    print("$1 unreadable", e)
PY
}
timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_inter.json 2> gpurun_out/bench_inter.err; report inter
HMX_LIB=$PWD/build/libhmx_nont.so timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_nont.json 2> gpurun_out/bench_nont.err; report nont
HMX_RTZ3_TASKS=contig timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_contig.json 2> gpurun_out/bench_contig.err; report contig
for v in abl1 abl2; do
  HMX_LIB=$PWD/build/libhmx_$v.so timeout 200 python bench.py --no-lisi --no-convergence --cpu-sample 0 --steps 3 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err; report $v
done
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail=12 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|FAILED|ERROR" gpurun_out/pytest_gpu_full.log | tail -8
