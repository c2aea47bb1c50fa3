#!/bin/bash
# round 2, call 1: full GPU parity suite (with the new bench-path tests; -rP prints the natural-run report) + a C3 bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -rP 2>&1 | grep -v "^$" | tail -80 > gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-convergence > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
grep -E "passed|failed|error|same schedule|diverged|bench path" gpurun_out/pytest_gpu.log | tail -40
cut -c1-600 gpurun_out/bench_c3.json
