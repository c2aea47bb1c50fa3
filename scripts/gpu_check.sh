#!/bin/bash
# Run on the GPU box through gpurun: parity tests, bench line, rocprofv3 kernel stats.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
HMX_BENCH_FORCE_SHARD=1 python bench.py --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/bench_c3_rccl1.json 2> gpurun_out/bench_c3_rccl1.err
python bench.py --config c2 --steps 20 --warmup 2 --cpu-sample 0 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
if [ "$1" = "prof" ]; then
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-roofline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
find gpurun_out/prof -name '*kernel_trace.csv' -size +8M -delete
fi
tail -4 gpurun_out/pytest_gpu.log; cut -c1-400 gpurun_out/bench_c3.json; echo; cut -c1-300 gpurun_out/bench_c3_rccl1.json; tail -3 gpurun_out/bench_c3_rccl1.err
