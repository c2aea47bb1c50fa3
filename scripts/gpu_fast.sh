#!/bin/bash
# fast GPU loop for kernel work: the parity tests of the timed path + the C3 bench line (+ phase stamps if a
# profiling build exists: python -m harmonypy_amd._build -DHMX_ROUND_PROF -o build_abl/libhmx_prof.so)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "bench_path or golden or natural" 2>&1 | grep -E "Error|error|assert|passed|failed|relF|mismatch" | tail -12
for c in c3 c2; do
timeout 300 python bench.py --config $c --steps 10 --warmup 2 --cpu-sample 0 --no-convergence 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('$c', round(d['value']/1e6,2),'M cells/s/it', round(d['ms_per_step'],3),'ms/step', {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items() if v>0.01}, 'sweep_us', round(d['roofline']['avg_launch_us'],1))"
done
if [ -f build_abl/libhmx_prof.so ]; then
HMX_LIB=$PWD/build_abl/libhmx_prof.so timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-convergence 2>&1 | grep "prof\]"
fi
