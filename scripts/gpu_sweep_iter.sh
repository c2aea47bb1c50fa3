#!/bin/bash
# kernel iteration loop: quick parity subset, then phase stamps (profiling build) and a short C3 bench line
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "cluster_round or edge_shapes or bench_path_parity_c3" 2>&1 | tail -4
if [ -f build_abl/libhmx_prof.so ]; then
HMX_LIB=$PWD/build_abl/libhmx_prof.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-convergence 2>&1 >/dev/null | grep "k_sweep prof"
fi
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-convergence 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print(round(d['value']/1e6,2),'M cells/s/it', round(d['ms_per_step'],3),'ms/step', {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items() if v>0.01}, 'sweep_us', round(d['roofline']['avg_launch_us'],1))"
