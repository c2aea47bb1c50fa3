#!/bin/bash
# phase stamps of k_round (profiling build) + the C3 bench line of the regular build
timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --cpu-sample 0 --no-convergence 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('c3', round(d['value']/1e6,2),'M cells/s/it', round(d['ms_per_step'],3),'ms/step', {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items() if v>0.01}, 'sweep_us', round(d['roofline']['avg_launch_us'],1))"
HMX_LIB=$PWD/build_abl/libhmx_prof.so timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-convergence 2>&1 | grep "prof\]"
