#!/bin/bash
# quick GPU loop: parity tests (fail fast) + C3/C2 bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for c in c3 c2; do
timeout 300 python bench.py --config $c --steps 10 --warmup 2 --cpu-sample 0 --no-convergence 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('$c', round(d['value']/1e6,2),'M cells/s/it', round(d['ms_per_step'],3),'ms/step', {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items() if v>0.01}, 'sweep_us', round(d['roofline']['avg_launch_us'],1))"
done
