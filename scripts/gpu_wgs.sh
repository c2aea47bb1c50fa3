#!/bin/bash
# timing experiment: grid size of the persistent sweep kernel (HMX_ROUND_WGS) on C3
for w in 0 196 208 240 255; do
  HMX_ROUND_WGS=$w timeout 200 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-convergence 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('WGS=$w', round(d['value']/1e6,2), 'M cells/s/it; sweep_us', round(d['roofline']['avg_launch_us'],1), d['sweep_waits']['incomplete_polls_mean'])"
done
