#!/bin/bash
# same-box comparison of the round paths on C3: k_sweep (HMX_SWEEP=1) vs k_round + separate R^T.Z pass (HMX_SWEEP=0)
for rep in 1 2; do
for v in 1 0; do
  HMX_SWEEP=$v timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-convergence "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('HMX_SWEEP=$v', round(d['value']/1e6,2), 'M cells/s/it', round(d['ms_per_step'],3), 'ms/step', {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items() if v>0.05})"
done; done
