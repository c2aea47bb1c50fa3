#!/bin/bash
# same-box A/B of two builds: baseline (build_abl/libhmx_base.so) vs the tree's library, interleaved
for rep in 1 2 3; do
for v in base new; do
  if [ "$v" = base ]; then lib=$PWD/build_abl/libhmx_base.so; else lib=$PWD/harmonypy_amd/libhmx.so; fi
  HMX_LIB=$lib python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-convergence "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],3), {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items() if v>0.05})"
done; done
