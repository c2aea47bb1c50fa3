#!/bin/bash
# timing experiment: ablation builds of k_sweep (HMX_SABL) on C3
for v in 0 "$@"; do
  if [ "$v" = 0 ]; then lib=harmonypy_amd/libhmx.so; else lib=build_abl/libhmx_sabl$v.so; fi
  HMX_LIB=$PWD/$lib timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-convergence 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('SABL=$v sweep_us', round(d['roofline']['avg_launch_us'],1))"
done
