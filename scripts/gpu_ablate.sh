#!/bin/bash
# Timing experiment: variant builds of libhmx.so (HMX_ABL switches) on C3; prints the per-family kernel ms.
mkdir -p gpurun_out
: > gpurun_out/ablate.txt
for v in 0 "$@"; do
  if [ "$v" = 0 ]; then lib=harmonypy_amd/libhmx.so; else lib=build_abl/libhmx_abl$v.so; fi
  HMX_LIB=$PWD/$lib python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_total']; r=d['roofline']
print('ABL=$v', 'ms/step=%.2f'%d['ms_per_step'], 'assign_us=%.2f'%r['avg_launch_us'], {a:round(b/2,2) for a,b in k.items()})" >> gpurun_out/ablate.txt
done
cat gpurun_out/ablate.txt
