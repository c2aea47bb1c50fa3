#!/bin/bash
# Cycle stamps of k_assign_wide3 at the configs[4] shard (library built with -DHMX_WIDE3_PROF as harmonypy_amd/libhmx_w3prof.so)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2; do
env "$@" HMX_LIB=$PWD/harmonypy_amd/libhmx_w3prof.so timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline 2>&1 >/dev/null | grep "wide3 prof"
done
timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('   plain build:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms  dominant_us', round(r.get('avg_launch_us',0),1), d.get('kernel_ms_per_step'))"
} 2>&1 | tee gpurun_out/r6_w3prof.txt
