#!/bin/bash
# Cycle stamps of k_round (library built with -DHMX_ROUND_PROF as harmonypy_amd/libhmx_prof.so) and plain timings:
# group-affine map vs classic, with and without the side-stream list build, and the chain wave's experiment knobs.
export TMPDIR=/tmp
mkdir -p gpurun_out
cfg=${1:-c3}
run() {
  echo "== $cfg $*"
  env "$@" HMX_LIB=$PWD/harmonypy_amd/libhmx_prof.so timeout 300 python bench.py --config $cfg --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline 2>&1 >/dev/null | grep "k_round prof"
  env "$@" timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --cpu-sample 0 --no-convergence --no-lisi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('   plain build:', round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms  k_round_us', round(r.get('avg_launch_us',0),1))"
}
{
if [ -n "$R6_RUNS" ]; then
  IFS=';' read -ra RUNS <<< "$R6_RUNS"
  for r in "${RUNS[@]}"; do run $r; done
else
run HMX_ROUND_GA=1
run HMX_ROUND_GA=0
fi
} 2>&1 | tee gpurun_out/r6_stamps_$cfg.txt
