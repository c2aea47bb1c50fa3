#!/bin/bash
# cycle stamps of profiling builds (-DHMX_ROUND_PROF -DHMX_RTZ3_PROF) on one box: usage gpu_r4_stamps.sh <config> lib1 lib2 ...
export TMPDIR=/tmp
mkdir -p gpurun_out
CFG="$1"; shift
for lib in "$@"; do
  echo "== $lib ($CFG)"
  HMX_LIB=$PWD/$lib timeout 300 python bench.py --config $CFG --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]"
done | tee gpurun_out/stamps_$CFG.txt
