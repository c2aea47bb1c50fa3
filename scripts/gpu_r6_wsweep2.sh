#!/bin/bash
# persistent wide sweep x list prefetch on the side stream: where does the list build of the next round go?
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, open("gpurun_out/w.err").read()[-1500:])
PY
}
{
for rep in 1 2; do
  run sweep_prefetch X=1
  run sweep_inline HMX_PREFETCH_LISTS=0
  run blocks_prefetch HMX_WIDE_SWEEP=0
  run blocks_inline HMX_WIDE_SWEEP=0 HMX_PREFETCH_LISTS=0
done
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ws -o ws -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_ws/**/ws_kernel_stats.csv', recursive=True)
for row in list(csv.reader(open(f[0])))[:22]: print(row[:6])
PY
} 2>&1 | tee gpurun_out/r6_wsweep2.txt
