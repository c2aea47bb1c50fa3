#!/bin/bash
# Same-box A/B of variant libraries (HMX_LIB) against the default build:  gpu_r6_ab_libs.sh "<label>=<lib>;..." "cfg:steps ..."
export TMPDIR=/tmp
mkdir -p gpurun_out
IFS=';' read -ra LIBS <<< "$1"
run() {
  local label=$1 cfg=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $cfg --steps $steps --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    r = d.get("roofline", {})
    print(sys.argv[1], sys.argv[2], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", "dominant_us", round(r.get("avg_launch_us", 0), 1), d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex, open("gpurun_out/w.err").read()[-800:])
PY
}
for rep in 1 2 3; do
  for cs in ${2:-c3:10 c2:40}; do
    run default ${cs%%:*} ${cs##*:} X=1
    for l in "${LIBS[@]}"; do run "${l%%=*}" ${cs%%:*} ${cs##*:} HMX_LIB=$PWD/${l##*=}; done
  done
done 2>&1 | tee gpurun_out/ab_libs.txt
