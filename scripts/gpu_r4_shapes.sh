#!/bin/bash
# the two instances of k_round (default = bf16 pipe where it fits, f32 = HMX_ROUND_F32=1) on variations of a configuration's shape:
# usage gpu_r4_shapes.sh "c3=1000000,50,16,100" "c3=1250000,50,8,100" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for shape in "$@"; do
  cfg=${shape%%=*}
  for inst in default f32; do
    unset HMX_ROUND_F32; [ $inst = f32 ] && export HMX_ROUND_F32=1
    BENCH_SHAPE="$shape" timeout 300 python bench.py --config $cfg --steps 8 --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$inst" "$shape" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().splitlines()[0])
    print(sys.argv[2], sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", "k_round us", round(d["roofline"]["avg_launch_us"], 1), d["config"]["distance_gemm"][:6])
except Exception as ex:
    print(sys.argv[2], sys.argv[1], "FAILED", ex, open("gpurun_out/ab.err").read()[-300:])
PY
  done
done
