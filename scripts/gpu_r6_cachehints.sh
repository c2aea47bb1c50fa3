#!/bin/bash
# Round 6, VERDICT item 5: Z_cos (200 MB at C3, read by both kernels of a round) left cacheable while the R traffic streams --
# variant libraries built with -DHMX_RTZ3_Z_NT=0 (-DHMX_ROUND_R_NT=1): same-box A/B of the round's kernel times.
# (FETCH_SIZE / WRITE_SIZE count at the L2 <-> fabric interface: hits of the memory-side cache do not show there, time does.)
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    r = d.get("roofline", {})
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", "k_round_us", round(r.get("avg_launch_us", 0), 1), d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, open("gpurun_out/w.err").read()[-800:])
PY
}
for rep in 1 2 3; do
  run "default (R and Z requests nt in k_rtz3c, plain R stores in k_round)" X=1
  run "Z requests of k_rtz3c cacheable + R stores of k_round nt" HMX_LIB=$PWD/harmonypy_amd/libhmx_exp1.so
  run "Z requests of k_rtz3c cacheable" HMX_LIB=$PWD/harmonypy_amd/libhmx_exp2.so
done 2>&1 | tee gpurun_out/ab_zcos_cacheable.txt
