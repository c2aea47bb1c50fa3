#!/bin/bash
# Round 5, wide regime (K > 112 or d > 64; BASELINE configs[4]): parity gate on the wide tests incl. the direct A/B of the
# bf16-pipe kernels against the f32-input ones, then the configs[4] shard timed with each switch (one box), then the
# rocprofv3 kernel statistics of the default run.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -k "wide or edge_shapes or repeatable or c5 or large[c5shape] or lloyd or kmeans_initialisation_wide or without_the_streaming" -rP > gpurun_out/wide_gate_full.log 2>&1
grep -E "passed|failed|error|Error|wide path|wide bf16|configs\[4\] shape|bench path|c5shape" gpurun_out/wide_gate_full.log | tail -40
grep -B3 -A25 "^E  " gpurun_out/wide_gate_full.log | head -80
run() {
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex, open("gpurun_out/w.err").read()[-500:])
PY
}
for rep in 1 2; do
  run default X=1
  run assign_f32 HMX_ROUND_F32=1
  run rtz_f32 HMX_RTZ3_BF16=0
  run both_f32 HMX_ROUND_F32=1 HMX_RTZ3_BF16=0
done 2>&1 | tee gpurun_out/ab_wide_bf16.txt
rm -rf gpurun_out/prof_c5
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o r --output-format csv -- python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence --no-lisi > gpurun_out/prof_c5.json 2> gpurun_out/prof_c5.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_c5/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/c5_kernel_stats.txt", "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --config c5 --steps 2 --warmup 1 (configs[4] shard: 1.25 M cells x 200 PCs, K = 200, 32 batches)\n")
        out.write(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}\n")
        for r in rows[:16]:
            out.write(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}\n")
    print(open("gpurun_out/c5_kernel_stats.txt").read())
PY
find gpurun_out/prof_c5 -name '*kernel_trace.csv' -size +4M -delete
