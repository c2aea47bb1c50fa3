#!/bin/bash
# Round-5 evidence on ONE box: GPU suite + smoke + default bench + rocprofv3 kernel statistics (gpu_final.sh), the counter
# passes of C3 (HBM, MFMA, SQ) and of the configs[4] shard (SQ, MFMA), then same-box A/B of the bf16-pipe instances against
# the f32-input ones (C3, configs[1], configs[4] shard).
export TMPDIR=/tmp
bash scripts/gpu_final.sh
bash scripts/gpu_pmc.sh
bash scripts/gpu_pmc_mfma.sh
bash scripts/gpu_pmc_sq.sh
bash scripts/gpu_pmc_sq_c5.sh
rm -rf gpurun_out/pmc_mfma_c5
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d gpurun_out/pmc_mfma_c5 -o p --output-format csv -- python bench.py --config c5 --steps 1 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence > gpurun_out/pmc_mfma_c5.json 2> gpurun_out/pmc_mfma_c5.err
python - <<'PY'
import csv, glob, collections, os
files = glob.glob("gpurun_out/pmc_mfma_c5/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_mfma_c5_summary.txt", "w") as out:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0])))[:8]:
        n = len(d.get("GRBM_GUI_ACTIVE", []))
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        line = f"{k:50s} n={n:4d} " + " ".join(f"{c}={mean[c]:.4g}" for c in sorted(mean))
        print(line); out.write(line + "\n")
for f in files:
    if os.path.getsize(f) > 4_000_000: os.remove(f)
PY
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
run() {
  local label=$1 cfg=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $cfg --steps $steps --warmup 2 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/w.json 2> gpurun_out/w.err
  python - "$label" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/w.json").read().splitlines()[0])
    print(sys.argv[1], sys.argv[2], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 3), "ms", d.get("kernel_ms_per_step"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex, open("gpurun_out/w.err").read()[-500:])
PY
}
for rep in 1 2; do
  for cs in c3:10 c2:40 c5:3; do
    run default ${cs%%:*} ${cs##*:} X=1
    run sweep_f32_input ${cs%%:*} ${cs##*:} HMX_ROUND_F32=1
    run rtz_f32_input ${cs%%:*} ${cs##*:} HMX_RTZ3_BF16=0
  done
done 2>&1 | tee gpurun_out/ab_bf16_pipe_r5.txt
