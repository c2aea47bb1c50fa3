#!/bin/bash
# MFMA utilisation counters of the round kernels (own pass; no trace domains besides --kernel-trace)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d gpurun_out/pmc_mfma -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence > gpurun_out/pmc_mfma.json 2> gpurun_out/pmc_mfma.err
python - <<'PY'
import csv, glob, collections, os
files = glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_mfma_summary.txt", "w") as out:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        n = len(d.get("GRBM_GUI_ACTIVE", []))
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        line = f"{k:50s} n={n:4d} " + " ".join(f"{c}={mean[c]:.4g}" for c in sorted(mean))
        print(line); out.write(line + "\n")
for f in files:
    if os.path.getsize(f) > 4_000_000: os.remove(f)
PY
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
tail -3 gpurun_out/pmc_mfma.err
