#!/bin/bash
# MFMA utilisation / wait counters of the LISI kernels (own pass; no trace domains besides --kernel-trace)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_lisi
cat > gpurun_out/lisi_run.py <<'PY'
import numpy as np, pandas as pd, time
import harmonypy_amd as hm
rng = np.random.default_rng(0)
n, d = 400_000, 50
cent = rng.normal(size=(50, d)) * 3
X = cent[rng.integers(0, 50, n)] + rng.normal(size=(n, d))
meta = pd.DataFrame({"batch": pd.Categorical.from_codes(rng.integers(0, 8, n), categories=[f"b{i}" for i in range(8)])})
t = time.perf_counter(); hm.compute_lisi(X, meta, ["batch"], 30); print("lisi 400k", time.perf_counter() - t)
PY
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_lisi/$tag -o p --output-format csv -- env PYTHONPATH=$PWD python gpurun_out/lisi_run.py > gpurun_out/pmc_lisi_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
files = glob.glob("gpurun_out/pmc_lisi/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_lisi_summary.txt", "w") as out:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        line = f"{k:50s} " + " ".join(f"{c}={mean[c]:.4g}" for c in sorted(mean))
        print(line); out.write(line + "\n")
for f in files:
    if os.path.getsize(f) > 2_000_000: os.remove(f)
PY
find gpurun_out/pmc_lisi -name '*kernel_trace.csv' -size +1M -delete
for f in gpurun_out/pmc_lisi_*.log; do tail -n 2 $f; done
