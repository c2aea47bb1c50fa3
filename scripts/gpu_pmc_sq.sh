#!/bin/bash
# Where the waves of the round kernels (k_round, k_rtz3c) spend their cycles: SQ wait / issue / active counters (own pass; no trace domains besides --kernel-trace)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/pmc_sq -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence > gpurun_out/pmc_sq.json 2> gpurun_out/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace -d gpurun_out/pmc_sq2 -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence > gpurun_out/pmc_sq2.json 2> gpurun_out/pmc_sq2.err
python - <<'PY'
import csv, glob, collections, os
for d0 in ("gpurun_out/pmc_sq", "gpurun_out/pmc_sq2"):
    files = glob.glob(d0 + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(d0 + "_summary.txt", "w") as out:
        for k, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:4]:
            mean = {c: sum(v) / len(v) for c, v in d.items()}
            line = f"{k:40s} n={len(next(iter(d.values()))):4d} " + " ".join(f"{c}={mean[c]:.4g}" for c in sorted(mean))
            print(line); out.write(line + "\n")
    for f in files:
        if os.path.getsize(f) > 4_000_000: os.remove(f)
PY
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
tail -2 gpurun_out/pmc_sq.err
