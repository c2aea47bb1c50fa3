#!/bin/bash
# HBM traffic counters of the round kernels (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together)
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  ls gpurun_out/pmc_$c | head
done
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    with open(f"gpurun_out/pmc_{c}_summary.txt", "w") as out:
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            line = f"{k:60s} dispatches={len(v):5d} mean_{c}={sum(v)/len(v):14.1f} total={sum(v):16.1f}"
            print(line); out.write(line + "\n")
    for f in files:
        import os
        if os.path.getsize(f) > 4_000_000: os.remove(f)
PY
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
