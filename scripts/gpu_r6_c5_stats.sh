#!/bin/bash
# rocprofv3 kernel statistics of the configs[4] shard alone (the default bench line measures it in a process of its own, which the trace of the default command does not follow)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_c5
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o r --output-format csv -- python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline > gpurun_out/c5_prof.json 2> gpurun_out/c5_prof.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_c5/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/c5_kernel_stats.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline  (configs[4] shard: 1.25 M cells x 200 PCs, K = 200, 32 batches; 4 steps of 10 rounds + ridge)\n")
    out.write(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}\n")
    for r in rows[:30]:
        out.write(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}\n")
print(open("gpurun_out/c5_kernel_stats.txt").read())
PY
find gpurun_out/prof_c5 -name '*kernel_trace.csv' -size +8M -delete
