#!/bin/bash
# one configs[4]-shard round in the rocprofv3 kernel trace, persistent wide sweep and one launch per block: who runs beside what
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in 1 0; do
rm -rf /tmp/prof_c5
HMX_WIDE_SWEEP=$mode timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_c5 -o c5 --output-format csv -- python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline > /tmp/c5prof.out 2> /tmp/c5prof.err || tail -5 /tmp/c5prof.err
python - $mode <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/prof_c5/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = "k_sweep_wide3" if sys.argv[1] == "1" else "k_rtzw2b<13, 7, true>"
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
mid = idx[-4]
t0 = int(rows[mid]["Start_Timestamp"])
print(f"# HMX_WIDE_SWEEP={sys.argv[1]}: start us, end us, duration us, queue, kernel (0 = start of a {key} launch)")
n = 0
for r in rows[mid - 10:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s / 1e3 > 2600: break
    name = r["Kernel_Name"][:46]
    if "k_assign_wide3" in name:
        n += 1
        if n > 3 and n < 19: continue
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?')} {name}")
PY
done 2>&1 | tee gpurun_out/r6_c5_timeline.txt
