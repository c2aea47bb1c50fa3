#!/bin/bash
# one round of a bench configuration in the rocprofv3 kernel trace:  gpu_r6_timeline.sh <config> <kernel name fragment> [env...]
export TMPDIR=/tmp
mkdir -p gpurun_out
cfg=$1; key=$2; shift 2
rm -rf /tmp/prof_tl
env "$@" timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_tl -o t --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline > /tmp/tl.out 2> /tmp/tl.err || tail -5 /tmp/tl.err
python - "$cfg" "$key" <<'PY' | tee gpurun_out/r6_timeline_$cfg.txt
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
mid = idx[-6]
t0 = int(rows[mid]["Start_Timestamp"])
print(f"# {sys.argv[1]}: start us, end us, duration us, queue, kernel (0 = start of a launch of {sys.argv[2]})")
for r in rows[mid - 8:mid + 22]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?')} {r['Kernel_Name'][:56]}")
PY
