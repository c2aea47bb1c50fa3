#!/bin/bash
# Round-end evidence: GPU parity suite, smoke(), default bench line, rocprofv3 kernel stats of the same command.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 > gpurun_out/pytest_gpu_full.log
grep -E "passed|failed|error|same schedule|diverged|bench path|2 shards|Z_corr vs|time-out|1 vs 2|wide path|configs.4. shape|c3full|4 ranks" gpurun_out/pytest_gpu_full.log | tail -70 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r --output-format csv -- python bench.py --cpu-sample 0 --no-roofline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python - <<'PY'
# per-kernel averages split by grid size (the side configurations launch the same templates on other grids), and the
# timeline of one C3 round on the engine's streams
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    agg[(r["Kernel_Name"][:56], wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("gpurun_out/kernel_stats_by_grid.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0 --no-roofline  (C3 loop, run to convergence, LISI, side configurations)\n")
    out.write(f"{'kernel':56s} {'workgroups':>10s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s}\n")
    for (k, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
        out.write(f"{k:56s} {wg:10d} {len(v):6d} {sum(v)/1e3:10.2f} {sum(v)/len(v):10.1f}\n")
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
c3 = [r for r in rows if r["Kernel_Name"].startswith("void k_round<7, 13, true>")]
wgs_c3 = collections.Counter(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) for r in c3).most_common(1)[0][0] if c3 else -1   # (the C3 loop's grid: 224 with the classic tile map, ~243 with the group-affine one)
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_round<7, 13, true>") and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) == wgs_c3]
if idx:
    mid = idx[len(idx) // 3]
    t0 = int(rows[mid]["Start_Timestamp"])
    with open("gpurun_out/timeline_round.txt", "w") as out:
        out.write("# one C3 round in the rocprofv3 kernel trace: start us, end us, duration us, queue, kernel (0 = start of a k_round launch)\n")
        for r in rows[mid - 12:mid + 14]:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            out.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?')} {r['Kernel_Name'][:50]}\n")
PY
find gpurun_out/prof -name '*kernel_trace.csv' -size +8M -delete
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log
python - <<'PY'
import json, csv, glob
d = json.loads(open("gpurun_out/bench_default.json").read().splitlines()[0])
print(round(d["value"] / 1e6, 2), "M cells/s/it", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d.get("lisi", {}).get("seconds"), d["convergence"]["wall_s"])
print({k: (round(d[k]["value"] / 1e6, 2), round(d[k]["ms_per_step"], 2)) for k in ("configs_1", "configs_3_on_one_gpu", "configs_4_shard") if k in d})
f = glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/kernel_stats_summary.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0 --no-roofline  (default: C3, 5 steps of 10 rounds + ridge; run to convergence; LISI of 1M cells before/after; side configurations C2, 10M cells on one GPU, wide shard)\n")
    out.write(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}\n")
    for r in rows[:45]:
        out.write(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}\n")
print(open("gpurun_out/kernel_stats_by_grid.txt").read()[:2600])
print(open("gpurun_out/timeline_round.txt").read())
PY
