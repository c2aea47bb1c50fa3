#!/bin/bash
# Round-3 working check (2): occupancy print, GPU suite, default bench, k_round variant (per-wave slot sums), kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
HMX_DEBUG=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
# gate: the benchmarked shape first -- a fault here ends the script instead of burning the budget on core dumps
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "bench_path_parity_c3 or forced_schedule" > gpurun_out/pytest_gate.log 2>&1 || { tail -20 gpurun_out/pytest_gate.log; echo "GATE FAILED"; exit 1; }
tail -2 gpurun_out/pytest_gate.log
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail=12 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|FAILED|ERROR|Z_corr vs|time-out replay|2 shards, every|bench path" gpurun_out/pytest_gpu_full.log | tail -30 > gpurun_out/pytest_gpu.log
tail -16 gpurun_out/pytest_gpu.log
HMX_DEBUG=1 timeout 600 python bench.py --no-lisi > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; grep "k_rtz3<" gpurun_out/bench_default.err | head -3
HMX_LIB=$PWD/build/libhmx_v1.so timeout 300 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err
timeout 300 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r --output-format csv -- python bench.py --cpu-sample 0 --no-roofline --no-lisi --no-convergence > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
rows = []
for f in glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    agg[(r["Kernel_Name"][:56], wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("gpurun_out/kernel_stats_by_grid.txt", "w") as out:
    out.write(f"{'kernel':56s} {'workgroups':>10s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s}\n")
    for (k, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
        out.write(f"{k:56s} {wg:10d} {len(v):6d} {sum(v)/1e3:10.2f} {sum(v)/len(v):10.1f}\n")
print(open("gpurun_out/kernel_stats_by_grid.txt").read()[:2200])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_round<7, 13>")]
if idx:
    mid = idx[len(idx) // 2]
    t0 = int(rows[mid]["Start_Timestamp"])
    with open("gpurun_out/timeline_round.txt", "w") as out:
        for r in rows[mid - 10:mid + 12]:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            out.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?')} {r['Kernel_Name'][:50]}\n")
    print(open("gpurun_out/timeline_round.txt").read())
for name in ("bench_default", "bench_v0", "bench_v1"):
    try:
        d = json.loads(open(f"gpurun_out/{name}.json").read().splitlines()[0])
        print(name, round(d["value"] / 1e6, 2), "M cells/s/it", round(d["ms_per_step"], 3), "ms", d.get("roofline", {}).get("frac"), d.get("kernel_ms_total"))
        for k in ("configs_1", "configs_3_on_one_gpu", "configs_4_shard"):
            if k in d: print("   ", k, round(d[k]["value"] / 1e6, 2), round(d[k]["ms_per_step"], 2))
        if "convergence" in d: print("    conv", d["convergence"]["wall_s"], d["convergence"]["kmeans_rounds"])
    except Exception as e:
        print(name, "unreadable:", e)
PY
find gpurun_out/prof -name '*kernel_trace.csv' -size +8M -delete
tail -3 gpurun_out/bench_default.err
