#!/bin/bash
# Round-3 working check: smoke, the whole GPU suite, the default bench line, the same with the list-order R^T.Z kernel
# (HMX_RTZ=2, A/B), rocprofv3 kernel statistics of the C3 loop.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail=12 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|FAILED|ERROR|engine .*Z_corr|time-out replay|2 shards, every|bench path|same schedule|diverged" gpurun_out/pytest_gpu_full.log | tail -70 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
HMX_RTZ=2 timeout 300 python bench.py --no-lisi --no-convergence --cpu-sample 0 > gpurun_out/bench_rtz2.json 2> gpurun_out/bench_rtz2.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r --output-format csv -- python bench.py --cpu-sample 0 --no-roofline --no-lisi --no-convergence > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        agg[(r["Kernel_Name"][:56], wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("gpurun_out/kernel_stats_by_grid.txt", "w") as out:
    out.write(f"{'kernel':56s} {'workgroups':>10s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s}\n")
    for (k, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
        out.write(f"{k:56s} {wg:10d} {len(v):6d} {sum(v)/1e3:10.2f} {sum(v)/len(v):10.1f}\n")
print(open("gpurun_out/kernel_stats_by_grid.txt").read()[:2600])
for name in ("bench_default", "bench_rtz2"):
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
tail -5 gpurun_out/bench_default.err
