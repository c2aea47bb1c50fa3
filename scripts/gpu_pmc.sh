#!/bin/bash
# HBM traffic counters of the round kernels (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together);
# writes gpurun_out/pmc_hbm.json (copy to profiles/<round>_c3_pmc_hbm.json: bench.py reads the newest one whose
# engine_version matches the running kernels)
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
done
python - <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import harmonypy_amd
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    val[c] = {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
    for f in files:
        if os.path.getsize(f) > 4_000_000: os.remove(f)
kern = {}
for k in val["FETCH_SIZE"]:
    if k not in val["WRITE_SIZE"]:
        continue
    n, f = val["FETCH_SIZE"][k]
    _, w = val["WRITE_SIZE"][k]
    kern[k] = {"dispatches": n, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
               "hbm_bytes_uncorrected": (f + w) * 1024, "hbm_bytes_corrected": (2 * f + w) * 1024}
out = {"engine_version": harmonypy_amd.engine_version(),
       "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence (two separate passes, scripts/gpu_pmc.sh)",
       "config": "C3: 1M cells x 50 PCs, 8 batches, K=100, 1 MI355X (timed loop only)",
       "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (mean over dispatches); hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts half the bytes of 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section)",
       "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"] * kv[1]["dispatches"])[:12])}
# the same kernel's average duration in the rocprofv3 kernel trace of THIS build (scripts/gpu_final.sh, run before this script on the same
# box): bench.py quotes roofline.frac_trace from it
try:
    best = None
    for line in open("gpurun_out/kernel_stats_by_grid.txt"):
        if line.startswith("void k_round<7, 13, true>(RoundArgs)"):
            f = line.split()
            wgs, calls, avg = int(f[-4]), int(f[-3]), float(f[-1])
            if best is None or calls > best[1]:
                best = (wgs, calls, avg)
    if best:
        out["trace"] = {"k_round_avg_us": best[2], "workgroups": best[0], "calls": best[1],
                        "source": "rocprofv3 --kernel-trace --stats of `python bench.py --cpu-sample 0 --no-roofline` on the same box and build (kernel_stats_by_grid)"}
except OSError:
    pass
json.dump(out, open("gpurun_out/pmc_hbm.json", "w"), indent=1)
for k, r in list(out["kernels"].items())[:8]:
    print(f"{k:60s} n={r['dispatches']:4d} corrected MB = {r['hbm_bytes_corrected'] / 1e6:9.1f}")
PY
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
