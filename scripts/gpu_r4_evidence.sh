#!/bin/bash
# Round-4 evidence on ONE box: suite + smoke + default bench + rocprofv3 kernel stats (gpu_final.sh), then the counter passes
# for C3 (HBM, MFMA, SQ) and the configs[4] shard, then the cycle stamps of the profiling build.
export TMPDIR=/tmp
bash scripts/gpu_final.sh
bash scripts/gpu_pmc.sh
bash scripts/gpu_pmc_mfma.sh
bash scripts/gpu_pmc_sq.sh
if [ -z "$SKIP_C5" ]; then   # (SKIP_C5=1: the configs[4] passes, when the wide kernels have not changed)
bash scripts/gpu_pmc_sq_c5.sh
rm -rf gpurun_out/pmc_mfma_c5
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d gpurun_out/pmc_mfma_c5 -o p --output-format csv -- python bench.py --config c5 --steps 1 --warmup 1 --cpu-sample 0 --no-roofline --no-convergence > gpurun_out/pmc_mfma_c5.json 2> gpurun_out/pmc_mfma_c5.err
python - <<'PY'
import csv, glob, collections, os
files = glob.glob("gpurun_out/pmc_mfma_c5/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_mfma_c5_summary.txt", "w") as out:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0])))[:8]:
        n = len(d.get("GRBM_GUI_ACTIVE", []))
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        line = f"{k:50s} n={n:4d} " + " ".join(f"{c}={mean[c]:.4g}" for c in sorted(mean))
        print(line); out.write(line + "\n")
for f in files:
    if os.path.getsize(f) > 4_000_000: os.remove(f)
PY
fi
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]" | tee gpurun_out/stamps_c3.txt
[ -z "$SKIP_C5" ] && HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]" | tee gpurun_out/stamps_c5.txt
timeout 200 python bench.py --config c2 --steps 20 --warmup 5 --cpu-sample 0 --no-convergence --no-lisi > gpurun_out/bench_c2_alone.json 2> gpurun_out/bench_c2_alone.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_c2_alone.json').read().splitlines()[0]); print('c2 alone', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms')"
# the two instances of k_round on this box: distance GEMM on the bf16 pipe (default) and the f32-input form (HMX_ROUND_F32=1)
CFGS="c3 c2" bash scripts/gpu_r4_ab.sh "" default f32 2>&1 | tee gpurun_out/ab_bf16_pipe.txt
