#!/bin/bash
# round 4, call 1: full GPU suite (with the repeat-run stress tests of the wide path), a PC-sampling attempt on one C3 step,
# the default bench line of this box
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -rP > gpurun_out/r4_pytest1_full.log 2>&1
grep -E "passed|failed|error|Error|wide path|configs\[4\] shape|same schedule|diverged|bench path|2 shards|time-out replay|1 vs 2" gpurun_out/r4_pytest1_full.log | tail -60 > gpurun_out/r4_pytest1.log
tail -25 gpurun_out/r4_pytest1.log
for method in stochastic host_trap; do
  for unit_iv in "cycles 65536" "cycles 1048576" "time 100" "time 1"; do
    set -- $unit_iv
    rm -rf gpurun_out/pcs_$method
    timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $1 --pc-sampling-interval $2 \
        -d gpurun_out/pcs_$method -o r --output-format csv -- \
        python bench.py --config c3 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi --no-roofline > gpurun_out/pcs_$method.log 2>&1
    rc=$?
    echo "pc sampling $method $1 $2: rc=$rc" | tee -a gpurun_out/pcs_attempts.txt
    tail -4 gpurun_out/pcs_$method.log | tee -a gpurun_out/pcs_attempts.txt
    if ls gpurun_out/pcs_$method/*pc_sampling* gpurun_out/pcs_$method/*/*pc_sampling* >/dev/null 2>&1; then
      ls -la gpurun_out/pcs_$method gpurun_out/pcs_$method/* | head -20
      break 2
    fi
  done
done
# keep the sample files small enough for the 64 MiB merge: head of each
for f in $(find gpurun_out -name "*pc_sampling*csv"); do wc -l $f; python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
rows = csv.DictReader(open(f))
cnt = collections.Counter()
cols = None
for r in rows:
    if cols is None:
        cols = list(r.keys()); print(cols)
    key = tuple(r.get(k, "") for k in ("Code_Object_Id", "Code_Object_Offset", "Instruction", "Instruction_Comment") if k in r)
    cnt[key] += 1
with open(f + ".hist.txt", "w") as out:
    for k, v in cnt.most_common():
        out.write(f"{v}\t" + "\t".join(k) + "\n")
print(len(cnt), "distinct PCs", sum(cnt.values()), "samples")
PY
done
find gpurun_out -name "*pc_sampling*csv" -size +20M -delete
timeout 900 python bench.py > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench0.json").read().splitlines()[0])
print(round(d["value"] / 1e6, 2), "M cells/s/it", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us"))
print({k: (round(d[k]["value"] / 1e6, 2), round(d[k]["ms_per_step"], 2)) for k in ("configs_1", "configs_3_on_one_gpu", "configs_4_shard") if k in d})
PY
