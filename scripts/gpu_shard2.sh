#!/bin/bash
# two ranks on ONE GPU (gloo + host transport + peer boxes through IPC): bench.py's own launcher and its default N > 1 workload
# (BASELINE configs[3]: 10 M cells split over the ranks, strong scaling) end to end, with and without the in-kernel exchange
export TMPDIR=/tmp
mkdir -p gpurun_out
export HMX_BENCH_BACKEND=gloo HMX_ROUND_WGS=100
for peer in 1 0; do
  HMX_PEER_EXCHANGE=$peer timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_2rank_peer$peer.json 2> gpurun_out/bench_2rank_peer$peer.err
  echo "peer=$peer rc=$?"
  python - $peer <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/bench_2rank_peer{sys.argv[1]}.json").read().splitlines()[0])
    print(d["n_gpus"], d["scaling"], round(d["value"] / 1e6, 2), "M", round(d["ms_per_step"], 2), "ms |", d["config"]["workload"][:90], "|", d["config"]["parallelism"][:120])
    print("ranks:", json.dumps(d.get("ranks"))[:600])
except Exception as ex:
    print("FAILED", ex, open(f"gpurun_out/bench_2rank_peer{sys.argv[1]}.err").read()[-600:])
PY
done
