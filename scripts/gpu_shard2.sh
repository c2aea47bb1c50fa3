#!/bin/bash
# two ranks on ONE GPU (gloo + host transport + peer boxes): exercises bench.py's sharded path end to end
mkdir -p gpurun_out
export HMX_BENCH_BACKEND=gloo HMX_ROUND_WGS=100
for peer in 1 0; do
HMX_PEER_EXCHANGE=$peer timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_2rank_peer$peer.json 2> gpurun_out/bench_2rank_peer$peer.err
echo "peer=$peer rc=$?"; cut -c1-260 gpurun_out/bench_2rank_peer$peer.json; python - <<PY
import json
d=json.loads(open('gpurun_out/bench_2rank_peer$peer.json').read().splitlines()[0]); print(d['config']['parallelism'], {k:round(v/d['steps'],3) for k,v in d['kernel_ms_total'].items()})
PY
grep -i "error\|warn\|Traceback" gpurun_out/bench_2rank_peer$peer.err | head -5
done
