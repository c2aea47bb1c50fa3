#!/bin/bash
# quick GPU loop: parity tests (fail fast) + C3/C2 bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --config c2 --steps 20 --warmup 2 --cpu-sample 0 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -12 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
for f in ['bench_c3','bench_c2']:
    try:
        d=json.loads(open(f'gpurun_out/{f}.json').read().splitlines()[0])
        print(f, round(d['value']/1e6,2),'M cells/s/it', round(d['ms_per_step'],3),'ms/step', {k:round(v/d['steps'],3) for k,v in d.get('kernel_ms_total',{}).items()})
    except Exception as ex:
        print(f, 'FAILED', ex); print(open(f'gpurun_out/{f}.err').read()[-1500:])
PY
