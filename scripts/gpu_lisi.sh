#!/bin/bash
# LISI on the GPU: parity tests, then wall-clock at growing sizes
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_lisi.py -m gpu -x -q 2>&1 | tail -25
timeout 600 python - <<'PY'
import time, numpy as np, pandas as pd
import harmonypy_amd as hm
rng = np.random.default_rng(0)
for n, d in [(100_000, 50), (1_000_000, 50), (1_000_000, 20)]:
    cent = rng.normal(size=(50, d)) * 3
    X = cent[rng.integers(0, 50, n)] + rng.normal(size=(n, d))
    meta = pd.DataFrame({"batch": pd.Categorical.from_codes(rng.integers(0, 8, n), categories=[f"b{i}" for i in range(8)])})
    hm.compute_lisi(X[:2000], meta[:2000], ["batch"], 30)
    t = time.perf_counter()
    out = hm.compute_lisi(X, meta, ["batch"], 30)
    dt = time.perf_counter() - t
    print(f"LISI n={n} d={d}: {dt:.3f} s  ({n/dt/1e3:.1f} k cells/s)  mean LISI {out.mean():.4f}", flush=True)
PY
