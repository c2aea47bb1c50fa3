#!/bin/bash
# LISI wall-clock only (1M x 50); HMX_LIB selects a variant build
python - <<'PY'
import time, numpy as np, pandas as pd
import harmonypy_amd as hm
rng = np.random.default_rng(0)
for n, d in [(1_000_000, 50)]:
    cent = rng.normal(size=(50, d)) * 3
    X = cent[rng.integers(0, 50, n)] + rng.normal(size=(n, d))
    meta = pd.DataFrame({"batch": pd.Categorical.from_codes(rng.integers(0, 8, n), categories=[f"b{i}" for i in range(8)])})
    hm.compute_lisi(X[:2000], meta[:2000], ["batch"], 30)
    t = time.perf_counter()
    out = hm.compute_lisi(X, meta, ["batch"], 30)
    dt = time.perf_counter() - t
    print(f"LISI n={n} d={d}: {dt:.3f} s", flush=True)
PY
