import os, sys
import numpy as np, pandas as pd
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import oracle_run_harmony
from test_parity_gpu import _run_engine
for (N, d, K, B, bs) in [(3000, 60, 150, 3, 0.05), (640, 64, 120, 2, 0.05)]:
    rng = np.random.default_rng(N)
    Z = rng.normal(size=(N, d)).astype(np.float32) * (1.0 / np.sqrt(1 + np.arange(d))).astype(np.float32)
    batch = rng.integers(0, B, size=N); batch[:B] = np.arange(B)
    Z += (batch[:, None] * 0.3).astype(np.float32)
    meta = pd.DataFrame({"b": [f"b{i}" for i in batch]})
    kw = dict(nclust=K, block_size=bs, max_iter_harmony=1, max_iter_kmeans=2, random_state=1, epsilon_cluster=0.0, epsilon_harmony=-1e30)
    oo = oracle_run_harmony(Z, meta, ["b"], **kw)
    ho = _run_engine(Z, meta, ["b"], Y0=oo.Y0, **kw)
    Rg, Ro = ho.R, oo.R.T
    bad = np.abs(Rg - Ro).max(axis=1)
    print((N, d, K, B, bs), "R rows off > 1e-4:", int((bad > 1e-4).sum()), "obj", np.round(ho.objective_kmeans[:3], 3), np.round(oo.objective_kmeans[:3], 3),
          "O err", float(np.abs(ho.O - oo.O).max()), "counters", ho._engine.counters(), "rounds", ho.kmeans_rounds, "finite R", bool(np.isfinite(Rg).all()), "cross", np.round(ho.objective_kmeans_cross[:3], 3), np.round(oo.objective_kmeans_cross[:3], 3))
