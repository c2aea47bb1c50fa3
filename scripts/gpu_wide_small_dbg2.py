"""Debug aid: where the first NaN of the opt-in wide sweep appears on a small shape (one k-means round)."""
import os, sys
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import oracle_run_harmony
from test_parity_gpu import _run_engine
N, d, K, B, bs = 3000, 60, 150, 3, 0.05
rng = np.random.default_rng(N)
Z = rng.normal(size=(N, d)).astype(np.float32) * (1.0 / np.sqrt(1 + np.arange(d))).astype(np.float32)
batch = rng.integers(0, B, size=N); batch[:B] = np.arange(B)
Z += (batch[:, None] * 0.3).astype(np.float32)
meta = pd.DataFrame({"b": [f"b{i}" for i in batch]})
kw = dict(nclust=K, block_size=bs, max_iter_harmony=1, max_iter_kmeans=int(os.environ.get("MIK", "2")), random_state=1, epsilon_cluster=0.0, epsilon_harmony=-1e30)
oo = oracle_run_harmony(Z, meta, ["b"], **kw)
ho = _run_engine(Z, meta, ["b"], Y0=oo.Y0, **kw)
R = ho.R
nanrow = ~np.isfinite(R).all(axis=1)
print("rows with NaN:", int(nanrow.sum()), "of", N, "| NaN columns (clusters):", np.nonzero(~np.isfinite(R).all(axis=0))[0][:20], "count", int((~np.isfinite(R).all(axis=0)).sum()))
print("O NaN entries:", int((~np.isfinite(ho.O)).sum()), "of", ho.O.size, "| first NaN clusters in O:", np.unique(np.nonzero(~np.isfinite(ho.O))[0])[:20])
print("objective", ho.objective_kmeans[:4], "oracle", oo.objective_kmeans[:4], "counters", ho._engine.counters())
good = ~nanrow
if good.any():
    print("max |R - oracle| over finite rows:", float(np.abs(R[good] - oo.R.T[good]).max()))
