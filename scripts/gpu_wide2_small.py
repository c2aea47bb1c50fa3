"""Debug aid: k_assign_wide2 on small cluster counts against the oracle -- needs a library in which launch_assign's `a.mt >= 8`
test is lifted and k_assign_wide2<1..7> are instantiated (hmx_kernels.hip, HMX_WIDE2_CASE)."""
import os, sys
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import oracle_run_harmony
from test_parity_gpu import _run_engine
from conftest import z_errors
for (N, d, K, B, bs) in [(1500, 70, 20, 2, 0.1), (1500, 70, 20, 2, 0.05), (1500, 70, 64, 2, 0.1), (1500, 70, 100, 2, 0.1), (1500, 70, 48, 2, 0.1),
                         (1500, 70, 20, 1, 0.1), (4000, 70, 20, 2, 0.1), (1500, 80, 20, 2, 0.1), (1500, 70, 32, 2, 0.1), (1500, 70, 33, 2, 0.1)]:
    rng = np.random.default_rng(N)
    Z = rng.normal(size=(N, d)).astype(np.float32) * (1.0 / np.sqrt(1 + np.arange(d))).astype(np.float32)
    batch = rng.integers(0, B, size=N)
    batch[:B] = np.arange(B)
    Z += (batch[:, None] * 0.3).astype(np.float32)
    meta = pd.DataFrame({"b": [f"b{i}" for i in batch]})
    kw = dict(nclust=K, block_size=bs, max_iter_harmony=int(os.environ.get("MIH", "2")), max_iter_kmeans=int(os.environ.get("MIK", "3")), random_state=1, epsilon_cluster=0.0, epsilon_harmony=-1e30)
    oo = oracle_run_harmony(Z, meta, ["b"], **kw)
    ho = _run_engine(Z, meta, ["b"], Y0=oo.Y0, **kw)
    Rg, Ro = ho.R, oo.R.T
    bad = np.abs(Rg - Ro).max(axis=1)
    print((N, d, K, B, bs), "mt", (K + 15) // 16, "Z_corr", ["%.1e" % v for v in z_errors(ho.Z_corr, oo.result())], "R rows off by > 1e-4:", int((bad > 1e-4).sum()),
          "first bad rows", np.nonzero(bad > 1e-4)[0][:8], "obj", ho.objective_kmeans[:3], oo.objective_kmeans[:3])
