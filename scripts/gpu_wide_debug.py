"""Debug aid: the configs[4] shape on the persistent wide sweep against the oracle, printing every component."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("HMX_UPDATE_ORDER", "device")
from bench import quick_centroids, synthetic_dataset
from oracle.harmony_oracle import OracleHarmony, prepare_inputs
from test_parity_gpu import _device_perm_source, _run_engine

N, d, B, K = 40000, 200, 32, 200
NR = int(os.environ.get('NR', '2'))
seed = 11
Z, meta = synthetic_dataset(N, d, B, K, seed=3)
Y0 = quick_centroids(Z, K, seed=3, sample=20_000)
p = prepare_inputs(Z, meta, ["batch"], nclust=K)
oo = OracleHarmony(p["Z"], p["phi"], p["Pr_b"], p["sigma"], p["theta"], p["lamb"], K=K, run=False,
                   perm_source=_device_perm_source(N, seed), forced_rounds=[NR], ridge_dtype=np.float64)
oo.init_cluster(seed, Y0)
ho = _run_engine(Z, meta, ["batch"], Y0=Y0, nclust=K, max_iter_harmony=0, random_state=seed)
oo.cluster()
ho.cluster(_rounds=NR)
for name in ("objective_kmeans", "objective_kmeans_dist", "objective_kmeans_entropy", "objective_kmeans_cross"):
    print(name, getattr(ho, name)[:NR + 1], getattr(oo, name)[:NR + 1])
Rg, Ro = ho.R, oo.R.T
print("R relF", np.linalg.norm(Rg - Ro) / np.linalg.norm(Ro), "max", np.abs(Rg - Ro).max())
print("O max abs err", np.abs(ho.O - oo.O).max(), "O scale", np.abs(oo.O).max())
print("counters", ho._engine.counters() if hasattr(ho._engine, "counters") else None)
bad = np.abs(Rg - Ro).max(axis=1)
print("rows with error > 1e-3:", int((bad > 1e-3).sum()), "of", N)

from oracle.device_order import positions
pos = positions(np.arange(N), N, seed, NR - 1)          # positions of the last round
cpb = int(np.ceil(N * 0.05))
blk = pos // cpb
codes = meta["batch"].astype("category").cat.codes.to_numpy()
wrong = bad > 1e-3
print("wrong rows per block of the last round:", np.bincount(blk[wrong], minlength=20))
print("wrong rows per batch:", np.bincount(codes[wrong], minlength=B))
print("cells per block:", np.bincount(blk, minlength=20)[:5])
