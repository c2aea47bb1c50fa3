import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HMX_UPDATE_ORDER"] = "device"
from bench import quick_centroids, synthetic_dataset
from oracle.harmony_oracle import OracleHarmony, prepare_inputs
from test_parity_gpu import _device_perm_source, _run_engine
N, d, B, K = 150_000, 50, 8, 100
seed = 11
Z, meta = synthetic_dataset(N, d, B, K, seed=3)
Y0 = quick_centroids(Z, K, seed=3, sample=20_000)
p = prepare_inputs(Z, meta, ["batch"], nclust=K)
oo = OracleHarmony(p["Z"], p["phi"], p["Pr_b"], p["sigma"], p["theta"], p["lamb"], K=K, run=False,
                   perm_source=_device_perm_source(N, seed), forced_rounds=[5, 5], ridge_dtype=np.float64)
oo.init_cluster(seed, Y0)
ho = _run_engine(Z, meta, ["batch"], Y0=Y0, nclust=K, max_iter_harmony=0, random_state=seed)
for r in range(5):
    oo._forced_rounds = [1]; oo.cluster()
    ho.cluster(_rounds=1)
    f = lambda a, b: (a[-1] - b[-1]) / abs(b[-1])
    Oe, Oo = ho.O.astype(np.float64), oo.O.astype(np.float64)
    Ee, Eo = ho.E.astype(np.float64), oo.E.astype(np.float64)
    Rt = oo.R.astype(np.float64)
    O_exact = Rt @ oo.Phi.astype(np.float64).T
    E_exact = np.outer(Rt.sum(axis=1), oo.Pr_b.astype(np.float64))
    print(f"round {r}: total {f(ho.objective_kmeans, oo.objective_kmeans):.2e} dist {f(ho.objective_kmeans_dist, oo.objective_kmeans_dist):.2e} "
          f"ent {f(ho.objective_kmeans_entropy, oo.objective_kmeans_entropy):.2e} cross {f(ho.objective_kmeans_cross, oo.objective_kmeans_cross):.2e} | "
          f"O eng-orc {np.abs(Oe-Oo).max():.2e} O orc-exact {np.abs(Oo-O_exact).max():.2e} O eng-exact {np.abs(Oe-O_exact).max():.2e} | "
          f"E eng-orc {np.abs(Ee-Eo).max():.2e} E orc-exact {np.abs(Eo-E_exact).max():.2e} E eng-exact {np.abs(Ee-E_exact).max():.2e}")
    print("   values", ho.objective_kmeans_dist[-1], ho.objective_kmeans_entropy[-1], ho.objective_kmeans_cross[-1])
