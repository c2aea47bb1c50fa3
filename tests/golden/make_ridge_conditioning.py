#!/usr/bin/env python3
"""Evidence for DESIGN.md §5 "the reference's fp32 ridge is ill-conditioned at large N".

Runs ONLY in the build container (imports /root/reference, harmonypy v0.2.0, device='cpu').
BASELINE configs[1] shape (69k cells x 50 PCs, 4 batches, K=30), identical initial centroids (the
sklearn fit is replaced by a fixed Y0), 5 k-means rounds (epsilon_cluster=0), ONE ridge correction.
The reference is run with 1 and with 8 torch threads, and once with harmony.py:553's inverse
evaluated in float64.  Writes tests/golden/ridge_conditioning.json:

  R_relF_1_vs_8_threads        -- what reaches the ridge step differs by fp32 summation noise only
  Zcorr_relF_1_vs_8_threads    -- ... and the fp32 ridge turns that into this
  Zcorr_relF_f32_vs_f64_inverse
  cond_cov_median / max        -- condition number of cov (harmony.py:550) over the K clusters

    python tests/golden/make_ridge_conditioning.py
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import logging  # noqa: E402
import torch  # noqa: E402
import harmonypy as hm  # noqa: E402
import harmonypy.harmony as hh  # noqa: E402
from bench import quick_centroids, synthetic_dataset  # noqa: E402

logging.getLogger("harmonypy").setLevel(logging.WARNING)
_state = {"Y0": None, "conds": []}


class _FixedKMeans:
    """Stands in for sklearn.KMeans at harmony.py:370-372: hands back the prepared centroids."""
    def __init__(self, *a, **k):
        pass

    def fit(self, X):
        self.cluster_centers_ = _state["Y0"].T.astype(np.float64)
        return self


hh.KMeans = _FixedKMeans
_inv = torch.linalg.inv


def run(threads, inv64=False, N=69_000, d=50, B=4, K=30, seed=0):
    torch.set_num_threads(threads)
    Z, meta = synthetic_dataset(N, d, B, K, seed=seed)
    if _state["Y0"] is None or _state["Y0"].shape != (d, K):
        _state["Y0"] = quick_centroids(Z, K, seed=seed, sample=20_000 if seed else 50_000)

    def inv(x):
        _state["conds"].append(float(np.linalg.cond(x.double().numpy())))
        return _inv(x.double()).float() if inv64 else _inv(x)
    torch.linalg.inv = inv
    try:
        ho = hm.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=1, max_iter_kmeans=5, epsilon_cluster=0.0,
                            epsilon_harmony=-1e30, verbose=False, random_state=0, device="cpu")
    finally:
        torch.linalg.inv = _inv
    return ho.R.copy(), ho.Z_corr.copy()


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def case(label, **shape):
    _state["conds"].clear()
    R1, Z1 = run(1, **shape)
    conds = list(_state["conds"])
    R8, Z8 = run(8, **shape)
    _, Z64 = run(1, inv64=True, **shape)
    return {
        "shape": label + "; 5 rounds + 1 ridge, same Y0, same randperm stream",
        "R_relF_1_vs_8_threads": rel(R8, R1),
        "Zcorr_relF_1_vs_8_threads": rel(Z8, Z1),
        "Zcorr_maxabs_over_max_1_vs_8_threads": float(np.abs(Z8 - Z1).max() / np.abs(Z1).max()),
        "Zcorr_relF_f32_vs_f64_inverse": rel(Z1, Z64),
        "Zcorr_maxabs_over_max_f32_vs_f64_inverse": float(np.abs(Z1 - Z64).max() / np.abs(Z64).max()),
        "cond_cov_median": float(np.median(conds)), "cond_cov_max": float(np.max(conds)),
    }


def main():
    out = {
        "reference": "harmonypy v0.2.0 at /root/reference, device='cpu', torch " + torch.__version__,
        "configs_1": case("69000 cells x 50 PCs, 4 batches, K=30 (BASELINE configs[1])"),
        # shape of tests/test_parity_gpu.py::test_bench_path_parity_c5_shape: lamb[0] = 0 (harmony.py:150-152) leaves
        # cov[0,0] = the cluster's mass, and K=200 clusters over 100 cell types leave many clusters nearly empty
        "configs_4_shape": case("40000 cells x 200 PCs, 32 batches, K=200 (BASELINE configs[4] shape)",
                                N=40_000, d=200, B=32, K=200, seed=3),
    }
    with open(os.path.join(HERE, "ridge_conditioning.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
