#!/usr/bin/env python3
"""The benchmarked size on the benchmarked path, pinned to the REFERENCE ITSELF.

Runs ONLY in the build container (imports /root/reference, harmonypy v0.2.0, device='cpu').  BASELINE configs[2] exactly as
bench.py times it: synthetic_dataset(1_000_000, 50, 8, 100, seed=0, cell_seed=0), initial centroids quick_centroids(seed=0),
and -- the difference to make_ridge_conditioning.py -- the UPDATE ORDER OF THE ENGINE'S LARGE-JOB PATH: `torch.randperm`
(harmony.py:471) is replaced by the keyed bijection the GPU evaluates (oracle/device_order.positions; seed 0 = bench.py's
random_state, round counter 0, 1, ...), so the reference walks the very blocks hmx_cluster builds on the device.  5 k-means
rounds (epsilon_cluster=0) + ONE ridge correction, twice: (i) the plain reference, (iii) the reference's own
moe_correct_ridge on float64 copies of its tensors (tests/golden/make_ridge_conditioning.py explains why (iii) is the pin
of Z_corr at this size).  Writes tests/golden/large_c3full.npz (2000 evenly spaced rows of R and of both Z_corr, O, E, the
column sums of R, the four objective histories, Y0) -- about 2 MB; takes about 10 minutes and 10 GB here.

    python tests/golden/make_c3_full.py
"""
import os
import sys
import time

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
from oracle.device_order import positions  # noqa: E402

logging.getLogger("harmonypy").setLevel(logging.WARNING)
N, D, B, K, SEED, ROUNDS, SAMPLE_ROWS = 1_000_000, 50, 8, 100, 0, 5, 2000
_state = {"Y0": None, "counter": 0}


class _FixedKMeans:
    """Stands in for sklearn.KMeans at harmony.py:370-372: hands back the prepared centroids."""
    def __init__(self, *a, **k):
        pass

    def fit(self, X):
        self.cluster_centers_ = _state["Y0"].T.astype(np.float64)
        return self


def _device_order_randperm(n, *a, **k):
    """harmony.py:471 with the engine's device order: position p of round r holds the cell whose keyed-bijection
    position is p (the same construction tests/test_parity_gpu.py::_device_perm_source feeds the oracle)."""
    assert n == N
    pos = positions(np.arange(n), n, SEED, _state["counter"])
    _state["counter"] += 1
    return torch.from_numpy(np.argsort(pos, kind="stable"))


_ridge = hh.Harmony.moe_correct_ridge


def _ridge_in_float64(self):
    names = ["_Z_orig", "_R", "_Phi_moe", "_lamb", "_E"]
    saved = {n: getattr(self, n) for n in names}
    for n in names:
        setattr(self, n, saved[n].double())
    try:
        _ridge(self)
    finally:
        for n in names:
            setattr(self, n, saved[n])
    self._Z_corr = self._Z_corr.float()
    self._Z_cos = self._Z_cos.float()


def run(Z, meta, ridge64):
    _state["counter"] = 0
    hh.KMeans = _FixedKMeans
    hh.Harmony.moe_correct_ridge = _ridge_in_float64 if ridge64 else _ridge
    real = torch.randperm
    torch.randperm = _device_order_randperm
    t0 = time.time()
    try:
        ho = hm.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=1, max_iter_kmeans=ROUNDS, epsilon_cluster=0.0,
                            epsilon_harmony=-1e30, verbose=False, random_state=0, device="cpu")
    finally:
        torch.randperm = real
        hh.Harmony.moe_correct_ridge = _ridge
    assert _state["counter"] == ROUNDS, _state["counter"]
    print(f"reference run (ridge64={ridge64}): {time.time() - t0:.0f} s, rounds {ho.kmeans_rounds}", flush=True)
    return ho


def main():
    torch.set_num_threads(8)
    Z, meta = synthetic_dataset(N, D, B, K, seed=0, cell_seed=0)
    _state["Y0"] = quick_centroids(Z, K, seed=0)
    rows = np.linspace(0, N - 1, SAMPLE_ROWS).astype(np.int64)
    ho = run(Z, meta, ridge64=False)
    R1 = ho.R
    out = dict(
        shape=np.array([N, D, B, K, SEED, ROUNDS], dtype=np.int64), Y0=_state["Y0"].astype(np.float32), rows=rows,
        R_rows=R1[rows].astype(np.float32), R_colsum=R1.astype(np.float64).sum(axis=0),
        O=ho.O.astype(np.float32), E=ho.E.astype(np.float32),
        objective_kmeans=np.asarray(ho.objective_kmeans, dtype=np.float64),
        objective_kmeans_dist=np.asarray(ho.objective_kmeans_dist, dtype=np.float64),
        objective_kmeans_entropy=np.asarray(ho.objective_kmeans_entropy, dtype=np.float64),
        objective_kmeans_cross=np.asarray(ho.objective_kmeans_cross, dtype=np.float64),
        kmeans_rounds=np.asarray(ho.kmeans_rounds, dtype=np.int64),
        Zcorr_rows_plain=ho.Z_corr[rows].astype(np.float32),
        Zcorr_norm_plain=np.float64(np.linalg.norm(ho.Z_corr.astype(np.float64))))
    Z1 = ho.Z_corr.copy()
    del ho, R1
    ho = run(Z, meta, ridge64=True)
    Zr = ho.Z_corr
    out.update(Zcorr_rows_ridge64=Zr[rows].astype(np.float32), Zcorr_absmax=np.float64(np.abs(Zr).max()),
               Zcorr_norm_ridge64=np.float64(np.linalg.norm(Zr.astype(np.float64))),
               R_rows_relF_between_the_two_runs=np.float64(np.linalg.norm(ho.R[rows].astype(np.float64) - out["R_rows"]) /
                                                            np.linalg.norm(out["R_rows"].astype(np.float64))),
               plain_vs_ridge64_relF=np.float64(np.linalg.norm(Z1.astype(np.float64) - Zr) / np.linalg.norm(Zr.astype(np.float64))))
    np.savez_compressed(os.path.join(HERE, "large_c3full.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
