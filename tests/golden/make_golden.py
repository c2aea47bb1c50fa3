#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container, where the upstream tree is mounted at
/root/reference (harmonypy v0.2.0).  It imports that package, always with
``device='cpu'``, spies on it (no reference code is copied) and stores inputs,
initial centroids and outputs as small .npz files.  Nothing under tests/,
bench.py or __graft_entry__.py reads /root/reference at run time; they read the
files written here.

    python tests/golden/make_golden.py

Every case stores: the run_harmony kwargs (JSON), the sklearn centroids the
reference drew (``Y0``, d x K -- sklearn's result moves by ~2e-7 with the host
thread count, so it is part of the fixture), ``Z_corr``, the history lists and
the convergence margins  ``(|obj_old-obj_new|/|obj_old|)/epsilon_cluster`` at
every type-0 decision (harmony.py:523).  A margin within a few percent of 1.0
means the round count of that case is decided by fp32 rounding noise.
"""
import json
import os
import sys

import numpy as np
import pandas as pd

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import logging  # noqa: E402
import torch  # noqa: E402
import harmonypy as hm  # noqa: E402
import harmonypy.harmony as hh  # noqa: E402

logging.getLogger("harmonypy").setLevel(logging.WARNING)

# ---- spies -----------------------------------------------------------------
_captured = {}
_KMeans = hh.KMeans


class _SpyKMeans(_KMeans):
    def fit(self, X, *a, **k):
        out = super().fit(X, *a, **k)
        _captured["Y0"] = np.asarray(self.cluster_centers_.T, dtype=np.float32).copy()
        return out


hh.KMeans = _SpyKMeans
_steps = []
_orig = {n: getattr(hh.Harmony, n) for n in ("init_cluster", "update_R", "moe_correct_ridge")}


def _wrap(name, grab):
    def fn(self, *a, **k):
        r = _orig[name](self, *a, **k)
        if _captured.get("record_steps"):
            _steps.append((name, grab(self)))
        return r
    return fn


def _np(t):
    return t.detach().cpu().numpy().copy()


hh.Harmony.init_cluster = _wrap("init_cluster", lambda s: dict(
    R=_np(s._R).T, O=_np(s._O), E=_np(s._E), Y=_np(s._Y), obj=list(s.objective_kmeans)))
hh.Harmony.update_R = _wrap("update_R", lambda s: dict(
    R=_np(s._R).T, O=_np(s._O), E=_np(s._E), Y=_np(s._Y)))
hh.Harmony.moe_correct_ridge = _wrap("moe_correct_ridge", lambda s: dict(
    Z_corr=_np(s._Z_corr).T, Z_cos=_np(s._Z_cos).T))


def margins(objk, rounds, eps, window=3):
    """Type-0 decision ratios / eps at every check the reference made."""
    out, pos = [], 1
    for r in rounds:
        for i in range(window + 1, r):
            lst = objk[:pos + i + 1]
            old, new = sum(lst[-window - 1:-1]), sum(lst[-window:])
            out.append(abs(old - new) / abs(old) / eps if eps > 0 else float("inf"))
        pos += r
    return out


def run_case(name, data, meta, vars_use, steps=False, **kw):
    _steps.clear()
    _captured["record_steps"] = steps
    ho = hm.run_harmony(data, meta, vars_use, verbose=False, device="cpu", **kw)
    eps = kw.get("epsilon_cluster", 1e-5)
    mg = margins(ho.objective_kmeans, ho.kmeans_rounds, eps)
    rs = kw.get("random_state", 0)
    torch.manual_seed(rs)
    first_perm = torch.randperm(meta.shape[0]).numpy()
    out = dict(
        kwargs=json.dumps(kw), vars_use=json.dumps(vars_use),
        Y0=_captured["Y0"], Z_corr=np.ascontiguousarray(ho.Z_corr),
        kmeans_rounds=np.asarray(ho.kmeans_rounds),
        objective_harmony=np.asarray(ho.objective_harmony),
        objective_kmeans=np.asarray(ho.objective_kmeans),
        objective_kmeans_dist=np.asarray(ho.objective_kmeans_dist),
        objective_kmeans_entropy=np.asarray(ho.objective_kmeans_entropy),
        objective_kmeans_cross=np.asarray(ho.objective_kmeans_cross),
        margins=np.asarray(mg), O=ho.O, E=ho.E, Y=ho.Y,
        R_colsum=ho.R.sum(axis=0), theta=ho.theta, lamb=ho.lamb, sigma=ho.sigma, Pr_b=ho.Pr_b,
        first_perm_head=first_perm[:64],
        first_perm_crc=np.asarray([int(np.bitwise_xor.reduce(first_perm * np.arange(1, len(first_perm) + 1)))]),
    )
    if steps:
        for i, (nm, d) in enumerate(_steps):
            for k, v in d.items():
                out[f"step{i:03d}.{nm}.{k}"] = np.asarray(v)
        out["n_steps"] = np.asarray([len(_steps)])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    near = min((abs(m - 1) for m in mg), default=float("inf"))
    print(f"{name:28s} rounds={ho.kmeans_rounds} iters={len(ho.objective_harmony)-1} "
          f"min|margin-1|={near:.3f}  {os.path.getsize(path)/1e3:.0f} kB")


def synthetic(N, d, B, T, seed):
    """Seeded PC-like matrix with batch offsets (generator of SURVEY.md §8d)."""
    rng = np.random.default_rng(seed)
    scale = 1.0 / np.sqrt(1.0 + np.arange(d))
    centres = rng.normal(0, 3.0, (T, d)) * scale
    offsets = rng.normal(0, 1.0, (B, d)) * scale
    p = rng.dirichlet(5.0 * np.ones(B))
    batch = rng.choice(B, size=N, p=p)
    typ = rng.integers(0, T, size=N)
    Z = centres[typ] + offsets[batch] + rng.normal(0, 1.0, (N, d)) * scale
    return Z.astype(np.float32), batch


def main():
    meta = pd.read_csv(f"{REF}/data/pbmc_3500_meta.tsv.gz", sep="\t")
    pcs = pd.read_csv(f"{REF}/data/pbmc_3500_pcs.tsv.gz", sep="\t")
    harm = pd.read_csv(f"{REF}/data/pbmc_3500_pcs_harmonized.tsv.gz", sep="\t")
    rng = np.random.default_rng(20240925)
    tech = rng.choice(np.array(["v1", "v2", "v3", "v4"]), size=len(meta), p=[0.4, 0.3, 0.2, 0.1])
    meta2 = meta[["donor"]].copy()
    meta2["tech"] = tech
    np.savez_compressed(
        os.path.join(HERE, "pbmc_3500_inputs.npz"),
        pcs=pcs.to_numpy().astype(np.float32), donor=meta["donor"].to_numpy().astype("U1"),
        tech=tech.astype("U2"), r_harmonized=harm.to_numpy().astype(np.float32))

    # --- pbmc_3500, one variable ------------------------------------------------
    run_case("pbmc_default", pcs, meta, ["donor"])
    run_case("pbmc_seed7", pcs, meta, ["donor"], random_state=7)
    run_case("pbmc_short", pcs, meta, ["donor"], max_iter_harmony=2, max_iter_kmeans=2, random_state=42)
    run_case("pbmc_fixed_schedule", pcs, meta, ["donor"], max_iter_harmony=3, max_iter_kmeans=6,
             epsilon_cluster=0.0, epsilon_harmony=-1e30, random_state=5)
    run_case("pbmc_lambda_est", pcs, meta, ["donor"], lamb=-1, max_iter_harmony=3, random_state=1)
    run_case("pbmc_theta_tau", pcs, meta, ["donor"], theta=1.0, tau=5, sigma=0.2, nclust=20,
             max_iter_harmony=3, random_state=2)
    # --- pbmc_3500, two variables (multi-hot Phi) -------------------------------
    run_case("pbmc_two_vars", pcs, meta2, ["donor", "tech"], theta=[2.0, 1.0], lamb=[1.0, 0.5],
             max_iter_harmony=3, random_state=3)
    # --- small synthetic, ragged last block, step-level arrays -----------------
    Zs, bs = synthetic(1237, 12, 4, 5, seed=11)
    metas = pd.DataFrame({"batch": np.array([f"b{i}" for i in bs])})
    np.savez_compressed(os.path.join(HERE, "synth_small_inputs.npz"), Z=Zs, batch=bs.astype(np.int32))
    run_case("synth_small_steps", Zs, metas, ["batch"], steps=True, nclust=7, block_size=0.07,
             max_iter_harmony=2, max_iter_kmeans=4, random_state=9)
    run_case("synth_small_default", Zs, metas, ["batch"], nclust=7, random_state=4)
    run_case("synth_small_lambda_est", Zs, metas, ["batch"], nclust=7, lamb=-1, theta=[1.5],
             max_iter_harmony=3, random_state=6)


if __name__ == "__main__":
    main()
