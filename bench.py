#!/usr/bin/env python3
"""bench.py -- placeholder header, replaced below once the device-side update order lands."""
import numpy as np
import pandas as pd


def synthetic_dataset(N, d, B, K, seed=0):
    """Seeded PC-like matrix with batch offsets (SURVEY.md §8d / BASELINE.md §3):
    PC scale 1/sqrt(1+j); T = max(K//2, 5) cell-type centres N(0, 3^2)*scale; batch offsets
    N(0,1)*scale; batch proportions Dirichlet(5); unit noise*scale; float32; one categorical
    column ``batch`` with labels b0..b{B-1}."""
    rng = np.random.default_rng(seed)
    T = max(K // 2, 5)
    scale = (1.0 / np.sqrt(1.0 + np.arange(d))).astype(np.float32)
    centres = (rng.normal(0, 3.0, (T, d)) * scale).astype(np.float32)
    offsets = (rng.normal(0, 1.0, (B, d)) * scale).astype(np.float32)
    p = rng.dirichlet(5.0 * np.ones(B))
    batch = rng.choice(B, size=N, p=p).astype(np.int32)
    typ = rng.integers(0, T, size=N)
    Z = rng.standard_normal((N, d), dtype=np.float32)
    Z *= scale
    Z += centres[typ]
    Z += offsets[batch]
    labels = np.array([f"b{i}" for i in range(B)])
    meta = pd.DataFrame({"batch": pd.Categorical.from_codes(batch, categories=labels)})
    return Z, meta
