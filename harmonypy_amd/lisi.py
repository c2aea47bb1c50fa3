"""``compute_lisi`` with the reference's signature (harmonypy/lisi.py:24-66), computed on the MI355X.

The neighbour search (the reference: sklearn's kd-tree, lisi.py:53-54), the perplexity search and
the Simpson index (``compute_simpson``, lisi.py:69-133) run as HIP kernels behind
``hmx_compute_lisi`` (include/hmx.h).  There is no CPU path.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np
import pandas as pd

from . import _capi


def _device_index(device):
    if device is None or device == "cuda":
        return 0
    s = str(device)
    if s.startswith("cuda:"):
        return int(s.split(":", 1)[1])
    raise ValueError(f"harmonypy_amd runs on the MI355X only (device='cuda' or 'cuda:n'), got {device!r}")


def compute_lisi(
    X: np.array,
    metadata: pd.DataFrame,
    label_colnames: Iterable[str],
    perplexity: float = 30,
    device=None,
    return_neighbors: bool = False,
):
    """Local Inverse Simpson Index of every row of ``X`` for each column of ``metadata`` named in
    ``label_colnames``; returns an ``n_cells x n_labels`` float64 array like the reference.

    ``return_neighbors=True`` also returns the ``3*perplexity - 1`` nearest neighbours of every cell
    (distances, indices), nearest first -- what the reference gets from ``knn.kneighbors`` after
    dropping the first column (lisi.py:55-60).

    Limit of this build: ``3 * perplexity <= 2040`` neighbours (perplexity <= 680; the reference takes any): a larger
    value raises ``ValueError``.  Up to 120 neighbours (perplexity 40; the default is 30) the search keeps 256
    candidates per cell, up to 504 it keeps 1024, beyond 4096 -- each step costs memory (8 bytes x candidates x cells)
    and speed.
    """
    if isinstance(label_colnames, str):
        label_colnames = [label_colnames]
    label_colnames = list(label_colnames)
    Xv = X.values if hasattr(X, "values") else X
    Xv = np.ascontiguousarray(Xv, dtype=np.float64)
    if Xv.ndim != 2:
        raise ValueError("X must be a cells x features matrix")
    n, d = Xv.shape
    if metadata.shape[0] != n:
        raise ValueError("X and metadata do not have the same number of cells")
    codes = np.empty((len(label_colnames), n), dtype=np.int32)
    for i, label in enumerate(label_colnames):
        cat = pd.Categorical(metadata[label])                                   # lisi.py:63
        if (cat.codes < 0).any():
            raise ValueError(f"metadata[{label!r}] has missing values")
        codes[i] = cat.codes
    nn = int(perplexity * 3)                                                    # lisi.py:53
    lib = _capi.load()
    out = np.empty((n, len(label_colnames)), dtype=np.float64)
    kd = ki = None
    if return_neighbors:
        kd = np.empty((n, max(nn - 1, 0)), dtype=np.float64)
        ki = np.empty((n, max(nn - 1, 0)), dtype=np.int32)
    rc = lib.hmx_compute_lisi(_device_index(device), _capi._ptr(Xv), n, d, _capi._ptr(codes), len(label_colnames),
                              float(perplexity), _capi._ptr(out), _capi._ptr(kd), _capi._ptr(ki))
    if rc < 0:
        msg = lib.hmx_last_error().decode(errors="replace")
        if "n_neighbors" in msg or msg.startswith("perplexity"):
            raise ValueError(msg)                                               # what sklearn raises for the reference / this build's limit
        raise _capi.HmxError(f"libhmx: {msg} (code {rc})", code=int(rc))
    return (out, kd, ki) if return_neighbors else out
