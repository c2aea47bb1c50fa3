"""Golden vectors for LISI, made by running the REFERENCE here (harmonypy.compute_lisi, lisi.py).

    python tests/golden/make_lisi_golden.py      # needs /root/reference; writes tests/golden/lisi_*.npz

lisi_ref_fixture.npz : the reference's own known-answer test (tests/test_lisi.py:5-17): its inputs
                       data/lisi_x.tsv.gz, data/lisi_metadata.tsv.gz, its stored answer
                       data/lisi_lisi.tsv.gz, and what the reference computes from them here.
lisi_pbmc.npz        : the first 10 PCs of pbmc_3500 with the donor label and a synthetic 5-level label.
"""
import os
import sys

import numpy as np
import pandas as pd

REF = "/root/reference"
sys.path.insert(0, REF)
import harmonypy as hm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    X = pd.read_csv(f"{REF}/data/lisi_x.tsv.gz", sep="\t")
    meta = pd.read_csv(f"{REF}/data/lisi_metadata.tsv.gz", sep="\t")
    stored = pd.read_csv(f"{REF}/data/lisi_lisi.tsv.gz", sep="\t").iloc[:, -2:].to_numpy()
    got = hm.compute_lisi(X, meta, meta.columns, 30)
    assert np.allclose(got, stored)
    codes = np.stack([pd.Categorical(meta[c]).codes for c in meta.columns]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "lisi_ref_fixture.npz"), X=X.to_numpy(), codes=codes,
                        lisi_stored=stored, lisi_reference=got, perplexity=30)

    pcs = pd.read_csv(f"{REF}/data/pbmc_3500_pcs.tsv.gz", sep="\t").to_numpy()[:, :10]
    pm = pd.read_csv(f"{REF}/data/pbmc_3500_meta.tsv.gz", sep="\t")
    rng = np.random.default_rng(0)
    pm = pd.DataFrame({"donor": pm["donor"].to_numpy(), "noise": rng.integers(0, 5, len(pm)).astype(str)})
    for perp in (30, 15):
        got = hm.compute_lisi(pcs, pm, ["donor", "noise"], perp)
        codes = np.stack([pd.Categorical(pm[c]).codes for c in pm.columns]).astype(np.int32)
        np.savez_compressed(os.path.join(HERE, f"lisi_pbmc_p{perp}.npz"), X=pcs, codes=codes, lisi_reference=got,
                            perplexity=perp)


if __name__ == "__main__":
    main()
