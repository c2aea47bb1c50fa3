"""Large-N parity pinned to the REFERENCE ITSELF (tests/golden/large_*.npz, written by
tests/golden/make_ridge_conditioning.py from /root/reference in the build container; nothing here reads it).

Shapes: BASELINE configs[1] (69k x 50, 4 batches, K=30), configs[2]'s shape at 150k cells (K=100, 8 batches) and
configs[4]'s shape at 40k cells (200 PCs, K=200, 32 batches); 5 k-means rounds + 1 ridge correction from the stored Y0 on
the reference's own torch.randperm stream.  What pins what (numbers: tests/golden/ridge_conditioning.json):

  * R, O, E, the four objective histories -- against (i), the plain reference: it reproduces itself to 1e-6 there.
  * Z_corr -- against (iii), the reference's OWN moe_correct_ridge (harmony.py:535-569) run on float64 copies of its
    tensors: self-noise 3e-8..6e-8 between 1 and 8 threads.  The plain fp32 ridge is not a pin at 1e-4 on the 50-PC shapes:
    cov's condition number (1e3..7e3) amplifies the fp32 SUMMATION order of cov and of the right-hand sides, the reference
    moves by 1.0e-3 (69k) and 1.6e-3 (150k) between 1 and 8 threads and sits as far from (iii) -- taking only the inverse in
    float64, variant (ii), does not help (same 1e-3 spread).  At the configs[4] shape cond is ~ 200..500 and (i) sits
    3.6e-6 from (iii): there Z_corr is checked against (i) at 1e-4 as well.
  * against (i) on the 50-PC shapes Z_corr is held to 3x the reference's own 1-vs-8-thread spread (stored in the file).

CPU (`-m "not gpu"`): the oracle against the same goldens -- fp32 mode for R / objectives / C5-shape Z_corr, and
`ridge_dtype=float64` against (iii): the oracle variant the other large-N GPU tests compare with is thereby pinned too.
GPU: the engine through the C ABI."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = ["c2", "c3shape", "c5shape"]
KW = dict(max_iter_harmony=1, max_iter_kmeans=5, epsilon_cluster=0.0, epsilon_harmony=-1e30, random_state=0)


def load_large(name):
    from bench import synthetic_dataset
    g = np.load(os.path.join(GOLDEN, f"large_{name}.npz"))
    N, d, B, K, seed = (int(x) for x in g["shape"])
    Z, meta = synthetic_dataset(N, d, B, K, seed=seed)
    return Z, meta, K, g


def rows_err(Zrows, ref_rows, absmax):
    a, b = Zrows.astype(np.float64), ref_rows.astype(np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b)), float(np.abs(a - b).max() / absmax)


def check_pre_ridge(name, R_rows, R_colsum, O, E, obj, g):
    """Everything that enters the ridge step, against the plain reference."""
    relR = np.linalg.norm(R_rows.astype(np.float64) - g["R_rows"]) / np.linalg.norm(g["R_rows"].astype(np.float64))
    assert relR <= 1e-4, f"{name}: R rows relF={relR:.2e}"
    np.testing.assert_allclose(R_colsum, g["R_colsum"], rtol=3e-4, atol=3e-4)
    scale = float(np.abs(g["O"]).max())
    np.testing.assert_allclose(O, g["O"], rtol=3e-4, atol=3e-6 * scale)
    np.testing.assert_allclose(E, g["E"], rtol=3e-4, atol=3e-6 * scale)
    for key in ("objective_kmeans", "objective_kmeans_dist", "objective_kmeans_entropy"):
        np.testing.assert_allclose(obj[key], g[key], rtol=2e-5, err_msg=key)
    np.testing.assert_allclose(obj["objective_kmeans_cross"], g["objective_kmeans_cross"], rtol=1e-4)
    return relR


def check_z(name, Zrows, g, who):
    absmax = float(g["Zcorr_absmax"])
    rel3, max3 = rows_err(Zrows, g["Zcorr_rows_ridge64"], absmax)
    rel1, max1 = rows_err(Zrows, g["Zcorr_rows_plain"], absmax)
    noise = float(g["plain_relF_1_vs_8_threads"])
    print(f"{who} {name}: Z_corr vs (iii) reference ridge in float64 relF={rel3:.2e} max={max3:.2e}; vs (i) plain relF={rel1:.2e} "
          f"(reference 1 vs 8 threads: {noise:.1e})")
    return rel3, max3, rel1, max1, noise


# ------------------------------------------------------------------------------------------------
# CPU: the oracle (both ridge modes) against the reference's outputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_reference_large(name):
    from oracle import oracle_run_harmony
    Z, meta, K, g = load_large(name)
    rows = g["rows"]
    oo = oracle_run_harmony(Z, meta, ["batch"], nclust=K, Y0=g["Y0"], **KW)
    obj = {k: getattr(oo, k) for k in ("objective_kmeans", "objective_kmeans_dist", "objective_kmeans_entropy", "objective_kmeans_cross")}
    check_pre_ridge(name, oo.R.T[rows], oo.R.astype(np.float64).sum(axis=1), oo.O, oo.E, obj, g)
    rel3, max3, rel1, max1, noise = check_z(name, oo.result()[rows], g, "oracle fp32")
    # the fp32 oracle is one more realisation of the reference's fp32 arithmetic: within the reference's own spread of
    # (i), and -- where (i) is a pin (configs[4] shape) -- within 1e-4 of it
    assert rel1 <= max(1e-4, 3 * noise) and max1 <= max(1e-4, 3 * noise)
    # the float64-ridge variant the large-N engine tests use, against the reference's own float64 evaluation: the ridge
    # step starts from Z_orig and the (unchanged) R, so it is simply run again on the same state
    oo.ridge_dtype = np.dtype(np.float64)
    oo.moe_correct_ridge()
    rel3, max3, _, _, _ = check_z(name, oo.result()[rows], g, "oracle ridge_dtype=float64")
    assert rel3 <= 2e-6 and max3 <= 2e-6                     # measured 3e-8 .. 1.2e-7 (R entering the ridge differs by 1e-6)


# ------------------------------------------------------------------------------------------------
# GPU: the engine through the C ABI
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_vs_reference_large(name, monkeypatch):
    from harmonypy_amd import harmony as H
    monkeypatch.setenv("HMX_UPDATE_ORDER", "torch")            # the reference's own randperm stream (harmony.py:471)
    Z, meta, K, g = load_large(name)
    rows = g["rows"]
    ho = H.run_harmony(Z, meta, ["batch"], nclust=K, verbose=False, _y0=g["Y0"], **KW)
    assert ho.kmeans_rounds == [int(r) for r in g["kmeans_rounds"]]
    R = ho.R
    obj = {k: getattr(ho, k) for k in ("objective_kmeans", "objective_kmeans_dist", "objective_kmeans_entropy", "objective_kmeans_cross")}
    relR = check_pre_ridge(name, R[rows], R.astype(np.float64).sum(axis=0), ho.O, ho.E, obj, g)
    rel3, max3, rel1, max1, noise = check_z(name, ho.Z_corr[rows], g, f"engine (R relF={relR:.1e})")
    # measured 5e-8 .. 1.4e-7; the bar is the size of the R difference entering the ridge (1e-6), far below the 1e-4 of the
    # small goldens and below the reference's own 1-vs-8-thread spread of (i)
    assert rel3 <= 2e-6 and max3 <= 2e-6, f"{name}: Z_corr vs the reference's float64 ridge relF={rel3:.2e} max={max3:.2e}"
    assert rel1 <= max(1e-4, 3 * noise) and max1 <= max(1e-4, 3 * noise), f"{name}: Z_corr vs the plain reference relF={rel1:.2e}"


# ------------------------------------------------------------------------------------------------
# The benchmarked size on the benchmarked path: BASELINE configs[2], 1M cells, through hmx_cluster with the update order
# built on the device -- against a run of the REFERENCE ITSELF on the same blocks (tests/golden/make_c3_full.py replaces
# harmony.py:471's randperm by the engine's keyed bijection; 5 rounds + 1 ridge).
# ------------------------------------------------------------------------------------------------
def load_c3_full():
    from bench import synthetic_dataset
    g = np.load(os.path.join(GOLDEN, "large_c3full.npz"))
    N, d, B, K, seed, rounds = (int(x) for x in g["shape"])
    Z, meta = synthetic_dataset(N, d, B, K, seed=0, cell_seed=0)     # bench.py's configs[2] data set (rank 0)
    return Z, meta, K, seed, rounds, g


def test_c3_full_golden_is_self_consistent():
    """CPU: the fixture is what its recipe says (shape, sample, histories) and the reference's R rows are distributions;
    the oracle is NOT run at this size here (two minutes of NumPy) -- it is pinned by the thirteen smaller goldens."""
    g = np.load(os.path.join(GOLDEN, "large_c3full.npz"))
    N, d, B, K, seed, rounds = (int(x) for x in g["shape"])
    assert (N, d, B, K, seed, rounds) == (1_000_000, 50, 8, 100, 0, 5)
    assert np.array_equal(g["rows"], np.linspace(0, N - 1, 2000).astype(np.int64))
    assert g["R_rows"].shape == (2000, K) and g["Zcorr_rows_ridge64"].shape == (2000, d) and g["Y0"].shape == (d, K)
    np.testing.assert_allclose(g["R_rows"].sum(axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(g["R_colsum"].sum(), N, rtol=1e-6)
    np.testing.assert_allclose(g["O"].sum(), N, rtol=1e-5)
    assert len(g["objective_kmeans"]) == rounds + 1 and [int(r) for r in g["kmeans_rounds"]] == [rounds]
    assert float(g["R_rows_relF_between_the_two_runs"]) == 0.0      # (i) and (iii) differ in the ridge step only
    assert 1e-4 < float(g["plain_vs_ridge64_relF"]) < 2e-3          # the plain fp32 ridge is no pin at this size (5e-4)


@pytest.mark.gpu
def test_engine_vs_reference_c3_full_on_the_bench_path(monkeypatch):
    """What `python bench.py` times, at the size it is timed on: configs[2]'s 1M cells, device-built update order, the rounds
    of the iteration inside hmx_cluster (objective read-back deferred), next round's lists on the side stream.  R rows and
    column sums, O, E and the four objective histories against the plain reference; Z_corr against the reference's own
    ridge in float64 at 2e-6 (and no further from the plain reference than 3x its distance from that)."""
    from harmonypy_amd import harmony as H
    monkeypatch.setenv("HMX_UPDATE_ORDER", "device")
    Z, meta, K, seed, rounds, g = load_c3_full()
    rows = g["rows"]
    ho = H.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=0, verbose=False, random_state=seed, _y0=g["Y0"])
    assert ho.update_order == "device"
    ho.cluster(_rounds=rounds)                               # hmx_cluster: all rounds in one call
    cnt = ho._engine.counters()
    assert cnt["sweep_waits"] > 0 and cnt["sweep_fallbacks"] == 0, cnt     # the persistent sweep ran
    R = ho.R
    obj = {k: getattr(ho, k) for k in ("objective_kmeans", "objective_kmeans_dist", "objective_kmeans_entropy", "objective_kmeans_cross")}
    relR = check_pre_ridge("c3full", R[rows], R.astype(np.float64).sum(axis=0), ho.O, ho.E, obj, g)
    ho.moe_correct_ridge()
    Zr = ho.Z_corr[rows]
    absmax = float(g["Zcorr_absmax"])
    rel3, max3 = rows_err(Zr, g["Zcorr_rows_ridge64"], absmax)
    rel1, max1 = rows_err(Zr, g["Zcorr_rows_plain"], absmax)
    spread = float(g["plain_vs_ridge64_relF"])
    print(f"engine c3full (1M cells, bench path; R relF={relR:.1e}): Z_corr vs (iii) reference ridge in float64 relF={rel3:.2e} "
          f"max={max3:.2e}; vs (i) plain relF={rel1:.2e} (plain vs (iii): {spread:.1e})")
    assert rel3 <= 2e-6 and max3 <= 2e-6, f"Z_corr vs the reference's float64 ridge relF={rel3:.2e} max={max3:.2e}"
    assert rel1 <= 3 * spread
