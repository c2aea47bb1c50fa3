"""CPU tests of the host side: argument normalisation, batch groups, tile lists, the C ABI
surface.  No compute call is made (there is no GPU here); the product has no CPU fallback."""
import os
import re

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, load_case


def test_library_exports_every_declared_symbol():
    """libhmx.so loads and exports exactly what include/hmx.h declares."""
    from harmonypy_amd import _capi
    header = open(os.path.join(ROOT, "include", "hmx.h")).read()
    declared = sorted(set(re.findall(r"\b(hmx_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations found"
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in hmx.h but not exported"
    assert sorted(_capi.EXPORTS) == declared
    assert lib.hmx_abi_version() == _capi.HMX_ABI_VERSION == 8


def test_engine_fails_loudly_without_gpu():
    """No silent fallback: without a HIP device the constructor raises with the library's text."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from harmonypy_amd import _capi
    with pytest.raises(_capi.HmxError) as ei:
        _capi.Engine(100, 5, 3, 2, 2, 1, 20)
    assert "HIP" in str(ei.value) or "device" in str(ei.value)


def test_lisi_fails_loudly_without_gpu_and_checks_arguments():
    """compute_lisi has no CPU path either; argument errors are reported before any device work."""
    import pandas as pd
    import torch
    import harmonypy_amd as hm
    from harmonypy_amd import _capi
    X = np.random.default_rng(0).normal(size=(200, 4))
    meta = pd.DataFrame({"a": ["x", "y"] * 100})
    with pytest.raises(ValueError):
        hm.compute_lisi(X, meta.iloc[:10], ["a"], 30)                 # metadata rows != cells
    with pytest.raises(ValueError):
        hm.compute_lisi(X, meta, ["a"], 30, device="cpu")
    with pytest.raises(ValueError):
        hm.compute_lisi(X[:50], meta.iloc[:50], ["a"], 30)            # 90 neighbours of 50 points (sklearn's ValueError text)
    with pytest.raises(ValueError):
        hm.compute_lisi(X, meta, ["a"], 700)                          # 2100 > the 2040 neighbours the largest candidate list ranks exactly
    if not torch.cuda.is_available():
        with pytest.raises(_capi.HmxError) as ei:
            hm.compute_lisi(X, meta, ["a"], 30)
        assert "HIP" in str(ei.value) or "device" in str(ei.value)


def test_inverse_order_and_radix_layout():
    from harmonypy_amd.harmony import build_layout, inverse_order
    rng = np.random.default_rng(3)
    for G in (1, 7, 300):                                             # uint8 and uint16 radix keys
        codes = rng.integers(0, G, size=(5000, 1)).astype(np.int32)
        _, order, gid_int, cells, tile_grp = build_layout(codes)
        assert np.array_equal(np.sort(order), np.arange(5000))
        assert np.all(np.diff(gid_int) >= 0)                          # grouped
        same = gid_int[1:] == gid_int[:-1]
        assert np.all(np.diff(order)[same] > 0)                       # stable inside a group
        rank = inverse_order(order)
        assert np.array_equal(rank[order], np.arange(5000))
        assert np.array_equal(cells[cells >= 0], np.arange(5000))


def test_bad_arguments_are_rejected_by_the_abi():
    import ctypes as C
    from harmonypy_amd import _capi
    lib = _capi.load()
    h = C.c_void_p()
    cfg = _capi.HmxConfig(n_cells=0, n_pcs=5, n_clusters=3, n_batches=2, n_groups=2, n_vars=1, n_blocks=20)
    assert lib.hmx_create(C.byref(cfg), C.byref(h)) == -1
    assert b"positive" in lib.hmx_last_error()
    cfg = _capi.HmxConfig(n_cells=10, n_pcs=5, n_clusters=999, n_batches=2, n_groups=2, n_vars=1, n_blocks=20)
    assert lib.hmx_create(C.byref(cfg), C.byref(h)) == -1
    assert lib.hmx_create(None, C.byref(h)) == -1
    assert lib.hmx_moe_correct_ridge(None) == -1
    assert lib.hmx_sync(None) == -1


def test_device_argument():
    from harmonypy_amd.harmony import _device_index
    assert _device_index(None) == 0
    assert _device_index("cuda") == 0
    assert _device_index("cuda:3") == 3
    assert _device_index("hip:1") == 1
    for bad in ("cpu", "mps", "xla"):
        with pytest.raises(ValueError):
            _device_index(bad)


CASES = [
    dict(),
    dict(theta=1.0, tau=5, sigma=0.2, nclust=20),
    dict(lamb=-1),
    dict(theta=[2.0, 1.0], lamb=[1.0, 0.5]),
    dict(theta=[2.0, 1.0, 0.5, 3.0, 1.0, 1.0, 2.0], lamb=[1, 2, 3, 4, 5, 6, 7]),
    dict(lamb=0.5, theta=3),
]


@pytest.mark.parametrize("kw", CASES)
def test_prepare_inputs_matches_oracle_front_end(kw):
    """harmony.py:116-173: same Z orientation, K, sigma, theta, lamb, Pr_b; codes == one-hot."""
    from harmonypy_amd.harmony import _prepare_inputs
    from oracle import prepare_inputs
    data, meta, _, _, _ = load_case("pbmc_two_vars")
    two = any(isinstance(v, list) and len(v) in (2, 7) for v in kw.values())
    vars_use = ["donor", "tech"] if two else "donor"
    a = _prepare_inputs(data, meta, vars_use, **kw)
    b = prepare_inputs(data, meta, vars_use, **kw)
    assert a["K"] == b["K"] and a["lambda_estimation"] == b["lambda_estimation"]
    np.testing.assert_array_equal(a["Z"], b["Z"])
    np.testing.assert_array_equal(a["codes"].dense(), b["phi"])
    for key in ("sigma", "theta", "lamb", "Pr_b"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
        assert a[key].dtype == np.float32
    # either orientation of data_mat is accepted (harmony.py:117-118)
    a2 = _prepare_inputs(data.T, meta, vars_use, **kw)
    np.testing.assert_array_equal(a2["Z"], a["Z"])
    a3 = _prepare_inputs(pd.DataFrame(data), meta, vars_use, **kw)
    np.testing.assert_array_equal(a3["Z"], a["Z"])


def test_prepare_inputs_errors():
    from harmonypy_amd.harmony import _prepare_inputs
    data, meta, _, _, _ = load_case("pbmc_short")
    with pytest.raises(AssertionError):
        _prepare_inputs(data[:100], meta, "donor")
    with pytest.raises(AssertionError):
        _prepare_inputs(data, meta, "donor", theta=[1.0, 2.0])


def test_batch_codes_dense_roundtrip():
    from harmonypy_amd.harmony import BatchCodes
    rng = np.random.default_rng(0)
    codes = np.stack([rng.integers(0, 3, 50), 3 + rng.integers(0, 4, 50)], axis=1)
    bc = BatchCodes(codes, 7)
    back = BatchCodes.from_dense(bc.dense())
    np.testing.assert_array_equal(back.codes, codes)
    bad = bc.dense()
    bad[:, 0] = 0
    with pytest.raises(ValueError):
        BatchCodes.from_dense(bad)


def test_layout_groups_and_static_tiles():
    from harmonypy_amd.harmony import TILE, build_layout, inverse_order
    rng = np.random.default_rng(1)
    codes = np.stack([rng.integers(0, 3, 1000), 3 + rng.integers(0, 2, 1000)], axis=1).astype(np.int32)
    group_cols, order, gid_int, cells, tile_grp = build_layout(codes)
    rank = inverse_order(order)
    G = group_cols.shape[0]
    assert G == len(np.unique(codes, axis=0))
    np.testing.assert_array_equal(rank[order], np.arange(1000))
    np.testing.assert_array_equal(codes[order], group_cols[gid_int])       # internal cell -> its columns
    assert np.all(np.diff(gid_int) >= 0)                                   # group-sorted
    assert cells.size == TILE * tile_grp.size
    live = cells[cells >= 0]
    np.testing.assert_array_equal(live, np.arange(1000))                   # identity list
    for t, g in enumerate(tile_grp):                                       # one group per tile
        c = cells[t * TILE:(t + 1) * TILE]
        assert np.all(gid_int[c[c >= 0]] == g)


@pytest.mark.parametrize("N,block_size", [(3500, 0.05), (1237, 0.07), (37, 0.05), (16, 0.3), (1000, 0.13)])
def test_block_lists_reproduce_reference_blocks(N, block_size):
    """Blocks of harmony.py:474-484 as sets; (block, group) runs padded to tiles."""
    from harmonypy_amd.harmony import TILE, build_block_lists, build_layout, inverse_order
    rng = np.random.default_rng(N)
    codes = rng.integers(0, 4, N).astype(np.int32)[:, None]
    group_cols, order, gid_int, _, _ = build_layout(codes)
    rank = inverse_order(order)
    G = group_cols.shape[0]
    upd = rng.permutation(N)
    nb = int(np.ceil(1.0 / block_size))
    cpb = int(N * block_size)
    cells, tile_grp, blk_start = build_block_lists(upd, rank, gid_int, nb, cpb, G)
    assert blk_start[0] == 0 and blk_start[-1] == tile_grp.size and cells.size == TILE * tile_grp.size
    seen = []
    for b in range(nb):
        lo = b * cpb
        hi = N if b == nb - 1 else (b + 1) * cpb
        ref_members = set(upd[lo:hi].tolist())                              # original cell ids
        mine = cells[blk_start[b] * TILE: blk_start[b + 1] * TILE]
        mine = mine[mine >= 0]
        assert set(order[mine].tolist()) == ref_members
        seen.append(mine)
        for t in range(blk_start[b], blk_start[b + 1]):
            c = cells[t * TILE:(t + 1) * TILE]
            assert np.all(gid_int[c[c >= 0]] == tile_grp[t]) and (c >= 0).any()
    allc = np.concatenate(seen)
    assert allc.size == N and len(set(allc.tolist())) == N                  # every cell exactly once


def test_check_convergence_semantics():
    """harmony.py:515-533 on hand-made histories."""
    from harmonypy_amd.harmony import Harmony
    ho = object.__new__(Harmony)
    ho.window_size, ho.epsilon_kmeans, ho.epsilon_harmony = 3, 1e-5, 1e-4
    ho.objective_kmeans = [10.0, 9.0, 8.0, 7.0]
    assert ho.check_convergence(0) is False                                 # needs > window+1 entries
    ho.objective_kmeans = [10.0, 9.0, 9.0, 9.0, 9.0]
    assert ho.check_convergence(0) == (abs(27.0 - 27.0) / 27.0 < 1e-5)
    ho.objective_kmeans = [10.0, 9.0, 8.0, 7.0, 6.0]
    assert not ho.check_convergence(0)
    ho.objective_harmony = [5.0]
    assert ho.check_convergence(1) is False
    ho.objective_harmony = [5.0, 5.0 - 1e-6]
    assert ho.check_convergence(1)
    ho.objective_harmony = [5.0, 6.0]                                       # signed test: an increase converges
    assert ho.check_convergence(1)
    ho.objective_harmony = [5.0, 4.0]
    assert not ho.check_convergence(1)
    assert ho.check_convergence(2) is True


def test_kmeans_mode_selection(monkeypatch):
    """Initial centroids: the reference's host fit for small jobs, seeds + GPU Lloyd for large ones."""
    from harmonypy_amd import harmony as H

    class Fake:
        _kmeans_mode = H.Harmony._kmeans_mode
        _wide_shape = H.Harmony._wide_shape
        K, d = 100, 50
    f = Fake()
    monkeypatch.delenv("HMX_KMEANS", raising=False)
    f.N_global = H.KMEANS_DEVICE_CELLS
    assert f._kmeans_mode() == "host"
    f.N_global = H.KMEANS_DEVICE_CELLS + 1
    assert f._kmeans_mode() == "device"
    assert not f._wide_shape()
    f.K = 200
    assert f._kmeans_mode() == "device" and f._wide_shape()   # device seeds, Lloyd on the subsample (no host fit of all cells)
    f.K = 100
    monkeypatch.setenv("HMX_KMEANS", "host")
    assert f._kmeans_mode() == "host"
    monkeypatch.setenv("HMX_KMEANS", "bogus")
    with pytest.raises(ValueError):
        f._kmeans_mode()


@pytest.mark.parametrize("n", [1, 2, 17, 1000, 4097, 100003])
def test_device_order_restatement_is_a_permutation(n):
    """oracle/device_order.py (the checker of the GPU's update-order lists): positions are a bijection of
    [0, n) for every round key, rounds differ, and the lists hold every cell once with one group per tile."""
    from oracle.device_order import block_lists, positions
    ids = np.arange(n)
    p0 = positions(ids, n, 7, 0)
    assert np.array_equal(np.sort(p0), ids)
    if n > 16:
        assert not np.array_equal(p0, positions(ids, n, 7, 1))
        assert not np.array_equal(p0, positions(ids, n, 8, 0))
    grp = (ids * 3 // max(n, 1)) % 3
    nb, cpb = 20, int(n * 0.05)
    cells, tg, bs = block_lists(ids, grp, 3, n, 7, 0, cpb, nb)
    live = cells[cells >= 0]
    assert np.array_equal(np.sort(live), ids) and bs[0] == 0 and bs[-1] * 16 == cells.size and tg.size == bs[-1]
    assert np.array_equal(np.repeat(tg, 16)[cells >= 0], grp[live])
    for b in range(nb):
        blk = cells[bs[b] * 16: bs[b + 1] * 16]
        want = (n - cpb * (nb - 1)) if b == nb - 1 else cpb
        assert (blk >= 0).sum() == want


def test_peer_attach_retries_with_coarse_boxes(monkeypatch):
    """harmonypy_amd.dist.Shard._attach_peers: fine-grained peer boxes first; if any rank cannot export / map / pass the self-test
    with them, all ranks together try once more with coarse-grained boxes, then give the in-kernel exchange up.  Driven here
    with a fake engine and a one-rank "world" (no GPU, no process group): the order of the calls and of HMX_PEER_BOX is the logic."""
    import os
    from harmonypy_amd import _capi
    from harmonypy_amd.dist import Shard

    class FakeEngine:
        def __init__(self, fail):
            self.fail, self.calls, self.kinds = fail, [], []

        def set_ranks(self, world, rank):
            self.calls.append("set_ranks")

        def peer_export(self):
            kind = os.environ.get("HMX_PEER_BOX")
            self.kinds.append(kind)
            self.calls.append(f"export:{kind}")
            if ("export", kind) in self.fail:
                raise _capi.HmxError("export failed", -2)
            return b"\0" * _capi.HMX_PEER_HANDLE_BYTES

        def peer_attach(self, handles):
            self.calls.append(f"attach:{self.kinds[-1]}")
            if ("attach", self.kinds[-1]) in self.fail:
                raise _capi.HmxError("hipIpcOpenMemHandle failed", -4)

        def peer_selftest(self):
            self.calls.append(f"selftest:{self.kinds[-1]}")
            return ("selftest", self.kinds[-1]) not in self.fail

        def peer_enable(self, on):
            self.calls.append(f"enable:{self.kinds[-1]}")

    def shard():
        s = Shard.__new__(Shard)
        s.rank, s.world = 0, 1
        s.allgather_object = lambda obj: [obj]
        s.allreduce_ = lambda arr: None
        return s

    monkeypatch.delenv("HMX_PEER_BOX", raising=False)
    monkeypatch.delenv("HMX_PEER_EXCHANGE", raising=False)
    e = FakeEngine(fail=set())
    assert shard()._attach_peers(e) is True
    assert e.calls == ["set_ranks", "export:fine", "attach:fine", "selftest:fine", "enable:fine"] and "HMX_PEER_BOX" not in os.environ
    e = FakeEngine(fail={("attach", "fine")})                      # e.g. IPC does not open fine-grained memory across devices
    assert shard()._attach_peers(e) is True
    assert e.calls == ["set_ranks", "export:fine", "attach:fine", "export:coarse", "attach:coarse", "selftest:coarse", "enable:coarse"]
    e = FakeEngine(fail={("selftest", "fine"), ("selftest", "coarse")})
    assert shard()._attach_peers(e) is False
    assert e.calls[-1] == "selftest:coarse" and not any(c.startswith("enable") for c in e.calls)
    monkeypatch.setenv("HMX_PEER_BOX", "coarse")                   # pinned: one attempt only, the variable stays
    e = FakeEngine(fail={("export", "coarse")})
    assert shard()._attach_peers(e) is False
    assert e.kinds == ["coarse"] and os.environ["HMX_PEER_BOX"] == "coarse"
    monkeypatch.setenv("HMX_PEER_EXCHANGE", "0")
    e = FakeEngine(fail=set())
    assert shard()._attach_peers(e) is False and e.calls == []
