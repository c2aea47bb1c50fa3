"""Cells sharded over ranks, CPU side (gloo, world_size 2): the host logic of harmonypy_amd.dist /
the sharded front end, and the math the sharded engine relies on -- summing only the small
cross-cell tables over ranks reproduces the unsharded run (oracle/sharded_oracle.py vs the
reference goldens).  No GPU, no engine calls."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_z_close, load_case


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(mode, case, outdir, world=2, opts=None, timeout=600):
    """Run tests/_shard_worker.py on `world` ranks; returns the per-rank result dicts."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py"), mode, case,
                                       str(outdir), json.dumps(opts or {})], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    if os.environ.get("HMX_DEBUG_BOX"):
        for r in range(world):
            print(f"--- rank {r} log ---\n{logs[r][-3000:]}")
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-4000:]}"
    return [dict(np.load(os.path.join(outdir, f"rank{r}.npz"), allow_pickle=False)) for r in range(world)]


@pytest.mark.parametrize("case", ["pbmc_short", "pbmc_two_vars", "synth_small_lambda_est"])
def test_sharded_oracle_reproduces_reference_golden(case, tmp_path):
    """world_size 2, gloo: every rank runs its slice, only the small tables are summed; the
    stitched Z_corr matches the REFERENCE's unsharded output within 1e-4."""
    data, meta, vars_use, kw, g = load_case(case)
    res = launch("oracle", case, tmp_path)
    assert [int(r["lo"]) for r in res] == [0, int(res[0]["hi"])] and int(res[1]["hi"]) == data.shape[0]
    assert int(res[0]["hi"]) != data.shape[0] // 2                       # uneven slices
    Z = np.concatenate([r["Z_corr"] for r in res], axis=0)
    rel_f, max_rel = assert_z_close(Z, g["Z_corr"])
    print(f"{case}: sharded oracle vs reference relF={rel_f:.2e} max={max_rel:.2e}")
    for r in res:
        # every rank sees the same history, and it is the reference's
        np.testing.assert_allclose(r["objective_kmeans"], g["objective_kmeans"], rtol=2e-5)
        np.testing.assert_allclose(r["objective_harmony"], g["objective_harmony"], rtol=2e-5)
        assert list(r["kmeans_rounds"]) == [int(x) for x in g["kmeans_rounds"]]
        assert int(r["n_collectives"]) > 0
    np.testing.assert_array_equal(res[0]["objective_kmeans"], res[1]["objective_kmeans"])


@pytest.mark.parametrize("world,case", [(4, "pbmc_short"), (4, "pbmc_two_vars"), (4, "pbmc_default"), (8, "pbmc_short")])
def test_sharded_oracle_world_4_and_8(world, case, tmp_path):
    """The same with 4 and 8 ranks (the C ABI allows 8: hmx_set_ranks): uneven contiguous slices of the donor-sorted
    pbmc cells, so most ranks hold a SINGLE donor (their other batch groups are empty) and no rank sees every level --
    batch levels, proportions and the cluster count must be the job's, and summing only the small tables over the ranks
    must still give the reference's unsharded Z_corr within 1e-4 and its objective histories."""
    data, meta, vars_use, kw, g = load_case(case)
    res = launch("oracle", case, tmp_path, world=world)
    los, his = [int(r["lo"]) for r in res], [int(r["hi"]) for r in res]
    assert los[0] == 0 and his[-1] == data.shape[0] and los[1:] == his[:-1]
    sizes = np.diff([0] + his)
    assert sizes.min() > 0 and sizes.max() - sizes.min() > 100                     # uneven slices
    donors_per_rank = [meta["donor"].iloc[lo:hi].nunique() for lo, hi in zip(los, his)]
    assert min(donors_per_rank) == 1 and meta["donor"].nunique() > 1               # ranks that hold one batch level only
    Z = np.concatenate([r["Z_corr"] for r in res], axis=0)
    rel_f, max_rel = assert_z_close(Z, g["Z_corr"])
    print(f"{case}, {world} ranks (donors per rank {donors_per_rank}): sharded oracle vs reference relF={rel_f:.2e} max={max_rel:.2e}")
    for r in res:
        np.testing.assert_allclose(r["objective_kmeans"], g["objective_kmeans"], rtol=2e-5)
        np.testing.assert_allclose(r["objective_harmony"], g["objective_harmony"], rtol=2e-5)
        assert list(r["kmeans_rounds"]) == [int(x) for x in g["kmeans_rounds"]]
        np.testing.assert_array_equal(r["objective_kmeans"], res[0]["objective_kmeans"])   # every rank walks the same history


def test_sharded_front_end_equals_unsharded(tmp_path):
    """Batch levels, batch proportions, cluster count and theta/tau scaling are those of the
    whole job on every rank (harmony.py:123-173 evaluated on the unsharded input)."""
    from harmonypy_amd.harmony import _prepare_inputs
    data, meta, vars_use, kw, g = load_case("pbmc_theta_tau")
    run_kw = {k: kw[k] for k in ("theta", "lamb", "sigma", "nclust", "tau") if k in kw}
    want = _prepare_inputs(data, meta, vars_use, **run_kw)
    res = launch("oracle", "pbmc_theta_tau", tmp_path, opts={"kw": {"max_iter_harmony": 1, "max_iter_kmeans": 1},
                                                             "forced": False})
    for r in res:
        assert int(r["K"]) == want["K"]
        np.testing.assert_array_equal(r["Pr_b"], want["Pr_b"])
        np.testing.assert_array_equal(r["theta"], want["theta"])


def test_block_lists_of_shards_partition_the_unsharded_lists():
    """harmony.py:471-484 with cells sharded: one permutation of all cells, every shard keeps its
    members of every block -- together exactly the unsharded blocks."""
    from harmonypy_amd.harmony import TILE, build_block_lists, build_layout, inverse_order
    rng = np.random.default_rng(5)
    N, B, nb = 1003, 3, 20
    codes = rng.integers(0, B, size=(N, 1)).astype(np.int32)
    combos = np.unique(codes, axis=0)
    order = rng.permutation(N)
    cpb = int(N * 0.05)
    _, order_all, gid_all, _, _ = build_layout(codes)
    rank_all = inverse_order(order_all)
    cells_all, tg_all, bs_all = build_block_lists(order, rank_all, gid_all, nb, cpb, B)
    inv_all = np.argsort(rank_all)                                    # internal -> original
    cut = 431
    members = [set() for _ in range(nb)]
    for lo, hi in ((0, cut), (cut, N)):
        _, order_loc, gid_loc, _, _ = build_layout(codes[lo:hi], combos)
        rank_loc = inverse_order(order_loc)
        cells, tg, bs = build_block_lists(order, rank_loc, gid_loc, nb, cpb, B, offset=lo)
        assert bs[0] == 0 and bs[-1] * TILE == cells.size
        for b in range(nb):
            blk = cells[bs[b] * TILE: bs[b + 1] * TILE]
            live = blk[blk >= 0]
            assert np.array_equal(np.repeat(tg[bs[b]:bs[b + 1]], TILE)[blk >= 0], gid_loc[live])
            members[b] |= set((order_loc[live] + lo).tolist())
    for b in range(nb):
        blk = cells_all[bs_all[b] * TILE: bs_all[b + 1] * TILE]
        assert members[b] == set(inv_all[blk[blk >= 0]].tolist()), b
    assert sum(len(m) for m in members) == N


def test_layout_with_job_wide_groups():
    from harmonypy_amd.harmony import build_layout
    codes = np.array([[0, 3], [1, 3], [1, 4], [0, 3]], dtype=np.int32)
    combos = np.array([[0, 3], [0, 4], [1, 3], [1, 4]], dtype=np.int32)   # the job has a group this shard lacks
    gc, order, gid, cells, tg = build_layout(codes, combos)
    assert gc.shape == (4, 2) and set(gid.tolist()) == {0, 2, 3}
    assert np.array_equal(np.sort(cells[cells >= 0]), np.arange(4))
    with pytest.raises(ValueError):
        build_layout(np.array([[2, 3]], dtype=np.int32), combos)
