"""Cells sharded over several engines, on the GPU (`-m gpu`).  A gpurun box has ONE MI355X, so
the two ranks of these tests are two processes (two engines) on the same device exchanging their
tables through the host transport (gloo) -- the partition logic, the exchange points and the
update-order construction are exactly those of a multi-GPU job; the RCCL transport itself is
exercised with a one-rank communicator."""
import numpy as np
import pytest

from conftest import assert_z_close, load_case, z_errors
from test_sharded_cpu import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("peer", ["1", "0"])
@pytest.mark.parametrize("case", ["pbmc_default", "pbmc_two_vars", "pbmc_lambda_est", "synth_small_default"])
def test_two_shards_match_reference_golden(case, peer, tmp_path, monkeypatch):
    """Reference's Y0, permutation stream and round schedule; cells split unevenly over two
    engines: the stitched Z_corr is the reference's within 1e-4 and both ranks hold the same
    O / E / Y and objective history.  peer=1: the block sums travel through the peer boxes inside
    the persistent sweep kernel; peer=0: one launch and one collective per update block."""
    monkeypatch.setenv("HMX_PEER_EXCHANGE", peer)
    data, meta, vars_use, kw, g = load_case(case)
    res = launch("engine", case, tmp_path, world=2, opts={"transport": "host", "order": "torch"})
    Z = np.concatenate([r["Z_corr"] for r in res], axis=0)
    rel_f, max_rel = assert_z_close(Z, g["Z_corr"])
    print(f"{case}: 2 shards vs reference relF={rel_f:.2e} max={max_rel:.2e}")
    for r in res:
        assert str(r["transport"]) == ("host+peer" if peer == "1" else "host")
        np.testing.assert_allclose(r["objective_harmony"], g["objective_harmony"], rtol=2e-5)
        np.testing.assert_allclose(r["objective_kmeans"], g["objective_kmeans"], rtol=2e-5)
        np.testing.assert_allclose(r["O"], g["O"], rtol=3e-4, atol=3e-4)
        np.testing.assert_allclose(r["E"], g["E"], rtol=3e-4, atol=3e-4)
    for key in ("O", "E", "Y", "objective_kmeans"):
        np.testing.assert_array_equal(res[0][key], res[1][key])
    np.testing.assert_allclose(res[0]["R_colsum_local"] + res[1]["R_colsum_local"], g["R_colsum"], rtol=3e-4, atol=3e-4)


@pytest.mark.parametrize("box", ["fine", "coarse"])
def test_peer_boxes_fine_grained_and_selftest_soak(box, tmp_path, monkeypatch):
    """The peer boxes are FINE-GRAINED device memory (hipExtMallocWithFlags: what HIP promises to keep coherent for another
    device's writes while a kernel polls them; HMX_PEER_BOX=coarse keeps the plain hipMalloc of earlier rounds), exported through
    IPC as before, and the self-test that gates the in-kernel exchange soaks: 3 000 exchange cycles instead of 8, every cycle
    with the stale-line step (the reader touches the lines the peer overwrites next BEFORE it lets the peer go on).  Then the
    job itself: two engines, block sums through the boxes, the reference's Z_corr within 1e-4."""
    monkeypatch.setenv("HMX_PEER_EXCHANGE", "1")
    monkeypatch.setenv("HMX_PEER_SELFTEST_ITERS", "3000")
    monkeypatch.setenv("HMX_PEER_BOX", box)
    case = "pbmc_short"
    data, meta, vars_use, kw, g = load_case(case)
    res = launch("engine", case, tmp_path, world=2, opts={"transport": "host", "order": "torch"})
    for r in res:
        assert str(r["transport"]) == "host+peer" and str(r["peer_box"]) == box and int(r["sweep_fallbacks"]) == 0
    Z = np.concatenate([r["Z_corr"] for r in res], axis=0)
    rel_f, max_rel = assert_z_close(Z, g["Z_corr"])
    print(f"peer boxes {box}-grained, self-test soak of 3000 cycles: 2 shards vs reference relF={rel_f:.2e} max={max_rel:.2e}")


def test_device_order_does_not_depend_on_sharding(tmp_path):
    """The device-side update order is a function of (seed, round, global cell id): one engine
    and two engines walk the same blocks, so their results agree to summation-order noise."""
    case = "pbmc_short"
    data, meta, vars_use, kw, g = load_case(case)
    d1, d2 = tmp_path / "w1", tmp_path / "w2"
    d1.mkdir(), d2.mkdir()
    one = launch("engine", case, d1, world=1, opts={"transport": "host", "order": "device"})
    two = launch("engine", case, d2, world=2, opts={"transport": "host", "order": "device"})
    Z1 = one[0]["Z_corr"]
    Z2 = np.concatenate([r["Z_corr"] for r in two], axis=0)
    rel_f, max_rel = z_errors(Z2, Z1)
    print(f"device order, 1 vs 2 shards: relF={rel_f:.2e} max={max_rel:.2e}")
    assert rel_f < 2e-5 and max_rel < 2e-5
    np.testing.assert_allclose(two[0]["objective_kmeans"], one[0]["objective_kmeans"], rtol=1e-5)


def test_rccl_transport_one_rank(tmp_path):
    """hmx_comm_unique_id / hmx_comm_init / ncclAllReduce at every exchange point, with a
    communicator of one rank (all a one-GPU box allows): same answer as the reference."""
    case = "pbmc_short"
    data, meta, vars_use, kw, g = load_case(case)
    res = launch("engine", case, tmp_path, world=1, opts={"transport": "rccl", "order": "torch"})
    assert str(res[0]["transport"]).startswith("rccl")
    assert_z_close(res[0]["Z_corr"], g["Z_corr"])
    np.testing.assert_allclose(res[0]["objective_kmeans"], g["objective_kmeans"], rtol=2e-5)


def test_sharded_device_kmeans_and_device_order(tmp_path, monkeypatch):
    """Everything a large sharded job uses at once: seeds on a gathered subsample + Lloyd iterations with
    job-wide sums, device-side update order, peer exchange inside the sweep kernel, natural round
    counts.  Not the reference's random stream, so: per-PC correlation with its output > 0.99, the
    same history on both ranks."""
    from scipy.stats import pearsonr
    case = "pbmc_default"
    data, meta, vars_use, kw, g = load_case(case)
    monkeypatch.setenv("HMX_KMEANS", "device")
    res = launch("engine", case, tmp_path, world=2, opts={"transport": "host", "order": "device", "Y0": False, "forced": False})
    Z = np.concatenate([r["Z_corr"] for r in res], axis=0)
    cors = [pearsonr(Z[:, j], g["Z_corr"][:, j])[0] for j in range(Z.shape[1])]
    assert min(cors) > 0.99, min(cors)
    np.testing.assert_array_equal(res[0]["objective_kmeans"], res[1]["objective_kmeans"])
    assert list(res[0]["kmeans_rounds"]) == list(res[1]["kmeans_rounds"])
    assert abs(res[0]["objective_harmony"][-1] / g["objective_harmony"][-1] - 1) < 2e-2


def test_two_shards_survive_a_sweep_timeout(tmp_path, monkeypatch):
    """Two engines, peer exchange inside the persistent kernel, and HMX_SPIN_LIMIT=0: every wait for the other rank's
    flags gives up.  Both ranks must notice together (the objective block carries the count of failed waits and is
    all-reduced as a whole), replay the round EXACTLY with one launch + one collective per block (from the round's own
    start: saved O, the removal sums and centroids already summed over the ranks) and end with the same tables and
    history on both ranks -- and with the reference's Z_corr within 1e-4."""
    monkeypatch.setenv("HMX_PEER_EXCHANGE", "1")
    monkeypatch.setenv("HMX_SPIN_LIMIT", "0")
    case = "pbmc_short"
    data, meta, vars_use, kw, g = load_case(case)
    res = launch("engine", case, tmp_path, world=2, opts={"transport": "host", "order": "torch"})
    for r in res:
        assert int(r["sweep_fallbacks"]) >= 1
        assert np.isfinite(r["Z_corr"]).all() and np.isfinite(r["objective_kmeans"]).all()
    for key in ("O", "E", "Y", "objective_kmeans"):
        np.testing.assert_array_equal(res[0][key], res[1][key])
    Z = np.concatenate([r["Z_corr"] for r in res], axis=0)
    rel_f, max_rel = assert_z_close(Z, g["Z_corr"])
    print(f"2 shards, every sweep timed out: relF={rel_f:.2e} max={max_rel:.2e}")
    np.testing.assert_allclose(res[0]["objective_kmeans"], g["objective_kmeans"], rtol=2e-5)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain command line starts two ranks itself (torch.distributed.run, one process
    per GPU; here gloo + two engines on the one GPU of the box) and reports n_gpus = 2, strong scaling of a configs[3]
    shaped job, with every rank's transport / peer-exchange / wait statistics in the line."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HMX_BENCH_BACKEND="gloo", HMX_ROUND_WGS="100", HMX_BENCH_CELLS="60000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--cpu-sample", "0", "--no-convergence"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["cells_total"] == 120000
    assert len(line["ranks"]["per_rank"]) == 2 and line["ranks"]["transports"] == ["host+peer"]
    assert line["ranks"]["peer_exchange_on_all"] and line["value"] > 0


def test_bench_four_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 4`: four ranks (gloo + four engines on the one GPU of the box, 50 compute workgroups + a
    gateway each so that all four persistent sweeps are resident together), BASELINE configs[3]'s shape split in four.  The
    in-kernel exchange of the per-block sums must be on for every rank (peer boxes through IPC, 4-way flags), nothing may time
    out, and every rank issues the same small number of collectives: per round one for the removal sums and centroid
    numerators and one for the objective block, plus one per ridge step -- not one per update block."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HMX_BENCH_BACKEND="gloo", HMX_ROUND_WGS="50", HMX_BENCH_CELLS="40000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    steps, warmup, rounds = 2, 1, 10
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", str(steps), "--warmup", str(warmup),
                        "--cpu-sample", "0", "--no-convergence"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["config"]["cells_total"] == 160000
    ranks = line["ranks"]
    assert len(ranks["per_rank"]) == 4 and ranks["transports"] == ["host+peer"] and ranks["peer_exchange_on_all"]
    assert all(pr["fallback_rounds"] == 0 and pr["sweep_waits"] > 0 for pr in ranks["per_rank"]), ranks["per_rank"]
    assert len(ranks["collectives_per_rank"]) == 1                          # every rank issued the same number
    per_step = (rounds * 2 + 1)
    total = ranks["collectives_per_rank"][0]
    print("4 ranks on one GPU:", round(line["value"] / 1e6, 1), "M cells/s/iteration; collectives per rank", total)
    # warm-up + timed steps (+ the one-off collectives of set-up and init_cluster): far below the 20 per round of the
    # one-collective-per-block path (rounds * 22 per step)
    assert (steps + warmup) * per_step <= total <= (steps + warmup) * per_step + 40, total


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 8`: the rank count of BASELINE configs[3] / [4] -- eight engines on the one GPU of the box over
    gloo, 28 compute workgroups + a gateway each (232 of 256 CUs: all eight persistent sweeps resident together).  Executes
    what a node of eight GPUs would index: eight peer boxes per rank, 8-way flags, eight gateway workgroups writing into every
    box.  The in-kernel exchange must be on for every rank, nothing may time out, and the collective count per rank is the
    small one (two per round + one per ridge step)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HMX_BENCH_BACKEND="gloo", HMX_ROUND_WGS="28", HMX_BENCH_CELLS="20000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    steps, warmup, rounds = 1, 1, 10
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(steps), "--warmup", str(warmup),
                        "--cpu-sample", "0", "--no-convergence"], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["cells_total"] == 160000
    ranks = line["ranks"]
    assert len(ranks["per_rank"]) == 8 and ranks["transports"] == ["host+peer"] and ranks["peer_exchange_on_all"]
    assert all(pr["fallback_rounds"] == 0 and pr["sweep_waits"] > 0 for pr in ranks["per_rank"]), ranks["per_rank"]
    assert len(ranks["collectives_per_rank"]) == 1
    per_step = (rounds * 2 + 1)
    total = ranks["collectives_per_rank"][0]
    print("8 ranks on one GPU:", round(line["value"] / 1e6, 1), "M cells/s/iteration; collectives per rank", total)
    assert (steps + warmup) * per_step <= total <= (steps + warmup) * per_step + 40, total
