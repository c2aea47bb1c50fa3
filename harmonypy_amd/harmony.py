"""Host side of the MI355X Harmony engine: the ``run_harmony`` / ``Harmony`` API.

Mirrors the public interface of harmonypy v0.2.0 (``harmonypy/harmony.py:49-215`` for
``run_harmony``, ``:218-569`` for ``Harmony``): same argument names and meaning, same
attributes, history lists and NumPy-returning properties.  All arithmetic of the
``harmonize()`` loop runs in ``libhmx.so`` (HIP kernels for gfx950) through ``_capi``;
Python keeps what the reference keeps on the host: argument normalisation, the sklearn
k-means++ initialisation call (``:369-373``), the convergence tests on the history lists
(``:515-533``) and the random update order (``:471``), which is drawn from the same
generator as the reference's ``device='cpu'`` run (``torch.manual_seed`` / ``torch.randperm``)
so that both walk identical blocks.

There is no CPU fallback: without ``libhmx.so`` and a visible MI355X this module raises.
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np
import pandas as pd

from . import _capi

logger = logging.getLogger("harmonypy_amd")
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)

# Source of the per-round update order (harmony.py:471):
#   "torch"  -- torch.randperm on the CPU generator seeded by run_harmony, i.e. the very stream
#               of the reference's device='cpu' run; the order is regrouped on the host.
#   "device" -- keyed bijection evaluated on the GPU (hmx_cluster_round_seeded); no host work,
#               a different (statistically equivalent) stream.
#   "auto"   -- "torch" up to AUTO_DEVICE_ORDER_CELLS cells, "device" above: the host path
#               costs ~0.2 s per round per million cells and would dominate the loop.
# Override with the environment variable HMX_UPDATE_ORDER.
UPDATE_ORDER = "auto"
AUTO_DEVICE_ORDER_CELLS = 200_000
# sharded jobs: rank 0 fits the initial k-means on all cells up to this many, else on a subsample
KMEANS_GATHER_CELLS = 2_000_000
# Initial centroids (harmony.py:369-373).  "host": the reference's sklearn KMeans call on all cells
# (bit-identical Y0; 18 s at 1M cells).  "device": k-means++ seeding on a subsample of
# KMEANS_SEED_CELLS cells (~300 per cluster at K=100), then the 25 Lloyd iterations on the GPU over all cells (all ranks when
# sharded).
# "auto": host up to KMEANS_DEVICE_CELLS cells, device above.  Override: HMX_KMEANS.
KMEANS = "auto"
KMEANS_DEVICE_CELLS = 200_000
KMEANS_SEED_CELLS = 32_768
# Who seeds the device k-means: "device" = the engine's k-means++ (hmx_kmeans_seed, counter-based
# draws), "sklearn" = sklearn.cluster.kmeans_plusplus on the host (NumPy's generator; ~0.15-0.5 s
# for 32k cells).  Override: HMX_KMEANS_SEEDS.
KMEANS_SEEDS = "device"

TILE = _capi.HMX_TILE


class BatchCodes:
    """Integer form of the batch design: ``codes[i, v]`` is the Phi row that cell ``i``
    has set for variable ``v`` (what ``pd.get_dummies`` would one-hot, harmony.py:133)."""

    def __init__(self, codes, n_batches):
        codes = np.asarray(codes)
        if codes.ndim == 1:
            codes = codes[:, None]
        self.codes = np.ascontiguousarray(codes, dtype=np.int32)
        self.n_batches = int(n_batches)

    @classmethod
    def from_dense(cls, Phi):
        """From the reference's dense B x N indicator matrix."""
        Phi = np.asarray(Phi)
        B, N = Phi.shape
        per_cell = (Phi > 0).sum(axis=0)
        V = int(per_cell[0]) if N else 1
        if not np.all(per_cell == V) or V < 1:
            raise ValueError("Phi must hold the same number of active batch rows for every cell")
        rows, cols = np.nonzero(Phi.T > 0)          # row-major => sorted by cell, then Phi row
        return cls(cols.reshape(N, V), B)

    def dense(self):
        N, V = self.codes.shape
        Phi = np.zeros((self.n_batches, N), dtype=np.float32)
        for v in range(V):
            Phi[self.codes[:, v], np.arange(N)] = 1.0
        return Phi


def _device_index(device):
    """Reference accepts 'cpu' | 'cuda' | 'mps' | None (harmony.py:35-46).  This engine is
    MI355X only: None / 'cuda' / 'hip' [':n'] select a HIP device; anything else is an error."""
    if device is None:
        return 0
    s = str(device)
    base, _, idx = s.partition(":")
    if base not in ("cuda", "hip", "gpu"):
        raise ValueError(f"harmonypy_amd runs on AMD MI355X (HIP) devices only; device={device!r} is not available")
    return int(idx) if idx else 0


def run_harmony(
    data_mat: np.ndarray,
    meta_data: pd.DataFrame,
    vars_use,
    theta=None,
    lamb=None,
    sigma=0.1,
    nclust=None,
    tau=0,
    block_size=0.05,
    max_iter_harmony=10,
    max_iter_kmeans=20,
    epsilon_cluster=1e-5,
    epsilon_harmony=1e-4,
    alpha=0.2,
    verbose=True,
    random_state=0,
    device=None,
    *,
    shard=None,
    _y0=None,
    _schedule=None,
):
    """Run Harmony batch-effect correction on an MI355X.

    Same arguments, defaults and return type as ``harmonypy.run_harmony``
    (harmony.py:49-115): ``data_mat`` cells x PCs or PCs x cells, ``meta_data`` cells x
    variables, ``vars_use`` the batch column(s); ``lamb=-1`` estimates lambda per cluster.
    Returns a finished ``Harmony`` object (``.Z_corr`` is cells x PCs).

    ``shard`` (keyword only, not in the reference): a ``harmonypy_amd.Shard``.  Every rank of
    the process group then passes ITS slice of the cells (rank r holds the r-th contiguous
    slice) and gets back an object over those cells; batch proportions, cluster count, blocks,
    centroids and corrections are those of the whole job (``dist.py``).

    ``_y0`` / ``_schedule`` (keyword only, not in the reference): replay aids of the parity tests
    and of ``bench.py`` -- ``_y0`` is a d x K matrix of initial centroids used instead of the
    k-means initialisation (harmony.py:369-373), ``_schedule`` a list of k-means round counts,
    one per Harmony iteration, replayed instead of the objective thresholds of harmony.py:455-458
    (which the reference decides on a few fp32 ulps, NOTES.md section 5).
    """
    _validate_arguments(nclust, block_size, meta_data, vars_use)
    p = _prepare_inputs(data_mat, meta_data, vars_use, theta, lamb, sigma, nclust, tau, shard=shard)
    dev = _device_index(device)
    if verbose:
        logger.info(f"Running Harmony (HIP engine on MI355X device {dev})")
        logger.info("  Parameters:")
        logger.info(f"    max_iter_harmony: {max_iter_harmony}")
        logger.info(f"    max_iter_kmeans: {max_iter_kmeans}")
        logger.info(f"    epsilon_cluster: {epsilon_cluster}")
        logger.info(f"    epsilon_harmony: {epsilon_harmony}")
        logger.info(f"    nclust: {p['K']}")
        logger.info(f"    block_size: {block_size}")
        logger.info(f"    lamb: dynamic (alpha={alpha})" if p["lambda_estimation"] else f"    lamb: {p['lamb'][1:]}")
        logger.info(f"    theta: {p['theta']}")
        logger.info(f"    sigma: {p['sigma'][:5]}..." if len(p["sigma"]) > 5 else f"    sigma: {p['sigma']}")
        logger.info(f"    random_state: {random_state}")
        logger.info(f"  Data: {p['Z'].shape[0]} PCs x {p['Z'].shape[1]} cells")
        logger.info(f"  Batch variables: {vars_use}")

    # Seeds exactly as the reference sets them (harmony.py:199-200); the torch CPU
    # generator is the source of the update order of every round (harmony.py:471).
    import torch
    np.random.seed(random_state)
    torch.manual_seed(random_state)

    return Harmony(
        p["Z"], p["codes"], p["Pr_b"], p["sigma"],
        p["theta"], p["lamb"], alpha, p["lambda_estimation"],
        max_iter_harmony, max_iter_kmeans,
        epsilon_cluster, epsilon_harmony, p["K"], block_size, verbose,
        random_state, device, shard=shard, _y0=_y0, _schedule=_schedule,
    )


def _validate_arguments(nclust, block_size, meta_data, vars_use):
    """Limits of this build, reported as ValueError before any engine exists (the reference has none:
    harmony.py:123-124 caps only the default cluster count)."""
    if nclust is not None and not (1 <= int(nclust) <= _capi.HMX_MAX_CLUSTERS):
        raise ValueError(f"nclust={nclust}: this build supports 1..{_capi.HMX_MAX_CLUSTERS} clusters")
    if not (0 < float(block_size) <= 1) or int(np.ceil(1.0 / float(block_size))) > _capi.HMX_MAX_BLOCKS:
        raise ValueError(f"block_size={block_size}: must lie in [1/{_capi.HMX_MAX_BLOCKS}, 1]")
    n_vars = 1 if isinstance(vars_use, str) else len(vars_use)
    if not (1 <= n_vars <= _capi.HMX_MAX_VARS):
        raise ValueError(f"{n_vars} batch variables: this build supports 1..{_capi.HMX_MAX_VARS}")


def _prepare_inputs(data_mat, meta_data, vars_use, theta=None, lamb=None, sigma=0.1, nclust=None, tau=0, shard=None):
    """Argument normalisation of ``run_harmony`` (harmony.py:116-173, 203-205) with the batch
    design kept as integer codes instead of a dense one-hot matrix.  With ``shard`` the inputs
    are this rank's slice; cell counts, batch levels and batch sizes are those of the whole job."""
    N = meta_data.shape[0]
    if hasattr(data_mat, "values"):
        data_mat = data_mat.values
    data_mat = np.asarray(data_mat)
    if data_mat.shape[1] != N:                                  # harmony.py:117-118
        data_mat = data_mat.T
    assert data_mat.shape[1] == N, \
        "data_mat and meta_data do not have the same number of cells"
    if data_mat.shape[0] > _capi.HMX_MAX_PCS:
        raise ValueError(f"{data_mat.shape[0]} PCs: this build supports up to {_capi.HMX_MAX_PCS}")

    N_all = N
    if shard is not None:
        _, N_all = shard.layout(N)
    if nclust is None:                                          # harmony.py:123-124
        nclust = int(min(round(N_all / 30.0), 100))
    if isinstance(sigma, (float, int)) and not isinstance(sigma, bool):   # harmony.py:126-127
        sigma = np.repeat(float(sigma), nclust)
    sigma = np.asarray(sigma, dtype=np.float32)
    if isinstance(vars_use, str):
        vars_use = [vars_use]

    # Batch design as integer codes in pd.get_dummies' column order (harmony.py:133):
    # variables in vars_use order, levels sorted within a variable.
    codes = np.empty((N, len(vars_use)), dtype=np.int32)
    phi_n = []
    offset = 0
    for v, name in enumerate(vars_use):
        # levels that are declared but hold no cell (a subsetted categorical column) get no Phi row:
        # harmony.py:134 counts the levels present (describe()["unique"]), and an empty batch would
        # leave a group without cells
        full = pd.Categorical(meta_data[name])
        cat = full.remove_unused_categories()
        if shard is not None:
            # the levels of the whole job: identical declared categories keep their declared order
            # (minus the levels no rank uses), otherwise the sorted union of the levels in use
            used = np.bincount(full.codes[full.codes >= 0], minlength=len(full.categories))
            parts = shard.allgather_object((list(full.categories), used))
            if all(lv == parts[0][0] for lv, _ in parts):
                total = np.sum([u for _, u in parts], axis=0)
                levels = [lv for lv, n in zip(parts[0][0], total) if n > 0]
            else:
                levels = sorted(set().union(*[[lv for lv, n in zip(lvs, u) if n > 0] for lvs, u in parts]))
            cat = pd.Categorical(meta_data[name], categories=levels)
        if (cat.codes < 0).any():
            raise ValueError(f"meta_data[{name!r}] has missing values")
        codes[:, v] = cat.codes.astype(np.int32) + offset
        phi_n.append(len(cat.categories))
        offset += len(cat.categories)
    phi_n = np.asarray(phi_n, dtype=int)                        # harmony.py:134
    B = int(offset)

    if theta is None:                                           # harmony.py:137-147
        theta = np.repeat([2] * len(phi_n), phi_n).astype(np.float32)
    elif isinstance(theta, (float, int)):
        theta = np.repeat([theta] * len(phi_n), phi_n).astype(np.float32)
    elif len(theta) == len(phi_n):
        theta = np.repeat([theta], phi_n).astype(np.float32)
    else:
        theta = np.asarray(theta, dtype=np.float32)
    assert len(theta) == np.sum(phi_n), "each batch variable must have a theta"

    lambda_estimation = False                                   # harmony.py:150-166
    if lamb is None:
        lamb = np.insert(np.repeat([1] * len(phi_n), phi_n).astype(np.float32), 0, 0).astype(np.float32)
    elif np.isscalar(lamb) and lamb == -1:
        lambda_estimation = True
        lamb = np.zeros(1, dtype=np.float32)
    elif isinstance(lamb, (float, int)):
        lamb = np.insert(np.repeat([lamb] * len(phi_n), phi_n).astype(np.float32), 0, 0).astype(np.float32)
    elif len(lamb) == len(phi_n):
        lamb = np.insert(np.repeat([lamb], phi_n).astype(np.float32), 0, 0).astype(np.float32)
    else:
        lamb = np.asarray(lamb, dtype=np.float32)
        if len(lamb) == np.sum(phi_n):
            lamb = np.insert(lamb, 0, 0).astype(np.float32)

    counts = np.bincount(codes.ravel(), minlength=B).astype(np.int64)
    if shard is not None:
        shard.allreduce_(counts)
    N_b = counts.astype(np.float32)                                    # harmony.py:169 (phi.sum(axis=1))
    Pr_b = (N_b / N_all).astype(np.float32)                            # harmony.py:170
    if tau > 0:                                                        # harmony.py:172-173
        theta = theta * (1 - np.exp(-(N_b / (nclust * tau)) ** 2))

    return dict(Z=np.asarray(data_mat, dtype=np.float32), codes=BatchCodes(codes, B), Pr_b=Pr_b,
                sigma=sigma.astype(np.float32), theta=np.asarray(theta, dtype=np.float32), lamb=lamb,
                lambda_estimation=lambda_estimation, K=int(nclust), vars_use=list(vars_use))


def build_layout(codes, combos=None):
    """Group cells by their multi-hot batch pattern.

    Returns (group_cols G x V, order internal->original,
    gid_int group of every internal cell, static_cells, static_tile_grp): the static list is
    the identity over the group-sorted cells with every group padded to whole tiles.
    ``combos`` (sorted unique rows) fixes the groups of the whole job for a shard that may not
    hold cells of every group.
    """
    N = codes.shape[0]
    counts = None
    if combos is None and codes.shape[1] == 1:       # one batch variable: groups are its levels (fast path)
        col = codes[:, 0]
        per_code = np.bincount(col)                 # codes are small non-negative integers
        levels = np.flatnonzero(per_code)
        counts = per_code[levels]
        lut = np.full(int(levels[-1]) + 1 if len(levels) else 1, -1, dtype=np.int32)
        lut[levels] = np.arange(len(levels), dtype=np.int32)
        gid = lut[col]
        combos = levels.reshape(-1, 1)
    elif combos is None:
        combos, gid = np.unique(codes, axis=0, return_inverse=True)
    else:
        combos = np.asarray(combos, dtype=codes.dtype)
        both, inv = np.unique(np.vstack([combos, codes]), axis=0, return_inverse=True)
        if both.shape[0] != combos.shape[0]:
            raise ValueError("a cell's batch pattern is missing from the job-wide group table")
        inv = inv.reshape(-1)
        gid = inv[combos.shape[0]:]
    gid = gid.reshape(-1).astype(np.int32)
    G = combos.shape[0]
    # NumPy's stable sort of 8/16-bit keys is a radix sort: one O(N) pass per key byte
    key = gid.astype(np.uint8) if G <= 256 else gid.astype(np.uint16) if G <= 65536 else gid
    order = np.asarray(np.argsort(key, kind="stable"), dtype=np.int64)
    if counts is None:
        counts = np.bincount(gid, minlength=G)
    cells, tile_grp = [], []
    start = 0
    for g, c in enumerate(counts):
        nt = -(-int(c) // TILE)
        seg = np.full(nt * TILE, -1, dtype=np.int32)
        seg[:c] = np.arange(start, start + c, dtype=np.int32)
        cells.append(seg)
        tile_grp.append(np.full(nt, g, dtype=np.int32))
        start += int(c)
    return (np.ascontiguousarray(combos, dtype=np.int32), order, gid[order],
            np.concatenate(cells), np.concatenate(tile_grp))


def inverse_order(order):
    """rank: original cell -> internal (group-sorted) position."""
    rank = np.empty(order.shape[0], dtype=np.int32)
    rank[order] = np.arange(order.shape[0], dtype=np.int32)
    return rank


def build_block_lists(update_order, rank, gid_int, n_blocks, cells_per_block, G, offset=0):
    """Turn the reference's update order (harmony.py:471) into the engine's tile lists.

    Block b holds positions [b*cpb, (b+1)*cpb) of the order, the last block the remainder
    (harmony.py:482-484).  Only block membership matters to the update, so inside a block
    cells are regrouped by batch group and every (block, group) run is padded with -1 to a
    multiple of 16.  Returns (cells, tile_group, block_tile_start).

    ``update_order`` is a permutation of ALL cells of the job; a shard (``len(rank)`` cells,
    global ids ``offset ..``) keeps its own cells of every block.
    """
    Ng, nb, cpb = len(update_order), int(n_blocks), int(cells_per_block)
    update_order = np.asarray(update_order, dtype=np.int64)
    pos = np.arange(Ng, dtype=np.int64)
    blk = np.minimum(pos // cpb, nb - 1) if cpb > 0 else np.full(Ng, nb - 1, dtype=np.int64)
    if len(rank) != Ng or offset:
        mine = (update_order >= offset) & (update_order < offset + len(rank))
        update_order, blk = update_order[mine] - offset, blk[mine]
    N = len(update_order)
    cell_int = rank[update_order]
    key = blk * G + gid_int[cell_int]
    srt = np.argsort(key, kind="stable")
    key_s, cell_s = key[srt], cell_int[srt]
    counts = np.bincount(key_s, minlength=nb * G)
    tiles = -(-counts // TILE)
    pad_start = np.concatenate([[0], np.cumsum(tiles * TILE)])
    run_start = np.concatenate([[0], np.cumsum(counts)])
    within = np.arange(N, dtype=np.int64) - run_start[key_s]
    cells = np.full(int(pad_start[-1]), -1, dtype=np.int32)
    cells[pad_start[key_s] + within] = cell_s
    tile_grp = np.repeat(np.tile(np.arange(G, dtype=np.int32), nb), tiles)
    blk_tiles = tiles.reshape(nb, G).sum(axis=1)
    blk_start = np.concatenate([[0], np.cumsum(blk_tiles)]).astype(np.int32)
    return cells, tile_grp, blk_start


class Harmony:
    """Device-resident Harmony state with the reference's object API (harmony.py:218-569).

    ``Z`` is d x N (PCs x cells) like the reference's; ``Phi`` is the dense B x N indicator
    matrix of the reference or a ``BatchCodes``.  The constructor runs the whole algorithm
    (harmony.py:280-282); results are read through the same properties.
    """

    def __init__(
            self, Z, Phi, Pr_b, sigma, theta, lamb, alpha, lambda_estimation,
            max_iter_harmony, max_iter_kmeans,
            epsilon_kmeans, epsilon_harmony, K, block_size, verbose,
            random_state, device, shard=None, *, _y0=None, _schedule=None
    ):
        self.device = device
        self.shard = shard
        Z = np.asarray(Z, dtype=np.float32)
        self.d, self.N = Z.shape
        # cells of the whole job and this rank's first global cell id (N_global == N unsharded)
        self._offset, self.N_global = (0, self.N) if shard is None else shard.layout(self.N)
        self._codes = Phi if isinstance(Phi, BatchCodes) else BatchCodes.from_dense(Phi)
        if self._codes.codes.shape[0] != self.N:
            raise ValueError("Phi and Z disagree on the number of cells")
        self.B = self._codes.n_batches
        self.K = int(K)
        self.window_size = 3
        self.epsilon_kmeans = epsilon_kmeans
        self.epsilon_harmony = epsilon_harmony
        self.alpha = alpha
        self.lambda_estimation = bool(lambda_estimation)
        self.block_size = block_size
        self.max_iter_harmony = max_iter_harmony
        self.max_iter_kmeans = max_iter_kmeans
        self.verbose = verbose
        self._Pr_b = np.asarray(Pr_b, dtype=np.float32)
        self._sigma = np.asarray(sigma, dtype=np.float32)
        self._theta = np.asarray(theta, dtype=np.float32)
        self._lamb = np.asarray(lamb, dtype=np.float32)
        if self._sigma.shape != (self.K,):
            raise ValueError("sigma must have one entry per cluster")
        if self._theta.shape != (self.B,) or self._Pr_b.shape != (self.B,):
            raise ValueError("theta and Pr_b must have one entry per batch")
        if not self.lambda_estimation and self._lamb.shape != (self.B + 1,):
            raise ValueError("lamb must have B+1 entries (intercept first)")

        self.objective_harmony = []
        self.objective_kmeans = []
        self.objective_kmeans_dist = []
        self.objective_kmeans_entropy = []
        self.objective_kmeans_cross = []
        self.kmeans_rounds = []
        self._pending_objective = None
        # replay aids (see run_harmony): initial centroids and a fixed round schedule
        self._y0 = None if _y0 is None else np.asarray(_y0, dtype=np.float32)
        self._schedule = None if _schedule is None else [int(r) for r in _schedule]
        mode = os.environ.get("HMX_UPDATE_ORDER", UPDATE_ORDER)
        if mode not in ("auto", "torch", "device"):
            raise ValueError(f"HMX_UPDATE_ORDER={mode!r}: expected auto, torch or device")
        if mode == "auto":
            mode = "torch" if self.N_global <= AUTO_DEVICE_ORDER_CELLS else "device"
        self.update_order = mode
        # device order: the rounds of one cluster() call run inside the library (hmx_cluster).  HMX_CLUSTER_LOOP=python keeps
        # them in the Python loop below, one hmx_cluster_round_seeded per round -- the same rounds, the same test; it exists
        # so that tests can put the two drivers side by side (tests/test_parity_gpu.py)
        self._cluster_in_library = os.environ.get("HMX_CLUSTER_LOOP", "library").lower() != "python"
        self._seed = int(random_state) if random_state is not None else 0
        if verbose and mode == "device":
            logger.info("  update order: keyed bijection on the device (a different random stream than the reference's "
                        "torch.randperm; HMX_UPDATE_ORDER=torch replays the reference's)")

        # where the wall-clock of the constructor went (seconds per phase); read by bench.py
        self.timing = {}
        self._t_last = time.perf_counter()
        self.allocate_buffers(Z)
        self.init_cluster(random_state)
        self._lap("init_cluster")
        self.harmonize(self.max_iter_harmony, self.verbose)
        self._lap("harmonize")

    def _lap(self, name):
        now = time.perf_counter()
        self.timing[name] = self.timing.get(name, 0.0) + now - getattr(self, "_t_last", now)
        self._t_last = now

    # ------------------------------------------------------------------
    # device state (replaces harmony.py:234-271 uploads and :357-364 buffers)
    # ------------------------------------------------------------------
    def allocate_buffers(self, Z=None):
        codes = self._codes.codes
        V = codes.shape[1]
        # batch groups = distinct multi-hot rows of Phi; cells are stored group-sorted
        combos = None
        if self.shard is not None:   # the groups of the whole job
            parts = self.shard.allgather_object(np.unique(codes, axis=0))
            combos = np.unique(np.vstack(parts), axis=0)
        (self._group_cols, self._order, self._gid_int,
         self._static_cells, self._static_tile_grp) = build_layout(codes, combos)
        self._rank_cache = None
        self._lap("group_layout")
        self._G = self._group_cols.shape[0]
        self._n_blocks = int(np.ceil(1.0 / self.block_size))                     # harmony.py:474
        self._cells_per_block = int(self.N_global * self.block_size)             # harmony.py:475

        self._engine = _capi.Engine(self.N, self.d, self.K, self.B, self._G, V, self._n_blocks,
                                    lambda_estimation=self.lambda_estimation, alpha=self.alpha,
                                    device_id=_device_index(self.device), n_cells_global=self.N_global)
        self.transport = None if self.shard is None else self.shard.attach(self._engine)
        self._lap("engine_create")
        if Z is not None:
            # Z travels cells x d in the caller's order; the device regroups it (source_row).
            # A cell's id in the whole job = its row in the unsharded input.
            src = self._order.astype(np.int32)
            gid = src if self._offset == 0 else (self._offset + self._order).astype(np.int32)
            self._engine.upload(np.ascontiguousarray(Z.T), self._static_cells, self._static_tile_grp,
                                self._group_cols, self._Pr_b, self._theta, self._sigma,
                                None if self.lambda_estimation else self._lamb, global_id=gid, source_row=src)
            self._lap("upload")

    # ------------------------------------------------------------------
    # read-back (harmony.py:288-355): fresh float32 NumPy arrays, cells x features
    # ------------------------------------------------------------------
    def _rows(self, which):
        out = self._engine.get(which)
        res = np.empty_like(out)
        res[self._order] = out                    # internal (group-sorted) rows back to the caller's order
        return res

    @property
    def _rank(self):
        """original cell -> internal position (built on first use: the host update order, tests)."""
        if self._rank_cache is None:
            self._rank_cache = inverse_order(self._order)
        return self._rank_cache

    @property
    def Z_corr(self):
        """Corrected embedding (N x d)."""
        return self._rows(_capi.HMX_Z_CORR)

    @property
    def Z_orig(self):
        """Input embedding (N x d)."""
        return self._rows(_capi.HMX_Z_ORIG)

    @property
    def Z_cos(self):
        """L2-normalised embedding used for clustering (N x d)."""
        return self._rows(_capi.HMX_Z_COS)

    @property
    def R(self):
        """Soft cluster assignment (N x K)."""
        return self._rows(_capi.HMX_R)

    @property
    def Y(self):
        """Unit-length cluster centroids (d x K)."""
        return np.ascontiguousarray(self._engine.get(_capi.HMX_Y).T)

    @property
    def O(self):
        """Observed batch-by-cluster mass (K x B)."""
        og = self._engine.get(_capi.HMX_O_GROUP)                                 # G x K
        O = np.zeros((self.K, self.B), dtype=np.float64)
        for v in range(self._group_cols.shape[1]):
            np.add.at(O.T, self._group_cols[:, v], og)
        return O.astype(np.float32)

    @property
    def E(self):
        """Expected batch-by-cluster mass (K x B) = cluster mass x batch proportion."""
        T = self._engine.get(_capi.HMX_T_MASS).reshape(-1)
        return np.outer(T.astype(np.float32), self._Pr_b).astype(np.float32)

    @property
    def Phi(self):
        """One-hot batch indicators (N x B)."""
        return np.ascontiguousarray(self._codes.dense().T)

    @property
    def Phi_moe(self):
        """Batch indicators with a leading intercept column (N x (B+1))."""
        return np.concatenate([np.ones((self.N, 1), np.float32), self.Phi], axis=1)

    @property
    def Pr_b(self):
        return self._Pr_b.copy()

    @property
    def theta(self):
        return self._theta.copy()

    @property
    def sigma(self):
        return self._sigma.copy()

    @property
    def lamb(self):
        return self._lamb.copy()

    def result(self):
        """Corrected data as a NumPy array (N x d), harmony.py:353-355."""
        return self.Z_corr

    # ------------------------------------------------------------------
    # harmony.py:366-392
    # ------------------------------------------------------------------
    def init_cluster(self, random_state):
        if self._y0 is not None:
            Y0 = self._y0                                                        # d x K
        elif self._kmeans_mode() == "device":
            Y0 = self._device_kmeans(random_state)
        else:
            if self.verbose:
                logger.info("Computing initial centroids with sklearn.KMeans...")
            Z_cos = self.Z_cos                                                   # N x d, original order
            if self.shard is not None:
                # the reference fits on every cell (harmony.py:369-372); rank 0 does so on the
                # gathered cells up to KMEANS_GATHER_CELLS, above that on an even subsample
                take = np.arange(self.N)
                if self.N_global > KMEANS_GATHER_CELLS:
                    n = max(1, int(round(KMEANS_GATHER_CELLS * self.N / self.N_global)))
                    take = np.linspace(0, self.N - 1, n).astype(np.int64)
                parts = self.shard.allgather_object(Z_cos[take])
                Z_cos = np.concatenate(parts, axis=0) if self.shard.rank == 0 else None
            Y0 = None
            if Z_cos is not None:
                from sklearn.cluster import KMeans
                model = KMeans(n_clusters=self.K, init="k-means++", n_init=1, max_iter=25,
                               random_state=random_state)                        # harmony.py:370-371
                model.fit(Z_cos)
                Y0 = np.asarray(model.cluster_centers_.T, dtype=np.float32)
            if self.shard is not None:
                Y0 = self.shard.broadcast_object(Y0)
            if self.verbose:
                logger.info("KMeans initialization complete.")
        if Y0.shape != (self.d, self.K):
            raise ValueError("initial centroids must be d x K")
        self._Y0 = Y0
        triple = self._engine.init_cluster(np.ascontiguousarray(Y0.T))           # :376-391 on device
        self._pending_objective = triple
        self.compute_objective()
        self.objective_harmony.append(self.objective_kmeans[-1])                 # :392

    def _kmeans_mode(self):
        mode = os.environ.get("HMX_KMEANS", KMEANS)
        if mode not in ("auto", "host", "device"):
            raise ValueError(f"HMX_KMEANS={mode!r}: expected auto, host or device")
        if mode == "auto":
            mode = "host" if self.N_global <= KMEANS_DEVICE_CELLS else "device"
        return mode

    def _wide_shape(self):
        """Shapes the device Lloyd kernel is not built for (K > 112 or d > 64, e.g. BASELINE config 5)."""
        return self.K > 112 or self.d > 64

    def _device_kmeans(self, random_state):
        """k-means++ seeding on a subsample (device), Lloyd iterations over all cells (device)."""
        if self.verbose:
            logger.info("Computing initial centroids: k-means++ seeding on a subsample, Lloyd iterations on the GPU "
                        "(not sklearn's fit on all cells: HMX_KMEANS=host selects that)...")
        n = max(1, int(round(KMEANS_SEED_CELLS * self.N / self.N_global)))
        # evenly spaced over the group-sorted cells: every batch group in proportion
        take = np.linspace(0, self.N - 1, min(n, self.N)).astype(np.int64)
        sub = self._engine.get_rows(_capi.HMX_Z_COS, take)                       # unit rows of the sampled cells
        if self.shard is not None:
            parts = self.shard.allgather_object(sub)
            sub = np.concatenate(parts, axis=0) if self.shard.rank == 0 else None
        centers = None
        seeds = os.environ.get("HMX_KMEANS_SEEDS", KMEANS_SEEDS)
        if seeds not in ("device", "sklearn"):
            raise ValueError(f"HMX_KMEANS_SEEDS={seeds!r}: expected device or sklearn")
        if sub is not None and seeds == "device":
            centers, _ = self._engine.kmeans_seed(sub, 0 if random_state is None else int(random_state))
        elif sub is not None:
            from sklearn.cluster import kmeans_plusplus
            centers, _ = kmeans_plusplus(np.ascontiguousarray(sub, dtype=np.float32), n_clusters=self.K,
                                         random_state=random_state)
            centers = np.asarray(centers, dtype=np.float32)
        if self.shard is not None:
            centers = self.shard.broadcast_object(centers)
        self._lap("kmeans_seeds")
        # max_iter=25 (harmony.py:371) Lloyd iterations over ALL cells on the device -- wide shapes (K > 112 or d > 64) too:
        # hard assignment as a one-hot R, member sums as the R^T.Z statistics of it (hmx_kmeans_lloyd)
        # wide shapes only: the device Lloyd needs the streaming R^T.Z pass with one block column (hmx_capi.cpp, lloyd_wide);
        # more batch groups than its finish kernel tabulates (85 at 200 PCs), or a caller-built layout whose static tiles
        # do not hold consecutive cells, run the 25 iterations on a subsample on the host instead (sklearn's Lloyd from
        # the same seeds -- what harmony.py:370-372 does on all cells).  The engine is ASKED first (hmx_can_lloyd) and a
        # sharded job decides together, before any rank enters the iterations' all-reduces; whatever the call itself
        # raises afterwards (a HIP error, a bad state) is a failure, not a reason to fall back.
        can = self._engine.can_lloyd()
        if self.shard is not None:
            can = all(self.shard.allgather_object(bool(can)))
        if can:
            centers = self._engine.kmeans_lloyd(centers, 25)
        else:
            logger.warning("device k-means unavailable for this shape or layout (hmx_can_lloyd); Lloyd iterations on a subsample on the host")
            centers = self._host_lloyd_on_subsample(centers, random_state)
        self._lap("kmeans_lloyd")
        if self.verbose:
            logger.info("KMeans initialization complete.")
        return np.ascontiguousarray(centers.T)

    def _host_lloyd_on_subsample(self, centers, random_state, cells=200_000):
        from sklearn.cluster import KMeans
        n = max(1, int(round(cells * self.N / self.N_global)))
        take = np.linspace(0, self.N - 1, min(n, self.N)).astype(np.int64)
        sub = self._engine.get_rows(_capi.HMX_Z_COS, take)
        if self.shard is not None:
            parts = self.shard.allgather_object(sub)
            sub = np.concatenate(parts, axis=0) if self.shard.rank == 0 else None
        out = None
        if sub is not None:
            km = KMeans(n_clusters=self.K, init=np.ascontiguousarray(centers, dtype=np.float32), n_init=1, max_iter=25,
                        random_state=random_state).fit(np.ascontiguousarray(sub, dtype=np.float32))
            out = np.asarray(km.cluster_centers_, dtype=np.float32)
        if self.shard is not None:
            out = self.shard.broadcast_object(out)
        return out

    # ------------------------------------------------------------------
    # harmony.py:394-417: the three sums come back from the device with the round
    # ------------------------------------------------------------------
    def compute_objective(self):
        """Appends the objective of the current state to the history lists (harmony.py:394-417).

        The three sums are produced on the device by the sweep that last changed R, O and E
        (``update_R`` / ``init_cluster``).  The reference evaluates them from ``_R``, ``_dist_mat``,
        ``_O`` and ``_E``, none of which change between such sweeps (``moe_correct_ridge`` touches
        only Z_corr / Z_cos / W), so a further call appends the same values again -- as there."""
        if self._pending_objective is not None:
            self._last_objective = tuple(float(x) for x in self._pending_objective[:3])
            self._pending_objective = None
        if getattr(self, "_last_objective", None) is None:
            raise RuntimeError("compute_objective() needs init_cluster() or update_R() first")
        kmeans_error, _entropy, _cross_entropy = self._last_objective
        norm_const = 2000.0 / self.N_global
        self.objective_kmeans.append((kmeans_error + _entropy + _cross_entropy) * norm_const)
        self.objective_kmeans_dist.append(kmeans_error * norm_const)
        self.objective_kmeans_entropy.append(_entropy * norm_const)
        self.objective_kmeans_cross.append(_cross_entropy * norm_const)

    # ------------------------------------------------------------------
    # harmony.py:419-435
    # ------------------------------------------------------------------
    def harmonize(self, iter_harmony=10, verbose=True):
        converged = False
        for i in range(1, iter_harmony + 1):
            if verbose:
                logger.info(f"Iteration {i} of {iter_harmony}")
            self.cluster()
            self.moe_correct_ridge()
            converged = self.check_convergence(1)
            if converged:
                if verbose:
                    logger.info(f"Converged after {i} iteration{'s' if i > 1 else ''}")
                break
        if verbose and not converged:
            logger.info("Stopped before convergence")

    # ------------------------------------------------------------------
    # harmony.py:437-462
    # ------------------------------------------------------------------
    def cluster(self, *, _rounds=None):
        """``_rounds`` (keyword only, not in the reference): run exactly that many rounds."""
        rounds = 0
        forced = _rounds if _rounds is not None else (self._schedule.pop(0) if self._schedule else None)
        if self.update_order == "device" and self._cluster_in_library:
            # device-side update order: all rounds of this call inside the library (hmx_cluster) -- the same rounds, the
            # same test on the same numbers (harmony.py:455-458, 517-523), no trip through Python between rounds
            terms = self._engine.cluster(self._seed, self._cells_per_block, self.max_iter_kmeans, forced,
                                         self.window_size, self.epsilon_kmeans)
            for t in terms:
                self._pending_objective = t
                self.compute_objective()                                         # :453
            self.kmeans_rounds.append(len(terms))
            self.objective_harmony.append(self.objective_kmeans[-1])
            return
        for i in range(self.max_iter_kmeans if forced is None else forced):
            self._round(_capi.HMX_ROUND_ALL)                                     # :443-450
            self.compute_objective()                                             # :453
            if forced is None and i > self.window_size:
                if self.check_convergence(0):                                    # :455-458
                    rounds = i + 1
                    break
            rounds = i + 1
        self.kmeans_rounds.append(rounds)
        self.objective_harmony.append(self.objective_kmeans[-1])

    def update_R(self):
        """harmony.py:464-513 (one blocked sweep over a fresh random order)."""
        self._round(_capi.HMX_ROUND_UPDATE_R | _capi.HMX_ROUND_OBJECTIVE)

    def _update_order(self):
        import torch
        return torch.randperm(self.N_global).numpy()                             # harmony.py:471 (same stream on every rank)

    def _round(self, flags):
        if self.update_order == "device":
            self._pending_objective = self._engine.cluster_round_seeded(self._seed, self._cells_per_block, flags)
            return
        order = self._update_order()
        cells, tile_grp, blk_start = self._block_lists(order)
        self._pending_objective = self._engine.cluster_round(cells, tile_grp, blk_start, flags)

    def _block_lists(self, update_order):
        return build_block_lists(update_order, self._rank, self._gid_int, self._n_blocks,
                                 self._cells_per_block, self._G, offset=self._offset)

    # ------------------------------------------------------------------
    # harmony.py:515-533
    # ------------------------------------------------------------------
    def check_convergence(self, i_type):
        if i_type == 0:
            if len(self.objective_kmeans) <= self.window_size + 1:
                return False
            w = self.window_size
            obj_old = sum(self.objective_kmeans[-w - 1:-1])
            obj_new = sum(self.objective_kmeans[-w:])
            return abs(obj_old - obj_new) / abs(obj_old) < self.epsilon_kmeans
        if i_type == 1:
            if len(self.objective_harmony) < 2:
                return False
            obj_old = self.objective_harmony[-2]
            obj_new = self.objective_harmony[-1]
            return (obj_old - obj_new) / abs(obj_old) < self.epsilon_harmony
        return True

    # ------------------------------------------------------------------
    # harmony.py:535-569
    # ------------------------------------------------------------------
    def moe_correct_ridge(self):
        """Ridge regression correction for batch effects."""
        self._engine.moe_correct_ridge()
