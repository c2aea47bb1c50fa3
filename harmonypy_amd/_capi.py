"""ctypes binding of libhmx.so (include/hmx.h).  Thin: argument marshalling and error text only.

The library is the product's compute path; there is no fallback.  Importing this module
without a built ``libhmx.so`` raises, and every entry point raises ``HmxError`` with the
library's message when a call fails (for instance when no MI355X is visible).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMX_LIB") or os.path.join(_HERE, "libhmx.so")   # HMX_LIB: variant builds for timing experiments

# mirrors include/hmx.h
HMX_TILE = 16
HMX_MAX_CLUSTERS, HMX_MAX_PCS, HMX_MAX_BLOCKS, HMX_MAX_VARS = 320, 320, 250, 32   # limits checked by hmx_create
HMX_Z_ORIG, HMX_Z_COS, HMX_Z_CORR, HMX_R, HMX_Y, HMX_O_GROUP, HMX_T_MASS, HMX_W = range(8)
HMX_ROUND_BLOCK_START, HMX_ROUND_CELLS, HMX_ROUND_TILE_GROUP = 8, 9, 10
HMX_ROUND_CENTROIDS, HMX_ROUND_UPDATE_R, HMX_ROUND_OBJECTIVE = 1, 2, 4
HMX_ROUND_ALL = 7

EXPORTS = [
    "hmx_last_error", "hmx_abi_version", "hmx_create", "hmx_destroy", "hmx_upload", "hmx_init_cluster",
    "hmx_cluster_round", "hmx_cluster_round_seeded", "hmx_moe_correct_ridge", "hmx_get", "hmx_set", "hmx_sync", "hmx_device_ptr",
    "hmx_kernel_times", "hmx_enable_timing", "hmx_counters", "hmx_comm_unique_id", "hmx_comm_init", "hmx_set_host_allreduce",
    "hmx_build_id", "hmx_cluster", "hmx_set_timing_stride", "hmx_set_timing_families", "hmx_kmeans_lloyd", "hmx_can_lloyd", "hmx_kmeans_seed", "hmx_compute_lisi", "hmx_get_rows", "hmx_set_ranks", "hmx_peer_export", "hmx_peer_attach", "hmx_peer_selftest", "hmx_peer_enable",
]
HMX_PEER_HANDLE_BYTES = 64
HMX_ABI_VERSION = 8
HMX_UNIQUE_ID_BYTES = 128
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)


class HmxError(RuntimeError):
    """A libhmx call returned a negative code.  ``code`` is that return value (HMX_ERR_ARG = -1 bad argument / unsupported
    shape, HMX_ERR_HIP = -2, HMX_ERR_STATE = -3, HMX_ERR_COMM = -4; None when the library could not be loaded)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


HMX_ERR_ARG, HMX_ERR_HIP, HMX_ERR_STATE, HMX_ERR_COMM = -1, -2, -3, -4


class HmxConfig(C.Structure):
    _fields_ = [
        ("n_cells", C.c_int64), ("n_cells_global", C.c_int64), ("n_pcs", C.c_int32), ("n_clusters", C.c_int32),
        ("n_batches", C.c_int32), ("n_groups", C.c_int32), ("n_vars", C.c_int32), ("n_blocks", C.c_int32),
        ("device_id", C.c_int32), ("lambda_estimation", C.c_int32), ("alpha", C.c_float), ("reserved", C.c_int32 * 4),
    ]


_lib = None


def load():
    """dlopen libhmx.so and declare prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HmxError(f"{LIB_PATH} is missing: build it with `python -m harmonypy_amd._build` "
                       "(hipcc --offload-arch=gfx950); harmonypy_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.hmx_last_error.restype = C.c_char_p
    lib.hmx_last_error.argtypes = []
    lib.hmx_abi_version.restype = C.c_int
    lib.hmx_create.argtypes = [C.POINTER(HmxConfig), C.POINTER(vp)]
    lib.hmx_destroy.argtypes = [vp]
    lib.hmx_destroy.restype = None
    lib.hmx_upload.argtypes = [vp, vp, vp, i64, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.hmx_kmeans_seed.argtypes = [vp, vp, i64, C.c_uint64, vp, vp]
    lib.hmx_compute_lisi.argtypes = [i32, vp, i64, i32, vp, i32, C.c_double, vp, vp, vp]
    lib.hmx_comm_unique_id.argtypes = [vp]
    lib.hmx_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.hmx_set_host_allreduce.argtypes = [vp, HOST_ALLREDUCE_FN, vp]
    lib.hmx_set_ranks.argtypes = [vp, C.c_int, C.c_int]
    lib.hmx_peer_export.argtypes = [vp, vp]
    lib.hmx_peer_attach.argtypes = [vp, vp]
    lib.hmx_peer_selftest.argtypes = [vp]
    lib.hmx_peer_enable.argtypes = [vp, C.c_int]
    lib.hmx_init_cluster.argtypes = [vp, vp, vp]
    lib.hmx_kmeans_lloyd.argtypes = [vp, vp, C.c_int, vp]
    lib.hmx_can_lloyd.argtypes = [vp]
    lib.hmx_cluster_round.argtypes = [vp, C.c_int, vp, i64, vp, i32, vp, vp]
    lib.hmx_cluster_round_seeded.argtypes = [vp, C.c_int, C.c_uint64, i64, vp]
    lib.hmx_cluster.argtypes = [vp, C.c_uint64, i64, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp]
    lib.hmx_build_id.argtypes = []
    lib.hmx_moe_correct_ridge.argtypes = [vp]
    lib.hmx_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.hmx_set.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.hmx_get_rows.argtypes = [vp, C.c_int, vp, i32, vp, C.c_size_t]
    lib.hmx_sync.argtypes = [vp]
    lib.hmx_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.hmx_kernel_times.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_char_p)]
    lib.hmx_counters.argtypes = [vp, vp]
    lib.hmx_enable_timing.argtypes = [vp, C.c_int]
    lib.hmx_set_timing_stride.argtypes = [vp, C.c_int]
    lib.hmx_set_timing_families.argtypes = [vp, C.c_uint]
    for name in EXPORTS:
        if name not in ("hmx_last_error", "hmx_destroy", "hmx_build_id"):
            getattr(lib, name).restype = C.c_int
    lib.hmx_build_id.restype = C.c_char_p
    _lib = lib
    return lib


def build_id() -> str:
    """Identity of the kernel set in the loaded library (hash of csrc/ and hmx.h taken at build time)."""
    return load().hmx_build_id().decode()


def _check(rc):
    if rc < 0:
        raise HmxError(f"libhmx: {load().hmx_last_error().decode(errors='replace')} (code {rc})", code=int(rc))
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


KERNEL_FAMILIES = ["assign_block", "assign_init", "rtz_round", "rtz_reduce", "block_table",
                   "ridge_stats", "ridge_solve", "ridge_apply"]


class Engine:
    """One device-resident Harmony state (an ``hmx_engine``)."""

    def __init__(self, n_cells, n_pcs, n_clusters, n_batches, n_groups, n_vars, n_blocks,
                 lambda_estimation=False, alpha=0.2, device_id=0, n_cells_global=0):
        self._lib = load()
        self._h = C.c_void_p()
        self._host_cb = None
        cfg = HmxConfig(n_cells=n_cells, n_cells_global=n_cells_global, n_pcs=n_pcs, n_clusters=n_clusters, n_batches=n_batches,
                        n_groups=n_groups, n_vars=n_vars, n_blocks=n_blocks, device_id=device_id,
                        lambda_estimation=int(bool(lambda_estimation)), alpha=float(alpha))
        _check(self._lib.hmx_create(C.byref(cfg), C.byref(self._h)))
        self.N, self.d, self.K, self.B, self.G, self.V, self.nblk = (
            n_cells, n_pcs, n_clusters, n_batches, n_groups, n_vars, n_blocks)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hmx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, Z, static_cells, static_tile_group, group_cols, Pr_b, theta, sigma, lamb, global_id=None,
               source_row=None):
        """Z: N x d; with ``source_row`` in the caller's order (internal cell i = Z[source_row[i]])."""
        Z = _c(Z, np.float32)
        global_id = None if global_id is None else _c(global_id, np.int32)
        source_row = None if source_row is None else _c(source_row, np.int32)
        sc = _c(static_cells, np.int32)
        tg = _c(static_tile_group, np.int32)
        gc = _c(group_cols, np.int32)
        lamb = None if lamb is None else _c(lamb, np.float32)
        assert Z.shape == (self.N, self.d)
        _check(self._lib.hmx_upload(self._h, _ptr(Z), _ptr(sc), sc.size, _ptr(tg), tg.size, _ptr(gc),
                                    _ptr(_c(Pr_b, np.float32)), _ptr(_c(theta, np.float32)),
                                    _ptr(_c(sigma, np.float32)), _ptr(lamb), _ptr(global_id), _ptr(source_row)))

    # ---- transports of a sharded job (include/hmx.h) ------------------------------------------
    def comm_init(self, unique_id: bytes, n_ranks: int, rank: int):
        """RCCL communicator bound to the engine's stream."""
        assert len(unique_id) == HMX_UNIQUE_ID_BYTES
        buf = C.create_string_buffer(unique_id, HMX_UNIQUE_ID_BYTES)
        _check(self._lib.hmx_comm_init(self._h, buf, int(n_ranks), int(rank)))

    def set_ranks(self, n_ranks, rank):
        _check(self._lib.hmx_set_ranks(self._h, int(n_ranks), int(rank)))

    def peer_export(self) -> bytes:
        """Inter-process handle of this engine's peer box (in-kernel exchange of the block sums)."""
        buf = C.create_string_buffer(HMX_PEER_HANDLE_BYTES)
        _check(self._lib.hmx_peer_export(self._h, buf))
        return buf.raw

    def peer_attach(self, handles: bytes):
        buf = C.create_string_buffer(handles, len(handles))
        _check(self._lib.hmx_peer_attach(self._h, buf))

    def peer_selftest(self) -> bool:
        return _check(self._lib.hmx_peer_selftest(self._h)) == 1

    def peer_enable(self, on=True):
        _check(self._lib.hmx_peer_enable(self._h, int(bool(on))))

    def set_host_allreduce(self, fn):
        """``fn(np.ndarray[float64])`` must sum the array over all ranks in place."""
        if fn is None:
            _check(self._lib.hmx_set_host_allreduce(self._h, C.cast(None, HOST_ALLREDUCE_FN), None))
            self._host_cb = None
            return

        def _cb(_ctx, buf, count):
            try:
                fn(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:                                   # never let an exception cross the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._host_cb = HOST_ALLREDUCE_FN(_cb)                  # keep the thunk alive
        _check(self._lib.hmx_set_host_allreduce(self._h, self._host_cb, None))

    def kmeans_seed(self, points, seed=0):
        """k-means++ seeding on the device over ``points`` (n x d unit rows); returns (centres K x d, chosen)."""
        pts = _c(points, np.float32)
        assert pts.ndim == 2 and pts.shape[1] == self.d
        out = np.empty((self.K, self.d), np.float32)
        chosen = np.empty(self.K, np.int32)
        _check(self._lib.hmx_kmeans_seed(self._h, _ptr(pts), pts.shape[0], int(seed) & (2**64 - 1), _ptr(out), _ptr(chosen)))
        return out, chosen

    def can_lloyd(self) -> bool:
        """Whether kmeans_lloyd serves this engine's shape and layout (hmx_can_lloyd)."""
        return bool(_check(self._lib.hmx_can_lloyd(self._h)))

    def kmeans_lloyd(self, centers, n_iter=25):
        """Lloyd iterations over all cells of Z_cos on the device; centres K x d in and out."""
        cin = _c(centers, np.float32)
        assert cin.shape == (self.K, self.d)
        out = np.empty((self.K, self.d), np.float32)
        _check(self._lib.hmx_kmeans_lloyd(self._h, _ptr(cin), int(n_iter), _ptr(out)))
        return out

    def init_cluster(self, Y0_rows):
        Y0 = _c(Y0_rows, np.float32)
        assert Y0.shape == (self.K, self.d)
        out = np.zeros(4, np.float64)
        _check(self._lib.hmx_init_cluster(self._h, _ptr(Y0), _ptr(out)))
        return out

    def cluster_round(self, cells, tile_group, block_tile_start, flags=HMX_ROUND_ALL):
        cells = _c(cells, np.int32)
        tg = _c(tile_group, np.int32)
        bs = _c(block_tile_start, np.int32)
        assert bs.size == self.nblk + 1
        out = np.zeros(4, np.float64)
        _check(self._lib.hmx_cluster_round(self._h, flags, _ptr(cells), cells.size, _ptr(tg), tg.size,
                                           _ptr(bs), _ptr(out)))
        return out

    def cluster_round_seeded(self, seed, cells_per_block, flags=HMX_ROUND_ALL):
        out = np.zeros(4, np.float64)
        _check(self._lib.hmx_cluster_round_seeded(self._h, flags, int(seed) & (2**64 - 1), int(cells_per_block), _ptr(out)))
        return out

    def cluster(self, seed, cells_per_block, max_rounds, forced_rounds=None, window=3, epsilon=1e-5):
        """All rounds of one ``cluster()`` call on the device-side update order (hmx_cluster): returns the
        (rounds x 4) objective terms of the rounds that ran."""
        n = int(max_rounds if forced_rounds is None else forced_rounds)
        out = np.zeros((max(n, 1), 4), np.float64)
        done = C.c_int32(0)
        _check(self._lib.hmx_cluster(self._h, int(seed) & (2**64 - 1), int(cells_per_block), int(max_rounds),
                                     -1 if forced_rounds is None else int(forced_rounds), int(window), float(epsilon),
                                     _ptr(out), C.byref(done)))
        return out[:done.value]

    def moe_correct_ridge(self):
        _check(self._lib.hmx_moe_correct_ridge(self._h))

    _SHAPES = {
        HMX_Z_ORIG: ("N", "d", np.float32), HMX_Z_COS: ("N", "d", np.float32), HMX_Z_CORR: ("N", "d", np.float32),
        HMX_R: ("N", "K", np.float32), HMX_Y: ("K", "d", np.float32), HMX_O_GROUP: ("G", "K", np.float64),
        HMX_T_MASS: (1, "K", np.float64),
    }

    def _shape(self, which):
        if which == HMX_W:
            return (self.G, self.K, self.d), np.float32
        r, c, dt = self._SHAPES[which]
        r = getattr(self, r) if isinstance(r, str) else r
        return (r, getattr(self, c)), dt

    def get(self, which):
        shape, dt = self._shape(which)
        out = np.empty(shape, dt)
        _check(self._lib.hmx_get(self._h, which, _ptr(out), out.nbytes))
        return out

    def get_rows(self, which, rows):
        """A few rows of an N-sized float array (internal order), without downloading all of it."""
        rows = _c(rows, np.int32)
        shape, dt = self._shape(which)
        assert dt == np.float32
        out = np.empty((len(rows), shape[1]), np.float32)
        _check(self._lib.hmx_get_rows(self._h, which, _ptr(rows), len(rows), _ptr(out), out.nbytes))
        return out

    def set(self, which, arr):
        shape, dt = self._shape(which)
        arr = _c(arr, dt)
        assert arr.shape == tuple(shape), (arr.shape, shape)
        _check(self._lib.hmx_set(self._h, which, _ptr(arr), arr.nbytes))

    def round_lists(self):
        """(cells, tile_group, block_tile_start) of the last round, as the device holds them."""
        bs = np.empty(self.nblk + 1, np.int32)
        _check(self._lib.hmx_get(self._h, HMX_ROUND_BLOCK_START, _ptr(bs), bs.nbytes))
        nt = int(bs[-1])
        cells = np.empty(nt * HMX_TILE, np.int32)
        tg = np.empty(nt, np.int32)
        if nt:
            _check(self._lib.hmx_get(self._h, HMX_ROUND_CELLS, _ptr(cells), cells.nbytes))
            _check(self._lib.hmx_get(self._h, HMX_ROUND_TILE_GROUP, _ptr(tg), tg.nbytes))
        return cells, tg, bs

    def sync(self):
        _check(self._lib.hmx_sync(self._h))

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(HMX_UNIQUE_ID_BYTES)
        _check(load().hmx_comm_unique_id(buf))
        return buf.raw

    def enable_timing(self, on=True, families=None, stride=1):
        """Bracket kernel launches with HIP events (``kernel_times``).  ``families``: names from ``KERNEL_FAMILIES`` to
        bracket (default all); ``stride``: only every stride-th launch of a family.  Every bracketed launch costs two
        event records on the stream."""
        _check(self._lib.hmx_set_timing_stride(self._h, int(stride)))
        if on:
            mask = 0xFFFFFFFF if families is None else sum(1 << KERNEL_FAMILIES.index(f) for f in families)
            _check(self._lib.hmx_set_timing_families(self._h, mask))
        _check(self._lib.hmx_enable_timing(self._h, 1 if on else 0))

    def counters(self):
        """dict of the engine's event counters (hmx_counters)."""
        buf = np.zeros(16, np.int64)
        _check(self._lib.hmx_counters(self._h, _ptr(buf)))
        return {"collectives": int(buf[0]), "sweep_fallbacks": int(buf[1]), "seeded_rounds": int(buf[2]), "sweeps_bf16_pipe": int(buf[3]),
                "sweep_waits": int(buf[4]), "sweep_wait_polls": int(buf[5]), "sweep_wait_polls_max": int(buf[6]), "rtz_bf16_pipe": int(buf[7]),
                "sweeps_group_affine": int(buf[8]), "sweep_group_affine_wgs": int(buf[9]),
                "peer_box": {0: "none", 1: "coarse", 2: "fine"}[int(buf[10])], "rtz_presplit_z": int(buf[11]),
                "sweeps_wide_persistent": int(buf[12])}

    def kernel_times(self):
        """{family: (total_ms, launches)} since timing was enabled."""
        buf = np.zeros(2 * len(KERNEL_FAMILIES), np.float64)
        _check(self._lib.hmx_kernel_times(self._h, _ptr(buf), buf.size, None))
        return {n: (float(buf[2 * i]), int(buf[2 * i + 1])) for i, n in enumerate(KERNEL_FAMILIES)}
