"""Cells sharded over several engines: one process per GPU, ``torch.distributed`` for the plumbing.

The reference is strictly single-device (harmony.py:230); this module is what lets the same
``run_harmony`` / ``Harmony`` API run with every rank holding a contiguous slice of the cells
(SURVEY.md §8e).  All per-cell work of the ``harmonize()`` loop is independent given a handful of
small tables (centroids K x d, batch-by-cluster sums K x B, the ridge statistics); the engine sums
exactly those tables over the ranks (``include/hmx.h``: ``hmx_comm_init`` = RCCL over xGMI on the
engine's own stream, ``hmx_set_host_allreduce`` = any other transport).  ``Shard`` carries what the
Python side needs besides: rank/world, this rank's offset in the global cell order, and host-level
helpers (sums of a few counters, exchange of category labels, the RCCL unique id).

Usage (every rank, after ``torch.distributed.init_process_group``)::

    shard = harmonypy_amd.Shard()                      # default process group
    ho = harmonypy_amd.run_harmony(Z_local, meta_local, ["batch"], shard=shard)
    ho.Z_corr                                          # this rank's cells x PCs
"""
from __future__ import annotations

import numpy as np


class Shard:
    """This rank's place in a job whose cells are sharded by rank (rank r holds the r-th slice).

    ``transport``: ``"rccl"`` -- the engine's own RCCL communicator (device-side, stream-ordered);
    ``"host"`` -- tables staged through host memory and summed with ``torch.distributed`` on the
    given group (works with gloo; used by the tests); ``"auto"`` -- RCCL when the group's backend
    is nccl, else host.
    """

    def __init__(self, group=None, transport="auto"):
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before creating a Shard")
        if transport not in ("auto", "rccl", "host"):
            raise ValueError("transport must be auto, rccl or host")
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = str(dist.get_backend(group))
        self.transport = transport if transport != "auto" else ("rccl" if "nccl" in self.backend else "host")
        self.offset = 0          # set by layout(): first global cell id of this rank
        self.n_local = 0
        self.n_global = 0
        self.n_collectives = 0   # host-level collectives issued through this object

    # ---- host-level collectives (tiny payloads) ----------------------------------------------
    def _tensor(self, arr):
        import torch
        t = torch.from_numpy(arr)
        return t.cuda() if "nccl" in self.backend else t

    def allreduce_(self, arr):
        """Sum a float64/int64 NumPy array over the ranks, in place."""
        assert isinstance(arr, np.ndarray) and arr.flags.c_contiguous and arr.dtype in (np.float64, np.int64)
        t = self._tensor(arr)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        if t.device.type != "cpu":
            arr[...] = t.cpu().numpy()
        self.n_collectives += 1
        return arr

    def allgather_object(self, obj):
        out = [None] * self.world
        self._dist.all_gather_object(out, obj, group=self.group)
        self.n_collectives += 1
        return out

    def broadcast_object(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src, group=self.group)
        self.n_collectives += 1
        return box[0]

    def barrier(self):
        self._dist.barrier(group=self.group)

    # ---- layout --------------------------------------------------------------------------------
    def layout(self, n_local):
        """Exchange the shard sizes; returns (offset of this rank, cells of the whole job)."""
        sizes = np.zeros(self.world, dtype=np.int64)
        sizes[self.rank] = int(n_local)
        self.allreduce_(sizes)
        self.n_local = int(n_local)
        self.offset = int(sizes[:self.rank].sum())
        self.n_global = int(sizes.sum())
        self.sizes = sizes
        return self.offset, self.n_global

    # ---- engine transport ------------------------------------------------------------------------
    def attach(self, engine):
        """Give ``engine`` (a ``_capi.Engine``) its transport.  If RCCL cannot be brought up on
        every rank, all ranks fall back to the host transport together (with a warning): a
        sharded job never silently runs unsharded.  On top of either transport the per-block
        sums of ``update_R`` travel through peer boxes inside the sweep kernel when every rank's
        self-test of that exchange succeeds (``HMX_PEER_EXCHANGE=0`` disables the attempt)."""
        name = self._attach_transport(engine)
        self.peer_exchange = self._attach_peers(engine)
        return name + ("+peer" if self.peer_exchange else "")

    def _attach_peers(self, engine):
        """Peer boxes for the in-kernel exchange of the per-block sums.  First attempt: FINE-GRAINED device memory (what HIP keeps
        coherent for another device's writes while a kernel polls; the engine's default).  If any rank cannot export, map or
        pass the self-test with those -- e.g. a stack whose IPC does not open fine-grained allocations across devices -- all
        ranks together try once more with plain (coarse-grained) boxes, the allocation of rounds 3-5; if that fails too, one
        collective per update block.  ``HMX_PEER_BOX`` pins the kind (no second attempt)."""
        import logging
        import os
        from . import _capi
        if os.environ.get("HMX_PEER_EXCHANGE", "1") == "0" or self.world > 8:
            return False
        pinned = os.environ.get("HMX_PEER_BOX")
        err = ""
        try:
            engine.set_ranks(self.world, self.rank)
            ranks_ok = 1
        except _capi.HmxError as exc:
            ranks_ok, err = 0, str(exc)
        for kind in ([pinned] if pinned else ["fine", "coarse"]):
            ok, handle = ranks_ok, b""
            os.environ["HMX_PEER_BOX"] = kind                      # read by hmx_peer_export
            try:
                if ok:
                    handle = engine.peer_export()
            except _capi.HmxError as exc:
                ok, err = 0, str(exc)
            finally:
                if pinned is None:
                    os.environ.pop("HMX_PEER_BOX", None)
            handles = self.allgather_object(handle)
            if ok and all(len(h) == _capi.HMX_PEER_HANDLE_BYTES for h in handles):
                try:
                    engine.peer_attach(b"".join(handles))
                except _capi.HmxError as exc:
                    ok, err = 0, str(exc)
            else:
                ok = 0
            flag = np.array([ok], dtype=np.int64)
            self.allreduce_(flag)                          # also a barrier: every box is mapped everywhere
            if int(flag[0]) == self.world:
                try:
                    ok = 1 if engine.peer_selftest() else 0
                except _capi.HmxError as exc:
                    ok, err = 0, str(exc)
                flag = np.array([ok], dtype=np.int64)
                self.allreduce_(flag)
            if int(flag[0]) == self.world:
                engine.peer_enable(True)
                return True
            logging.getLogger("harmonypy_amd").warning(
                f"rank {self.rank}: in-kernel peer exchange unavailable with {kind}-grained boxes {err}")
        logging.getLogger("harmonypy_amd").warning(
            f"rank {self.rank}: in-kernel peer exchange unavailable; one collective per update block instead")
        return False

    def _attach_transport(self, engine):
        import logging
        from . import _capi
        if self.transport == "rccl":
            uid, err = None, ""
            if self.rank == 0:
                try:
                    uid = _capi.Engine.comm_unique_id()
                except _capi.HmxError as exc:
                    err = str(exc)
            uid = self.broadcast_object(uid)
            ok = 0
            if uid is not None:
                try:
                    engine.comm_init(uid, self.world, self.rank)
                    ok = 1
                except _capi.HmxError as exc:
                    err = str(exc)
            flag = np.array([ok], dtype=np.int64)
            self.allreduce_(flag)
            if int(flag[0]) == self.world:
                return "rccl"
            logging.getLogger("harmonypy_amd").warning(
                f"rank {self.rank}: RCCL transport unavailable on {self.world - int(flag[0])} rank(s) {err}; "
                "using the host transport")
        engine.set_host_allreduce(self.allreduce_)
        return "host"
