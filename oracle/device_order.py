"""NumPy restatement of the engine's device-side update order.  TEST INFRASTRUCTURE ONLY.

The reference draws ``torch.randperm(N)`` per ``update_R`` call and cuts it into blocks
(harmony.py:471-484).  For large jobs the engine replaces the permutation by a keyed bijection pi
of [0, N_global) evaluated on the GPU (hmx_cluster_round_seeded): position of a cell
= pi^-1(global id) = six inverse Feistel rounds with cycle walking; round keys = splitmix64 of
(seed, round counter); block = min(position // cells_per_block, n_blocks - 1); inside a block cells
are grouped by batch group, in increasing internal id, every (block, group) run padded to 16 with
-1.  This file states exactly that with integer NumPy arithmetic, so the GPU tests can compare
the engine's lists bit for bit.
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _mix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def round_keys(seed, counter):
    """splitmix64 of (seed, round counter) -> (key0, key1)."""
    mask = (1 << 64) - 1
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(counter) + 1)) & mask
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    z ^= z >> 31
    return z & 0xFFFFFFFF, ((z >> 32) & 0xFFFFFFFF) | 1


def positions(global_ids, n_global, seed, counter):
    """pi^-1 of every global id."""
    bits = 1
    while (1 << bits) < n_global:
        bits += 1
    half = (bits + 1) // 2
    mask = np.uint64((1 << half) - 1)
    k0, k1 = round_keys(seed, counter)
    x = np.asarray(global_ids, dtype=np.uint64).copy()
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        v = x[todo]
        l, r = v >> np.uint64(half), v & mask
        for i in range(5, -1, -1):
            f = _mix32(((l * np.uint64(0x9E3779B1)) + np.uint64(k0) + np.uint64(i) * np.uint64(k1)) & M32) & mask
            l, r = r ^ f, l
        v = (l << np.uint64(half)) | r
        x[todo] = v
        todo[todo] = v >= np.uint64(n_global)
    return x.astype(np.int64)


def block_lists(global_ids, group_of_cell, n_groups, n_global, seed, counter, cells_per_block, n_blocks, tile=16):
    """(cells, tile_group, block_tile_start) exactly as the engine builds them (internal cell ids)."""
    pos = positions(global_ids, n_global, seed, counter)
    blk = np.minimum(pos // cells_per_block, n_blocks - 1) if cells_per_block > 0 else np.full(len(pos), n_blocks - 1)
    key = blk * n_groups + np.asarray(group_of_cell, dtype=np.int64)
    cells, tgrp, bstart = [], [], [0]
    for b in range(n_blocks):
        for g in range(n_groups):
            members = np.flatnonzero(key == b * n_groups + g).astype(np.int32)      # increasing internal id
            nt = -(-len(members) // tile)
            seg = np.full(nt * tile, -1, dtype=np.int32)
            seg[:len(members)] = members
            cells.append(seg)
            tgrp.append(np.full(nt, g, dtype=np.int32))
        bstart.append(bstart[-1] + sum(len(t) for t in tgrp[-n_groups:]))
    return np.concatenate(cells), np.concatenate(tgrp), np.asarray(bstart, dtype=np.int32)
