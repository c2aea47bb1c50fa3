"""Index maps of the streaming R^T.Z pass (csrc/hmx_rtz3.hip), replayed in NumPy -- no GPU.

k_rtz3 permutes the MFMA row / column / k indices so that a lane's operands are 16-byte LDS reads, and k_rtz3_finish
undoes the permutation when it sums the slabs.  This test transcribes both sides' index arithmetic (lane by lane, with
the fragment conventions of v_mfma_f32_16x16x4_f32 from hmx_device.h) and checks that, for every shape family the
launcher instantiates, the round trip delivers  Y[k][pc] = sum_cells R[cell][k] Z[cell][pc]  and
S[blk][k] = sum_{cells in blk} R[cell][k]  -- a slip in either map shows up here before it costs GPU minutes."""
import numpy as np
import pytest


def mfma16(a_lane, b_lane, acc):
    """v_mfma_f32_16x16x4_f32: a_lane[l] = A[i = l & 15][k = l >> 4], b_lane[l] = B[k = l >> 4][j = l & 15],
    acc[l][r] = D[row = 4 (l >> 4) + r][col = l & 15]."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a_lane[l]
        B[l >> 4, l & 15] = b_lane[l]
    D = A @ B
    for l in range(64):
        for r in range(4):
            acc[l, r] += D[4 * (l >> 4) + r, l & 15]


def kernel_tile(Rt, Zt, blk, MT, KS, NTB, acc):
    """One 16-cell tile through k_rtz3's fragment construction (Rt: 16 x Kp, Zt: 16 x 4KS, blk: 16 block ids)."""
    NT, DP = 4 + NTB, 4 * KS
    H, REM = MT // 4, MT % 4
    Kp = Rt.shape[1]
    flatR = np.concatenate([Rt.ravel(), Zt.ravel()])        # a read past a row's end continues into what follows it
    for ks in range(4):
        afr = np.zeros((MT, 64))
        bfr = np.zeros((NT, 64))
        for lane in range(64):
            c16, q = lane & 15, lane >> 4
            cell = 4 * q + ks
            base = cell * Kp
            for h in range(H):
                for j in range(4):
                    afr[4 * h + j, lane] = flatR[base + 64 * h + 4 * c16 + j]
            for j in range(REM):
                afr[4 * H + j, lane] = flatR[base + 64 * H + REM * c16 + j]
            z = Zt[cell, 4 * min(c16, KS - 1): 4 * min(c16, KS - 1) + 4]
            bid = int(blk[4 * q + ks])
            for nt in range(4):
                bfr[nt, lane] = z[nt] if c16 < KS else (1.0 if bid == 4 * c16 + nt - DP else 0.0)
            for e in range(NTB):
                bfr[4 + e, lane] = 1.0 if bid == (64 - DP) + 16 * e + c16 else 0.0
        for mt in range(MT):
            for nt in range(NT):
                mfma16(afr[mt], bfr[nt], acc[mt][nt])


def finish(slab, MT, KS, NTB, K, d, nblk):
    """k_rtz3_finish's read-out of one slab [mt][nt][lane][r]."""
    NT, DP = 4 + NTB, 4 * KS
    Hq, rem = MT // 4, MT % 4

    def col_pc(pc):
        return 16 * (pc & 3) + (pc >> 2)

    def col_blk(j):
        if j < 64 - DP:
            c = DP + j
            return 16 * (c & 3) + (c >> 2)
        x = j - (64 - DP)
        return 16 * (4 + x // 16) + (x & 15)
    Y = np.zeros((K, d))
    S = np.zeros((nblk, K))
    for k in range(K):
        if k < 64 * Hq:
            mt, m = 4 * (k // 64) + (k & 3), (k & 63) >> 2
        else:
            x = k - 64 * Hq
            m, mt = x // rem, 4 * Hq + x % rem

        def val(v):
            nt, n = v >> 4, v & 15
            return slab[(mt * NT + nt) * 256 + (16 * (m >> 2) + n) * 4 + (m & 3)]
        for pc in range(d):
            Y[k, pc] = val(col_pc(pc))
        for b in range(nblk):
            S[b, k] = val(col_blk(b))
    return Y, S


@pytest.mark.parametrize("K,d,nblk", [(100, 50, 20), (30, 50, 20), (100, 30, 20), (64, 64, 20), (17, 33, 8), (112, 52, 44),
                                      (5, 3, 1), (48, 64, 32), (100, 50, 1)])
def test_rtz3_round_trip(K, d, nblk):
    rng = np.random.default_rng(K * 1000 + d)
    Kp, MT = (K + 3) & ~3, (K + 15) // 16
    dp = 32 if d <= 32 else 52 if d <= 52 else 64
    KS = dp // 4
    NTB = max(0, (nblk - (64 - dp) + 15) // 16)
    assert NTB <= 2
    NT = 4 + NTB
    n_tiles = 3
    R = rng.random((16 * n_tiles, Kp))
    R[:, K:] = 0.0
    Z = rng.normal(size=(16 * n_tiles, dp))
    Z[:, d:] = 0.0
    blk = rng.integers(0, nblk, size=16 * n_tiles)
    blk[5] = 255                                             # a padding position: no block
    acc = [[np.zeros((64, 4)) for _ in range(NT)] for _ in range(MT)]
    for t in range(n_tiles):
        kernel_tile(R[16 * t:16 * t + 16], Z[16 * t:16 * t + 16], blk[16 * t:16 * t + 16], MT, KS, NTB, acc)
    slab = np.zeros(MT * NT * 256)
    for mt in range(MT):
        for nt in range(NT):
            for r in range(4):
                for lane in range(64):
                    slab[(mt * NT + nt) * 256 + lane * 4 + r] = acc[mt][nt][lane, r]
    Y, S = finish(slab, MT, KS, NTB, K, d, nblk)
    np.testing.assert_allclose(Y, R[:, :K].T @ Z[:, :d], rtol=1e-12, atol=1e-12)
    S_ref = np.zeros((nblk, K))
    for i in range(16 * n_tiles):
        if blk[i] < nblk:
            S_ref[blk[i]] += R[i, :K]
    np.testing.assert_allclose(S, S_ref, rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------------------------------
# k_rtz3c (round 5's first cut was k_rtz3b: same maps, four waves): the same pass on the bf16 matrix pipe.  A k-step of v_mfma_f32_16x16x32_bf16 is a PAIR of the wave's tiles;
# lane (c16, q) supplies cells 8 (q & 1) .. + 8 of tile q >> 1 (hmx_device.h: A[i = l & 15][k = 8 (l >> 4) .. + 8]).
# The split into three bf16 terms is value-preserving (tests/test_split_gemm.py), so the replay multiplies the values.
# ------------------------------------------------------------------------------------------
def mfma32(a_lane, b_lane, acc):
    """v_mfma_f32_16x16x32_bf16: a_lane[l][j] = A[i = l & 15][k = 8 (l >> 4) + j], b_lane likewise B[k][j = l & 15]."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4): 8 * (l >> 4) + 8] = a_lane[l]
        B[8 * (l >> 4): 8 * (l >> 4) + 8, l & 15] = b_lane[l]
    D = A @ B
    for l in range(64):
        for r in range(4):
            acc[l, r] += D[4 * (l >> 4) + r, l & 15]


def kernel_pair_b(tiles, MT, KS, NTB, Kp, acc):
    """One pair of tiles through k_rtz3c's fragment construction.  tiles: two (Rt 16 x Kp, Zt 16 x 4KS, blk 16, n_live);
    n_live = 0 stands for the missing second tile of an odd count (its buffer holds the first tile again, zeroed, ids 255)."""
    NT, DP = 4 + NTB, 4 * KS
    H, REM = MT // 4, MT % 4
    bufs = []
    for Rt, Zt, blk, n_live in tiles:
        Rt, Zt, blk = Rt.copy(), Zt.copy(), blk.copy()
        Rt[n_live:] = 0.0
        Zt[n_live:] = 0.0
        if n_live == 0:
            blk[:] = 255
        bufs.append((Rt, Zt, blk))
    afr = np.zeros((MT, 64, 8))
    bfr = np.zeros((NT, 64, 8))
    for lane in range(64):
        c16, q = lane & 15, lane >> 4
        Rt, Zt, blk = bufs[q >> 1]
        for j in range(8):
            cell = 8 * (q & 1) + j
            bid = int(blk[cell])
            zc = 4 * min(c16, KS - 1)
            for nt in range(4):
                bfr[nt, lane, j] = Zt[cell, zc + nt] if c16 < KS else (1.0 if bid == 4 * c16 + nt - DP else 0.0)
            if NTB > 0:
                bfr[4, lane, j] = 1.0 if bid == (64 - DP) + c16 else 0.0
            for h in range(H):
                for jj in range(4):
                    afr[4 * h + jj, lane, j] = Rt[cell, 64 * h + 4 * c16 + jj]
            for jj in range(REM):
                col = 64 * H + REM * c16 + jj
                afr[4 * H + jj, lane, j] = Rt[cell, col] if col < Kp else 0.0
    for mt in range(MT):
        for nt in range(NT):
            mfma32(afr[mt], bfr[nt], acc[mt][nt])


@pytest.mark.parametrize("K,d,nblk,n_tiles,last_live", [
    (100, 50, 20, 4, 16), (100, 50, 20, 3, 9),      # C3: round pass (one one-hot tile), odd tile count + a group's ragged end
    (100, 50, 1, 2, 16), (30, 30, 1, 5, 3),         # ridge / centroid-only passes: no block tile at 52 / 32-float rows
    (64, 64, 1, 2, 16), (112, 64, 16, 3, 16),       # 64-float rows: the one-hot tile carries every block column
    (30, 30, 20, 2, 16), (17, 33, 8, 1, 5), (5, 3, 1, 1, 16), (48, 17, 28, 4, 1)])
def test_rtz3b_pair_round_trip(K, d, nblk, n_tiles, last_live):
    """Every (row length, cluster-tile count, block-tile count) family launch_rtz3 hands to k_rtz3c, with the pair
    bookkeeping: odd tile counts (the spare buffer requests the first tile again and counts for nothing) and a last tile
    that runs past the group's end."""
    rng = np.random.default_rng(K * 977 + d * 31 + nblk)
    Kp, MT = (K + 3) & ~3, (K + 15) // 16
    dp = 32 if d <= 32 else 52 if d <= 52 else 64
    KS = dp // 4
    NTB = max(0, (nblk - (64 - dp) + 15) // 16)
    assert NTB <= 1                                          # rtz3b_ok
    NT = 4 + NTB
    R = rng.random((16 * n_tiles, Kp))
    R[:, K:] = 0.0
    Z = rng.normal(size=(16 * n_tiles, dp))
    Z[:, d:] = 0.0
    blk = rng.integers(0, nblk, size=16 * n_tiles)
    live = [16] * (n_tiles - 1) + [last_live]
    acc = [[np.zeros((64, 4)) for _ in range(NT)] for _ in range(MT)]
    for i in range((n_tiles + 1) // 2):
        pair = []
        for u in range(2):
            t = 2 * i + u
            if t < n_tiles:
                pair.append((R[16 * t:16 * t + 16], Z[16 * t:16 * t + 16], blk[16 * t:16 * t + 16], live[t]))
            else:
                t = 2 * i                                    # request_pair: the first tile again, into the spare buffer
                pair.append((R[16 * t:16 * t + 16], Z[16 * t:16 * t + 16], blk[16 * t:16 * t + 16], 0))
        kernel_pair_b(pair, MT, KS, NTB, Kp, acc)
    slab = np.zeros(MT * NT * 256)
    for mt in range(MT):
        for nt in range(NT):
            for r in range(4):
                for lane in range(64):
                    slab[(mt * NT + nt) * 256 + lane * 4 + r] = acc[mt][nt][lane, r]
    Y, S = finish(slab, MT, KS, NTB, K, d, nblk)
    keep = np.concatenate([np.arange(16) < n for n in live])
    Rl, Zl = R[keep], Z[keep]
    np.testing.assert_allclose(Y, Rl[:, :K].T @ Zl[:, :d], rtol=1e-12, atol=1e-12)
    S_ref = np.zeros((nblk, K))
    for i in np.flatnonzero(keep):
        S_ref[blk[i]] += R[i, :K]
    np.testing.assert_allclose(S, S_ref, rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------------------------------
# k_rtzw2b: the wide streaming pass on the bf16 matrix pipe.  Four waves split the MT x NT output tiles 2 x 2 (row half
# x column half); a k-step is a pair of the task's tiles with k_rtz3c's k-slot map; plain column tiles (PC tile nt holds
# columns 16 nt .., the padding columns d .. dp-1 of the last PC tile carry the first one-hot block columns, whole extra
# tiles the rest); rows past the group's end and the missing tile of an odd count are ZEROS in LDS (store_tile).
# ------------------------------------------------------------------------------------------
def finish_wide(slab, MT, NT, K, d, DP, nblk):
    """k_rtz3_finish's read-out of one slab [mt][nt][lane][r] in its `wide` mode."""
    Hq, rem = MT // 4, MT % 4
    spare = DP - d
    Y = np.zeros((K, d))
    S = np.zeros((nblk, K))
    for k in range(K):
        if k < 64 * Hq:
            mt, m = 4 * (k // 64) + (k & 3), (k & 63) >> 2
        else:
            x = k - 64 * Hq
            m, mt = x // rem, 4 * Hq + x % rem

        def val(v):
            nt, n = v >> 4, v & 15
            return slab[(mt * NT + nt) * 256 + (16 * (m >> 2) + n) * 4 + (m & 3)]
        for pc in range(d):
            Y[k, pc] = val(pc)
        for b in range(nblk):
            S[b, k] = val(d + b if b < spare else DP + (b - spare))
    return Y, S


def rtzw2b_wave_pair(bufs, wv, MT, NTH, Kp, DP, d, NT, acc):
    """One pair of tile buffers through one wave of k_rtzw2b (bufs: two (Rt 16 x Kp, Zt 16 x DP, ids 16) images as
    store_tile leaves them).  acc: dict (mt, nt) -> 64 x 4."""
    MTA = (MT + 1) // 2
    H, REM = MT // 4, MT % 4
    NTP = DP // 16
    rh, ch = wv >> 1, wv & 1
    LO, HI = rh * MTA, (MT if rh else MTA)
    nt_lo = ch * NTH
    bplanes = []
    for u in range(NTH):
        nt = nt_lo + u
        b = np.zeros((64, 8))
        for lane in range(64):
            c16, q = lane & 15, lane >> 4
            Rt, Zt, ids = bufs[q >> 1]
            for j in range(8):
                cell = 8 * (q & 1) + j
                x = Zt[cell, 16 * min(nt, NTP - 1) + c16] if nt < NTP else 0.0
                if NTP - 1 <= nt < NT:
                    blk_col = 16 * nt + c16 - d
                    x += 1.0 if int(ids[cell]) == blk_col else 0.0
                b[lane, j] = x
        bplanes.append(b)
    for mt in range(LO, HI):
        a = np.zeros((64, 8))
        for lane in range(64):
            c16, q = lane & 15, lane >> 4
            Rt, Zt, ids = bufs[q >> 1]
            flat = np.concatenate([Rt.ravel(), np.full(256 * MT - Rt.size, np.nan)])   # the R segment is padded to MT KB: a 16-byte read
            for j in range(8):                                                          # past a row's end stays inside it (rows of clusters >= Kp: never read out)
                cell = 8 * (q & 1) + j
                if mt < 4 * H:
                    a[lane, j] = flat[cell * Kp + 64 * (mt >> 2) + 4 * c16 + (mt & 3)]
                else:
                    col = 64 * H + REM * c16 + (mt - 4 * H)
                    a[lane, j] = flat[cell * Kp + col] if col < Kp else 0.0
        for u in range(NTH):
            nt = nt_lo + u
            if nt < NT:                                      # (the kernel multiplies zeros there and does not store them)
                mfma32(a, bplanes[u], acc.setdefault((mt, nt), np.zeros((64, 4))))


@pytest.mark.parametrize("K,d,nblk,n_tiles,last_live", [(200, 200, 20, 3, 7), (200, 200, 1, 2, 16), (130, 100, 20, 4, 16),
                                                         (208, 120, 40, 1, 3), (177, 193, 20, 2, 16)])
def test_rtzw2b_round_trip(K, d, nblk, n_tiles, last_live):
    """Shapes launch_rtzw hands to k_rtzw2b (K > 112, seven to fourteen column tiles): configs[4] with its 20 update blocks
    (8 one-hot columns in the row padding + one extra tile) and as the ridge / k-means pass (one block column), odd tile
    counts, a ragged last tile, a remainder row tile, a one-hot tile per column half."""
    rng = np.random.default_rng(K * 131 + d * 7 + nblk)
    Kp, MT = (K + 3) & ~3, (K + 15) // 16
    DP = (d + 15) & ~15
    NT = DP // 16 + max(0, (nblk - (DP - d) + 15) // 16)
    NTH = (NT + 1) // 2
    assert 8 <= MT <= 13 and 4 <= NTH <= 7                   # rtzw2b_ok
    R = rng.random((16 * n_tiles, Kp))
    R[:, K:] = 0.0
    Z = rng.normal(size=(16 * n_tiles, DP))
    Z[:, d:] = 0.0
    blk = rng.integers(0, nblk, size=16 * n_tiles)
    live = [16] * (n_tiles - 1) + [last_live]

    def image(ti):                                           # store_tile: the tile as it sits in LDS
        tc = min(ti, n_tiles - 1)
        n_live = live[tc] if ti < n_tiles else 0
        Rt, Zt = R[16 * tc:16 * tc + 16].copy(), Z[16 * tc:16 * tc + 16].copy()
        Rt[n_live:] = 0.0
        Zt[n_live:] = 0.0
        return Rt, Zt, blk[16 * tc:16 * tc + 16]
    acc = {}
    for i in range((n_tiles + 1) // 2):
        bufs = [image(2 * i), image(2 * i + 1)]
        for wv in range(4):
            rtzw2b_wave_pair(bufs, wv, MT, NTH, Kp, DP, d, NT, acc)
    slab = np.zeros(MT * NT * 256)
    for (mt, nt), a in acc.items():
        for lane in range(64):
            for r in range(4):
                slab[(mt * NT + nt) * 256 + lane * 4 + r] = a[lane, r]
    assert len(acc) == MT * NT                               # every output tile has exactly one owner
    Y, S = finish_wide(slab, MT, NT, K, d, DP, nblk)
    keep = np.concatenate([np.arange(16) < n for n in live])
    np.testing.assert_allclose(Y, R[keep][:, :K].T @ Z[keep][:, :d], rtol=1e-12, atol=1e-12)
    S_ref = np.zeros((nblk, K))
    for i in np.flatnonzero(keep):
        S_ref[blk[i]] += R[i, :K]
    np.testing.assert_allclose(S, S_ref, rtol=1e-12, atol=1e-12)

