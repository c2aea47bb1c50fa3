// hmx_sweep.hip -- k_sweep: one whole update_R sweep (harmony.py:464-513) in ONE persistent launch:
//
//   * the distance GEMM against the LDS-resident centroids (harmony.py:447; never materialised),
//   * the diversity-penalised reassignment of every block (:466-468, 495-503) with the block's own
//     old assignments taken out of O / E first (:491-492) and its new ones put back (:506-507),
//   * the objective sums (:399, :402; the cross-entropy term :405-411 is closed by workgroup 0),
//   * the removal sums of block b+1 (column sums of the OLD R rows of its cells), formed while block b
//     is being finished: the rows travel global -> LDS by DMA, no separate pass and no registers.
//
// Per sweep the N-sized traffic is: Z_cos row (read), R row (read old, write new), 20 bytes of list
// entries per cell.
//
// Shape of the launch: one workgroup of EIGHT waves per CU (two per SIMD, 256 registers per lane), at
// most #CUs workgroups so that every workgroup is resident.  A wave owns a contiguous chunk of every
// block's tile list and runs ONE instruction stream per tile s:
//       finish of tile s (VALU: exp, penalty, renormalise, R rows out)   interleaved piece by piece with
//       the 91 MFMAs of the distance product of tile s+1                  and the loads of tiles s+2, s+3.
// The order inside a stream is pinned (TIE_*, below); the two waves of a SIMD run their streams out of
// step, so one wave's VALU pieces and waits fill the issue slots next to the other's MFMAs.
// (Measured alternatives, DESIGN.md §3: the same stream with the next round's centroid numerators
// R_new^T.Z_cos fused in needs 112 more accumulators = 415 registers = one wave per SIMD, which then eats
// every LDS / memory latency itself: 664-880 us per sweep at C3 against 561 us for sweep + separate pass.)
//
// Orientation of the product (hmx_device.h conventions; lane = (c, q), c = lane & 15, q = lane >> 4):
//   D[cell][cluster]: A = Z_cos tile (i = cell), B = Y (j = cluster)
//   -> lane (c, q) holds clusters 16 mt + c of cells 4q + r, r = 0..3 (register r of accumulator mt):
//      the softmax sums run over the 16 lanes of a DPP row, a cell's 4 x MT entries share the lane's
//      table values, the block sums are 3 additions + one 4-lane reduction per cluster tile.
//
// The 20 blocks are sequential through a G x K table only (O).  Hand-off p (p = 0 .. nblk) carries
//   D[p] = (new sums of block p-1) - (old sums of block p)      per (group, cluster),
// accumulated by every workgroup into one of four fp64 slot tables with agent-scope atomics, then one
// arrival counter; O for block b = O_start + D[0] + .. + D[b].  Cells sharded over ranks: an extra gateway
// workgroup pushes the rank's D[p] into every rank's peer box.
//
// Waits are bounded (spin_limit); on a time-out a.error is set, the workgroup keeps going with whatever
// it has (every later wait gives up at once) and the host repeats the round through the per-block path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <utility>

#include "hmx_internal.h"
#include "hmx_device.h"

#ifdef HMX_SWEEP_PROF   /* timing experiments only: per-workgroup phase stamps (s_memtime) of wave 0 */
#define SSTAMP(slot) do { if (tid == 0 && a.prof) a.prof[((size_t)wg * a.nblk + b) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(slot) do { } while (0)
#endif
#ifndef HMX_SABL
#define HMX_SABL 0   /* timing experiments only (results become wrong): 1 no old-R loads, 2 no R stores, 4 no MFMAs */
#endif
#ifndef SWEEP_WAVES
#define SWEEP_WAVES 8
#endif
#define SWEEP_THREADS (64 * SWEEP_WAVES)

namespace {

// compile-time loop: f(std::integral_constant<int, 0>{}), f(<1>), ...  The index is a constant expression inside f,
// so register arrays are indexed statically whatever the optimiser's unrolling heuristics say (a rolled loop would
// send them to scratch) and `if constexpr` selects what a step contains.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Order pins without instructions: an empty asm that "rewrites" two registers makes everything that produces either
// of them precede, and everything that consumes either of them follow -- the only ordering every compiler pass honours
// for register-only instructions (MFMAs are pure to the optimiser; scheduling barriers act too late to keep them
// apart).  TIE_ACC names an MFMA result without waiting for it: the asm is empty, nothing reads the accumulator.
#ifdef HMX_NO_TIES   /* timing experiments only */
#define TIE_VV(x, y) do { } while (0)
#define TIE_ACC(acc, x) do { } while (0)
#else
#define TIE_VV(x, y) asm("" : "+v"(x), "+v"(y))
#define TIE_ACC(acc, x) asm("" : "+v"(acc), "+v"(x))
#endif

template <int MT, int KS>
struct SweepShape {
    static constexpr int K16 = 16 * MT;
    static constexpr int NF = KS / 4, NT = KS % 4;
    static constexpr int LDZ = 4 * KS;                           // row of Z_cos in HBM (floats): 32, 52 or 64 (the engine's row lengths)
    static constexpr int LDY = (KS & 1) ? 4 * KS : 4 * KS + 4;   // row of Y in LDS: (LDY / 4) odd -> conflict-free 16-byte reads
    static constexpr int GSTEPS = NF + (NT > 0 ? 1 : 0);         // k-step groups of the distance product (16 columns each)
};

template <int KS>
struct ZRegs {   // a cell's Z_cos row as this lane's A fragments: 16-byte pieces q, q+4, .. + one float per tail k-step
    static constexpr int NF = KS / 4, NT = KS % 4;
    f32x4 zp[NF > 0 ? NF : 1];
    float zt[NT > 0 ? NT : 1];
};
template <int MT>
struct ArgTile {
    f32x4 arg[MT];   // [mt][r]: -dist / sigma of cluster 16 mt + c, cell 4q + r
    i32x4 cell4;     // list entries 4q .. 4q+3 (the cells whose assignments this lane holds)
    int grp;
};
template <int MT>
struct FinishTables { float pw[MT], lp[MT], sg[MT]; };
// working registers of one cell's finish when it is cut into pieces
template <int MT>
struct FinishRow {
    float arg[MT], th[MT], tl[MT], ex[MT], tt[MT], ts[MT];
    float e1, us, a1, a2, a3, den, scl, lden, rc, pin;
    int row_id;
};
// the register piece k of a cell's finish starts from / ends in (what the order pins hold on to): 5 pieces per cluster
// tile, then the closing pieces (row sums, scale, objective terms, row address, one store per cluster tile)
template <int MT>
__device__ __forceinline__ float& fin_in(FinishRow<MT>& W, int k) {
    if (k < 5 * MT) {
        const int mt = k / 5, p = k % 5;
        return p == 0 ? W.pin : p == 1 ? W.th[mt] : p == 2 ? W.ex[mt] : p == 3 ? W.tt[mt] : W.ts[mt];
    }
    const int u = k - 5 * MT;
    return u < 4 ? W.e1 : u == 4 ? W.us : u == 5 ? W.rc : u == 6 ? W.scl : u == 7 ? W.lden : u == 8 ? W.scl : W.tt[u - 9 < MT ? u - 9 : 0];
}
template <int MT>
__device__ __forceinline__ float& fin_out(FinishRow<MT>& W, int k) {
    if (k < 5 * MT) {
        const int mt = k / 5, p = k % 5;
        return p == 0 ? W.tl[mt] : p == 1 ? W.ex[mt] : p == 2 ? W.ts[mt] : p == 3 ? W.a2 : W.a3;
    }
    const int u = k - 5 * MT;
    return u < 4 ? W.us : u == 4 ? W.rc : u == 5 ? W.lden : u == 6 ? W.a1 : u == 7 ? W.a2 : u == 8 ? W.scl : W.tt[u - 9 < MT ? u - 9 : 0];
}

}  // namespace

template <int MT, int KS>
__global__ __launch_bounds__(SWEEP_THREADS) void k_sweep(SweepArgs a) {
    using S = SweepShape<MT, KS>;
    constexpr int K16 = S::K16, NF = S::NF, NT = S::NT, LDZ = S::LDZ, LDY = S::LDY, GSTEPS = S::GSTEPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int GK = a.G * K16;
    float* Ys = reinterpret_cast<float*>(smem);                           // K16 x LDY
    float* remtile = Ys + (size_t)K16 * LDY;                              // waves x (4 MT) x 64: landing zone of the old R rows (LDS-DMA)
    float* sig = remtile + (size_t)SWEEP_WAVES * 4 * MT * 64;             // K16
    float* nis = sig + K16;                                               // K16: -1/sigma (-60 for pads)
    float* rpT = nis + K16;                                               // G x K16
    float* lrpT = rpT + GK;                                               // G x K16
    float* rpc = lrpT + GK;                                               // B x K16 (several batch variables only)
    double* Ocur = reinterpret_cast<double*>(rpc + (size_t)K16 * a.B);    // G x K16 (all offsets so far are even)
    double* Sd = Ocur + GK;                                               // G x K16: this workgroup's share of the next hand-off
    double* Tm = Sd + GK;                                                 // K16
    double* objw = Tm + K16;                                              // waves x 2
    float* prb = reinterpret_cast<float*>(objw + 2 * SWEEP_WAVES);        // B
    float* tht = prb + a.B;                                               // B
    int* gcol = reinterpret_cast<int*>(tht + a.B);                        // G x V
    int* gst = gcol + a.G * a.V;                                          // G + 1: first cell of every group (cells are stored group-sorted)
    int* chunk_j0 = gst + a.G + 1;                                        // waves x (nblk + 1): first tile of the wave's chunk
    int* chunk_n = chunk_j0 + SWEEP_WAVES * (a.nblk + 1);                 // waves x (nblk + 1): tiles in the chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const bool multi = a.n_ranks > 1;
    const int wg = blockIdx.x, nwg = multi ? gridDim.x - 1 : gridDim.x;   // compute workgroups
    unsigned long long* my_flags = multi ? reinterpret_cast<unsigned long long*>(a.my_box + box_flags(a.n_ranks, GK)) : nullptr;

    if (multi && wg == nwg) {
        // ---- gateway workgroup: once every local workgroup has added its share of hand-off p, write the
        //      rank's total into every rank's box (xGMI peer writes), then raise this rank's flag there.
        bool gfail = false;
        for (int p = 0; p <= a.nblk; ++p) {
            if (wv == 0) {
                const unsigned want = (unsigned)(p + 1) * (unsigned)nwg;
                unsigned spins = 0;
                while (!gfail && ld_agent(a.counter) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) gfail = true;
                }
            }
            __syncthreads();
            for (int i = tid; i < GK; i += SWEEP_THREADS) {
                const double* sn = a.D_slots + (size_t)p * HMX_ROUND_SLOTS * GK + i;
                double v = 0.0;
#pragma unroll
                for (int s = 0; s < HMX_ROUND_SLOTS; ++s) v += ld_agent(sn + (size_t)s * GK);
                for (int r = 0; r < a.n_ranks; ++r) st_sys(a.peer_box[r] + box_data(a.n_ranks, GK, p & 1, a.rank) + i, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < a.n_ranks)
                st_sys(reinterpret_cast<unsigned long long*>(a.peer_box[tid] + box_flags(a.n_ranks, GK)) + (size_t)(p & 1) * a.n_ranks + a.rank,
                       a.epoch + (unsigned long long)p + 1ull);
        }
        if (gfail && tid == 0) {
            atomicExch(a.error, 1u);
            atomicAdd(&a.obj[0], __builtin_nan(""));   // every rank sees the failure in the all-reduce of the objective sums
        }
        return;
    }

    // ---- tables into LDS ----------------------------------------------------------------------------
    for (int i = tid; i < a.B; i += SWEEP_THREADS) {
        prb[i] = a.Pr_b[i];
        tht[i] = a.theta[i];
    }
    for (int i = tid; i < a.G * a.V; i += SWEEP_THREADS) gcol[i] = a.group_cols[i];
    for (int i = tid; i <= a.G; i += SWEEP_THREADS) gst[i] = a.gstart[i];
    for (int i = tid; i < K16; i += SWEEP_THREADS) {
        const float sgm = (i < a.K) ? a.sigma[i] : 0.f;
        sig[i] = sgm;
        nis[i] = (i < a.K) ? -1.0f / sgm : -60.f;   // pads: Y row 0 -> dist 2 -> arg -120 -> exp == 0
    }
    for (int i = tid; i < GK; i += SWEEP_THREADS) {
        Ocur[i] = a.O_start[i];
        Sd[i] = 0.0;
    }
    for (int i = tid; i < K16 * KS; i += SWEEP_THREADS) {
        const int row = i / KS, c4 = i - row * KS;
        st4(Ys + (size_t)row * LDY + 4 * c4, ld4(a.Y + (size_t)row * a.ldy + 4 * c4));
    }
    {
        // this wave's chunk of every block's tile list: balanced, contiguous (wave w of nw: the first
        // `extra` waves take one tile more); entry nblk is an empty sentinel
        const int w = wg + nwg * wv, nw = nwg * SWEEP_WAVES;
        for (int b = lane; b <= a.nblk; b += 64) {
            int j0 = 0, n = 0;
            if (b < a.nblk) {
                const int t0 = a.blk_start[b], ntl = a.blk_start[b + 1] - t0;
                const int base = ntl / nw, extra = ntl - base * nw;
                n = base + (w < extra ? 1 : 0);
                j0 = t0 + w * base + min(w, extra);
            }
            chunk_j0[wv * (a.nblk + 1) + b] = j0;
            chunk_n[wv * (a.nblk + 1) + b] = n;
        }
    }
    __syncthreads();

    const int* my_j0 = chunk_j0 + wv * (a.nblk + 1);
    const int* my_n = chunk_n + wv * (a.nblk + 1);
    auto chunk_first = [&](int b) { return __builtin_amdgcn_readfirstlane(my_j0[b]); };
    auto chunk_count = [&](int b) { return __builtin_amdgcn_readfirstlane(my_n[b]); };

    // group of a tile = group of its first list entry (never padding), found in the table of first cells: no memory
    // access that could miss (reading the tile_group list: one vector load + vmcnt(0) + readfirstlane per tile, the wait
    // also draining the row stores just issued -- 5k cycles -- or one scalar load, ~1k cycles on a miss)
    const int gst_lane = (lane + 1 < a.G) ? gst[lane + 1] : 0x7fffffff;   // lane l: first cell of group l + 1 (G <= 64)
    auto group_of = [&](int first_cell) {
        const int cellu = __builtin_amdgcn_readfirstlane(first_cell);
        return (int)__popcll(__ballot(cellu >= gst_lane));
    };

    // ---- building blocks ------------------------------------------------------------------------------
    auto cell_of = [&](int j) { return j >= 0 ? a.cells[(size_t)j * 16 + c] : -1; };
    // rows of dead list entries (padding, or no tile at all) read cell 0: finite values that meet a zero factor later
    auto row_of = [&](int cell16) { return a.Zcos + (size_t)(cell16 >= 0 ? cell16 : 0) * LDZ; };
    auto issue_rows = [&](const float* zr, ZRegs<KS>& Z) {
#pragma unroll
        for (int j = 0; j < NF; ++j) Z.zp[j] = ld4(zr + 16 * j + 4 * q);
#pragma unroll
        for (int s = 0; s < NT; ++s) Z.zt[s] = zr[16 * NF + 4 * s + q];
    };
    // dist = 2 (1 - Y.Z) (:447), arg = -dist / sigma (:466)
    auto gemm_end = [&](ArgTile<MT>& T) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float ni2 = 2.f * nis[16 * mt + c];
            const f32x4 one = (f32x4){1.f, 1.f, 1.f, 1.f};
            T.arg[mt] = (one - T.arg[mt]) * ni2;
        }
    };
    // per-(group, cluster) sums a wave carries over the tiles of one group run; lane (c, q) holds the
    // share of its four cells.  flush: reduce over q, one fp64 LDS atomic per (group, cluster).
    auto flush_run = [&](float (&sum)[MT], int grp, double sign) {
        float v[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v[mt] = wave_sum_q_swap(sum[mt]);
            sum[mt] = 0.f;
        }
        if (q == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) atomicAdd(&Sd[(size_t)grp * K16 + 16 * mt + c], sign * (double)v[mt]);
        }
    };
    float gsum[MT], rsum[MT];
    int gsum_grp = -1, rsum_grp = -1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) gsum[mt] = rsum[mt] = 0.f;
    double km_acc = 0.0, ent_acc = 0.0;

    // ---- table-dependent half of a tile, cut into pieces of about three VALU instructions ----------------
    // With ex = exp(arg), e1 = sum ex (:467-468), t = ex * ratio^theta (:500) and U = sum t:
    //   R = (t / e1) / max(U / e1, 1e-8) = t / max(U, 1e-8 e1)                         (:501-503)
    //   sum R dist          = -scl sum t sigma arg,                  scl = 1 / max(U, 1e-8 e1)   (:399, dist = -arg sigma)
    //   sum sigma R log R   =  scl [sum t sigma arg + sum t sigma log(ratio^theta) - log(max(U, 1e-8 e1)) sum t sigma]   (:402)
    // ONE exp per entry.  Cell r of the lane: 5 pieces per cluster tile, then FIN_TAIL closing pieces.
    const int trash_row = (int)a.n_cells;                   // K16 floats behind the last row
    const bool last_col_ok = 16 * (MT - 1) + c < a.Kp;      // cluster tiles before the last lie below K <= Kp entirely
    constexpr int FIN_TAIL = 9 + MT, NPART = 5 * MT, NVALU = NPART + FIN_TAIL;
    auto finish_begin = [&](const ArgTile<MT>& T, FinishTables<MT>& F) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            F.pw[mt] = rpT[(size_t)T.grp * K16 + 16 * mt + c];
            F.lp[mt] = lrpT[(size_t)T.grp * K16 + 16 * mt + c];
            F.sg[mt] = sig[16 * mt + c];
        }
    };
    auto fin_part = [&](const ArgTile<MT>& T, const FinishTables<MT>& F, FinishRow<MT>& W, int r, int mt, int p) __attribute__((always_inline)) {
        const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
        if (p == 0) {
            W.arg[mt] = T.arg[mt][r];
            W.th[mt] = W.arg[mt] * L2E_HI;                                      // fast_exp_finite, spread over two pieces
            W.tl[mt] = fmaf(W.arg[mt], L2E_LO, fmaf(W.arg[mt], L2E_HI, -W.th[mt]));
        } else if (p == 1) {
            const float pp = __builtin_amdgcn_exp2f(W.th[mt]);
            W.ex[mt] = fmaf(pp, W.tl[mt] * 0.693147182464599609375f, pp);      // :467
        } else if (p == 2) {
            W.e1 += W.ex[mt];
            W.tt[mt] = W.ex[mt] * F.pw[mt];                                     // :500 (the 1/e1 of :468 cancels, see above)
            W.ts[mt] = W.tt[mt] * F.sg[mt];
        } else if (p == 3) {
            W.us += W.tt[mt];
            W.a1 = fmaf(W.ts[mt], W.arg[mt], W.a1);
            W.a2 = fmaf(W.ts[mt], F.lp[mt], W.a2);
        } else {
            W.a3 += W.ts[mt];
        }
    };
    auto fin_tail = [&](const ArgTile<MT>& T, FinishRow<MT>& W, float (&bsum)[MT], int r, int u) __attribute__((always_inline)) {
        if (u == 0) { W.e1 += DPP_F(W.e1, 0xB1); W.us += DPP_F(W.us, 0xB1); }            // row16_sum, one step per piece
        else if (u == 1) { W.e1 += DPP_F(W.e1, 0x4E); W.us += DPP_F(W.us, 0x4E); }
        else if (u == 2) { W.e1 += DPP_F(W.e1, 0x141); W.us += DPP_F(W.us, 0x141); }
        else if (u == 3) { W.e1 += DPP_F(W.e1, 0x140); W.us += DPP_F(W.us, 0x140); }    // column sums of :468 and :501
        else if (u == 4) {
            W.den = fmaxf(W.us, 1e-8f * W.e1);                                  // e1 * max(sum R_new, 1e-8)   (:501-502)
            W.rc = __builtin_amdgcn_rcpf(W.den);                                // v_rcp_f32 (1 ulp)
        } else if (u == 5) {
            W.scl = (T.cell4[r] >= 0) ? W.rc : 0.f;                             // dead entries (list padding) contribute exact zeros
            W.lden = __builtin_amdgcn_logf(W.den) * 0.693147182464599609375f;   // den >= 1e-26: a normal number
        } else if (u == 6) {
            km_acc -= (double)(W.scl * W.a1);
        } else if (u == 7) {
            ent_acc += (double)(W.scl * (W.a1 + W.a2 - W.lden * W.a3));
        } else if (u == 8) {
            W.row_id = (T.cell4[r] >= 0) ? T.cell4[r] : trash_row;
        } else {
            // unconditional stores: dead entries and the columns beyond Kp of the last cluster tile go to a trash row
            // behind R (a predicated store would cut the stream into exec-masked blocks)
            const int mt = u - 9;
            const float rv = W.tt[mt] * W.scl;                                  // :503
            W.tt[mt] = rv;
            bsum[mt] += rv;                                                     // :506-507
            const int rid = (mt < MT - 1 || last_col_ok) ? W.row_id : trash_row;
            if (!(HMX_SABL & 2)) a.R[(size_t)rid * a.Kp + 16 * mt + c] = rv;    // :509 (in place: the order is a list, not a copy)
        }
    };

    // ---- old assignments of a tile of the NEXT block: column sums only (:491-492) ---------------------------
    struct RemIds { int cell16; i32x4 cell4; bool valid; };
    auto rem_ids = [&](int j, bool valid) {
        RemIds t;
        t.valid = valid;
        t.cell16 = valid ? a.cells[(size_t)j * 16 + c] : -1;
        t.cell4 = valid ? *reinterpret_cast<const i32x4*>(a.cells + (size_t)j * 16 + 4 * q) : (i32x4){-1, -1, -1, -1};
        return t;
    };
    // The tile's 16 rows travel global -> LDS directly (global_load_lds: no destination registers -- held in VGPRs across
    // a tile they are spilled one by one, each spill waiting for its load).  Instruction k moves cluster tile k (16 columns
    // = 64 bytes) of all 16 rows: lane l brings the 16-byte piece l % 4 of row l / 4 and it lands at byte 16 (64 k + l) of
    // the wave's landing zone, i.e. the zone is laid out [cluster tile][row][16 columns].  Pieces beyond Kp re-read the
    // row's first piece (masked out when the sums are formed).  No instruction offset: it would move the LDS address too.
    float* rem_wave = remtile + (size_t)wv * 4 * MT * 64;
    const bool last_piece_ok = 16 * (MT - 1) + 4 * (lane & 3) < a.Kp;
    // lane l's source address: piece l % 4 of the row of list entry l / 4 (formed when the ids have landed for certain,
    // behind a full wait, so that no wait for them stands between a tile's row stores and the next loads)
    auto rem_source = [&](const RemIds& t) {
        const int cell = __shfl(t.cell16, lane >> 2, 64);
        return a.R + (size_t)(cell >= 0 ? cell : 0) * a.Kp + 4 * (lane & 3);
    };
    auto rem_issue = [&](const float* src) {
        if (HMX_SABL & 1) return;
        static_for<MT>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const float* sk = (k < MT - 1 || last_piece_ok) ? src + 16 * k : src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sk,
                                             (__attribute__((address_space(3))) void*)(rem_wave + 256 * k), 16, 0, 0);
        });
    };
    auto rem_consume = [&](const RemIds& t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows have landed (nothing else orders an LDS read behind an LDS-DMA)
        const int grp = group_of(t.cell4[0]);
        if (grp != rsum_grp) {
            if (rsum_grp >= 0) flush_run(rsum, rsum_grp, -1.0);
            rsum_grp = grp;
        }
        // all 4 MT reads first, unconditionally (written as a select they become 28 exec-masked read + wait pairs)
        float v[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[mt][r] = (HMX_SABL & 1) ? 0.f : rem_wave[(16 * mt + 4 * q + r) * 16 + c];
                asm("" : "+v"(v[mt][r]));
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool col_ok = 16 * mt + c < a.K;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += (col_ok && t.cell4[r] >= 0) ? v[mt][r] : 0.f;
            rsum[mt] += s;
        }
    };

    // ---- hand-off machinery ----------------------------------------------------------------------------
    bool failed = false;
    unsigned ws_n = 0, ws_sum = 0, ws_max = 0;   // this workgroup's grid-wide waits (wave 0)
    // add this workgroup's share of hand-off p to the slot tables, then arrive (all waves call this)
    auto publish = [&](int p) {
        if (gsum_grp >= 0) flush_run(gsum, gsum_grp, 1.0);
        if (rsum_grp >= 0) flush_run(rsum, rsum_grp, -1.0);
        gsum_grp = rsum_grp = -1;
        wg_barrier_lds();
        double* dst = a.D_slots + ((size_t)p * HMX_ROUND_SLOTS + (wg % HMX_ROUND_SLOTS)) * GK;
        for (int i = tid; i < GK; i += SWEEP_THREADS) {
            const double v = Sd[i];
            if (v != 0.0) atomicAdd(dst + i, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sums are performed
        wg_barrier_lds();
        if (tid == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // wait until hand-off p is complete on every workgroup (and rank)
    auto wait_handoff = [&](int p) {
        if (wv == 0) {
            unsigned spins = 0;
            if (a.spin_limit == 0) failed = true;   // test knob: give up without looking
            if (!multi) {
                const unsigned want = (unsigned)(p + 1) * (unsigned)nwg;
                while (!failed && ld_agent(a.counter) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (failed || ++spins > a.spin_limit) { failed = true; break; }
                }
            } else {
                const unsigned long long want = a.epoch + (unsigned long long)p + 1ull;
                const unsigned long long* fl = my_flags + (size_t)(p & 1) * a.n_ranks;
                while (true) {
                    const bool ok = lane >= a.n_ranks || ld_sys(fl + lane) >= want;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (failed || ++spins > a.spin_limit) { failed = true; break; }
                }
            }
            ws_n += 1; ws_sum += spins; ws_max = max(ws_max, spins);
        }
        wg_barrier_lds();
    };
    // O with hand-off p added (:491-492, 506-507); clears this workgroup's share for the next one
    auto apply_handoff = [&](int p) {
        for (int i = tid; i < GK; i += SWEEP_THREADS) {
            double add[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) add[s] = 0.0;
            if (!multi) {
                const double* sn = a.D_slots + (size_t)p * HMX_ROUND_SLOTS * GK + i;
#pragma unroll
                for (int s = 0; s < HMX_ROUND_SLOTS; ++s) add[s] = ld_agent(sn + (size_t)s * GK);   // independent loads
            } else {
                const double* bx = a.my_box + box_data(a.n_ranks, GK, p & 1, 0) + i;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < a.n_ranks) add[s] = ld_sys(bx + (size_t)s * GK);
            }
            double o = Ocur[i];
#pragma unroll
            for (int s = 0; s < 8; ++s) o += add[s];
            Ocur[i] = o;
            Sd[i] = 0.0;
        }
        wg_barrier_lds();
    };
    // diversity table of the block: ratio ** theta per (group, cluster) and its log  (:495-499)
    auto build_table = [&]() {
        if (a.V == 1) {
            // one batch variable: group g is batch g
            for (int i = tid; i < GK; i += SWEEP_THREADS) {
                const int g = i / K16, k = i - g * K16;
                double t = 0.0;
                for (int gg = 0; gg < a.G; ++gg) t += Ocur[(size_t)gg * K16 + k];
                const float O = (float)Ocur[i];
                const float E = (float)t * prb[g];                          // :491 (E kept as cluster mass x Pr_b)
                const float oe = fmaxf(O + E, 1e-8f);                       // :495-496
                const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);      // :497-498
                const float rp = pow_unit(ratio, tht[g]);                   // :499
                rpT[i] = rp;
                lrpT[i] = __builtin_amdgcn_logf(rp) * 0.693147182464599609375f;   // v_log_f32 (log2, 1 ulp) * ln 2
            }
        } else {
            for (int k = tid; k < K16; k += SWEEP_THREADS) {
                double t = 0.0;
                for (int g = 0; g < a.G; ++g) t += Ocur[(size_t)g * K16 + k];
                Tm[k] = t;
            }
            wg_barrier_lds();
            for (int i = tid; i < K16 * a.B; i += SWEEP_THREADS) {
                const int bb = i / K16, k = i - bb * K16;
                double Ob = 0.0;
                for (int g = 0; g < a.G; ++g) {
                    bool has = false;
                    for (int v = 0; v < a.V; ++v) has |= gcol[g * a.V + v] == bb;
                    if (has) Ob += Ocur[(size_t)g * K16 + k];
                }
                const float O = (float)Ob;
                const float E = (float)Tm[k] * prb[bb];
                const float oe = fmaxf(O + E, 1e-8f);
                const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);
                rpc[i] = pow_unit(ratio, tht[bb]);
            }
            wg_barrier_lds();
            for (int i = tid; i < GK; i += SWEEP_THREADS) {
                const int g = i / K16, k = i - g * K16;
                float s = 0.f;
                for (int v = 0; v < a.V; ++v) s += rpc[(size_t)gcol[g * a.V + v] * K16 + k];
                rpT[i] = s;                                                 // (ratio_pow @ Phi) for the cells of group g
                lrpT[i] = __builtin_amdgcn_logf(s) * 0.693147182464599609375f;
            }
        }
        wg_barrier_lds();
    };

    // ---- the wave's tile stream: an iterator running ahead of the tile being finished ---------------------
    int la_b = 0, la_i = 0, la_n = chunk_count(0), la_j0 = chunk_first(0);
    auto la_next = [&]() {   // next tile index of this wave's stream (blocks in which it has no tile are skipped), or -1
        while (la_b < a.nblk && la_i >= la_n) {
            ++la_b;
            la_i = 0;
            la_n = chunk_count(la_b);   // entry nblk: 0
            la_j0 = chunk_first(la_b);
        }
        const int j = la_b < a.nblk ? la_j0 + la_i : -1;
        ++la_i;
        return j;
    };

    // ---- one tile: finish of tile s || distance product of tile s+1 || loads of what comes after ------------------
    // MFMA slots: group g (16 columns of the rows, or the tail k-steps) x cluster tile x k-step, cluster-tile-major, so that
    // a centroid fragment (16 bytes of Y per cluster tile and group) serves consecutive MFMAs and is read from LDS one
    // fragment ahead; an MFMA sits in every other slot of a group (two MFMAs on one accumulator are >= 2 slots apart),
    // every slot carries one piece of the finish of cell g.  `side(S)`: hook for the loads, behind the MFMA of slot S.
    constexpr int SLOTS = (2 * 4 * MT > NVALU) ? 2 * 4 * MT : NVALU;   // slots of a group
    auto tile_stream = [&](const ArgTile<MT>& Tf, const FinishTables<MT>& F, float (&bsum)[MT], const ZRegs<KS>& Zr, ArgTile<MT>& Tn,
                           auto&& side) __attribute__((always_inline)) {
        const int ya_base = c * LDY + 4 * q, yat_base = c * LDY + q;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Tn.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 ya_cur = (f32x4){0.f, 0.f, 0.f, 0.f}, ya_nxt;
        if constexpr (NF > 0) ya_cur = ld4(Ys + ya_base);
        else
            static_for<NT>([&](auto sc) __attribute__((always_inline)) { ya_cur[decltype(sc)::value] = Ys[yat_base + 4 * decltype(sc)::value]; });
        ya_nxt = ya_cur;
        static_for<4>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            constexpr bool g_full = g < NF, g_tail = (g == NF && NT > 0);
            constexpr int per_mt = g_full ? 4 : (NT > 0 ? NT : 1);
            FinishRow<MT> W;
            W.e1 = W.us = W.a1 = W.a2 = W.a3 = 0.f;
            W.pin = 0.f;
            static_for<SLOTS>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int idx = k >> 1;
                constexpr int g_mt = idx / per_mt, g_i = idx % per_mt;
                constexpr bool is_gemm = (k & 1) == 0 && (g_full || g_tail) && g_mt < MT && !(HMX_SABL & 4);
                if constexpr (is_gemm) {
                    if constexpr (g_full) Tn.arg[g_mt] = MFMA16(Zr.zp[g_full ? g : 0][g_i], ya_cur[g_i], Tn.arg[g_mt]);
                    else Tn.arg[g_mt] = MFMA16(Zr.zt[g_i < NT ? g_i : 0], ya_cur[g_i], Tn.arg[g_mt]);
                    if constexpr (g_i == 0) {
                        // after the first use of a fragment, read the one after it (next cluster tile, or tile 0 of the next group)
                        constexpr int n_mt = g_mt + 1 < MT ? g_mt + 1 : 0;
                        constexpr int n_g = g_mt + 1 < MT ? g : g + 1;
                        if constexpr (n_g < GSTEPS) {
                            // (the pin holds on to a COPY of the lane's base address; the rest is the instruction's immediate)
                            int bs = n_g < NF ? ya_base : yat_base;
                            TIE_ACC(Tn.arg[g_mt], bs);
                            if constexpr (n_g < NF) ya_nxt = ld4(Ys + bs + (16 * n_mt * LDY + 16 * n_g));
                            else
                                static_for<NT>([&](auto sc) __attribute__((always_inline)) {
                                    constexpr int sx = decltype(sc)::value;
                                    ya_nxt[sx] = Ys[bs + (16 * n_mt * LDY + 16 * NF + 4 * sx)];
                                });
                        }
                    }
                    side(std::integral_constant<int, g * SLOTS + k>{}, Tn.arg[g_mt]);
                }
                // one piece of the finish of cell g, pinned behind this slot's MFMA; the next MFMA is pinned behind the piece
                if constexpr (k < NVALU) {
                    if constexpr (is_gemm) TIE_ACC(Tn.arg[g_mt], fin_in<MT>(W, k));
                    if constexpr (k < NPART) fin_part(Tf, F, W, g, k / 5, k % 5);
                    else fin_tail(Tf, W, bsum, g, k - NPART);
                    if constexpr (is_gemm && g_i == per_mt - 1) TIE_VV(fin_out<MT>(W, k), ya_nxt);
                    else if constexpr (is_gemm) TIE_VV(fin_out<MT>(W, k), ya_cur);
                }
                if constexpr (is_gemm && g_i == per_mt - 1) ya_cur = ya_nxt;
            });
        });
        gemm_end(Tn);
    };

    // ---- prologue: hand-off 0 = minus the old sums of block 0; first tiles start travelling ---------------
    // Tile s is being finished (cur); its successor's rows (s+1) are in flight in zq with its list entries in nx_*; of
    // tile s+2 the row address every lane starts from (zrow_2) is known.
    const int j_0 = la_next(), j_1 = la_next();
    int j_2 = la_next();
    ZRegs<KS> z_a, zq;
    issue_rows(row_of(cell_of(j_0)), z_a);
    issue_rows(row_of(cell_of(j_1)), zq);
    const float* zrow_2 = row_of(cell_of(j_2));
    ArgTile<MT> cur;
    cur.cell4 = j_0 >= 0 ? *reinterpret_cast<const i32x4*>(a.cells + (size_t)j_0 * 16 + 4 * q) : (i32x4){-1, -1, -1, -1};
    cur.grp = group_of(cur.cell4[0]);
    i32x4 nx_cell4 = j_1 >= 0 ? *reinterpret_cast<const i32x4*>(a.cells + (size_t)j_1 * 16 + 4 * q) : (i32x4){-1, -1, -1, -1};
    int nx_grp = group_of(nx_cell4[0]);
    // old assignments of a chunk without anything to hide behind (prologue; chunks that outlast the previous block's tiles)
    auto rem_chunk = [&](RemIds t, int j0, int u, int n) {
        for (; u < n; ++u) {
            rem_issue(rem_source(t));
            const RemIds tn = rem_ids(j0 + u + 1, u + 1 < n);
            rem_consume(t);
            t = tn;
        }
    };
    rem_chunk(rem_ids(chunk_first(0), chunk_count(0) > 0), chunk_first(0), 0, chunk_count(0));
    // the removal sums of block p are gathered between hand-off p-1 and hand-off p, one tile per tile finished
    int rn = chunk_count(1), rj0 = chunk_first(1), ru = 0;
    RemIds rid = rem_ids(rj0, rn > 0);
    const float* rsrc = rem_source(rid);
    publish(0);
    {   // distance product of the wave's first tile, plain (once per sweep)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) cur.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 ya = ld4(Ys + (size_t)(16 * mt + c) * LDY + 16 * j + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) cur.arg[mt] = MFMA16(z_a.zp[j][i], ya[i], cur.arg[mt]);
            }
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) cur.arg[mt] = MFMA16(z_a.zt[s], Ys[(size_t)(16 * mt + c) * LDY + 16 * NF + 4 * s + q], cur.arg[mt]);
        gemm_end(cur);
    }

    for (int b = 0; b < a.nblk; ++b) {
        const int n = chunk_count(b);
        SSTAMP(0);
        wait_handoff(b);
        SSTAMP(1);
        apply_handoff(b);
        build_table();
        SSTAMP(2);
        auto end_block = [&]() {
            // what is left of the next block's chunk, the ids its successor starts with, then hand-off b+1
            rem_chunk(rid, rj0, ru, rn);
            const int n2 = chunk_count(min(b + 2, a.nblk)), j2 = chunk_first(min(b + 2, a.nblk));
            const RemIds rid2 = rem_ids(j2, n2 > 0);
            SSTAMP(4);
            publish(b + 1);
            SSTAMP(5);
            rn = n2; rj0 = j2; ru = 0; rid = rid2;
            rsrc = rem_source(rid);
        };
        for (int i = 0; i < n; ++i) {
            ZRegs<KS> zn;
            const bool n2_valid = j_2 >= 0;
            i32x4 n2_cell4 = (i32x4){-1, -1, -1, -1};
            int j_3 = -1, c16_3 = -1;
            const bool rem_now = ru < rn;
            RemIds rid_next = rid;
            ArgTile<MT> nxt;
            FinishTables<MT> F;
            finish_begin(cur, F);
            if (cur.grp != gsum_grp) {
                if (gsum_grp >= 0) flush_run(gsum, gsum_grp, 1.0);
                gsum_grp = cur.grp;
            }
            // the loads of what comes after ride behind the first MFMAs (nothing of this tile depends on them)
            auto side = [&](auto Sc, f32x4& acc) __attribute__((always_inline)) {
                constexpr int S = decltype(Sc)::value;
                if constexpr (S == 0) {
                    TIE_ACC(acc, zrow_2);
                    issue_rows(zrow_2, zn);
                } else if constexpr (S == 2) {
                    const int* p4 = a.cells + (size_t)(n2_valid ? j_2 : 0) * 16 + 4 * q;
                    TIE_ACC(acc, p4);
                    const i32x4 v4 = *reinterpret_cast<const i32x4*>(p4);
                    n2_cell4 = n2_valid ? v4 : (i32x4){-1, -1, -1, -1};
                } else if constexpr (S == 4) {
                    TIE_ACC(acc, rsrc);
                    if (rem_now) rem_issue(rsrc);
                } else if constexpr (S == 6) {
                    j_3 = la_next();
                    const int* p1 = a.cells + (size_t)(j_3 >= 0 ? j_3 : 0) * 16 + c;
                    TIE_ACC(acc, p1);
                    const int v1 = *p1;
                    c16_3 = j_3 >= 0 ? v1 : -1;
                } else if constexpr (S == 8) {
                    if (rem_now) rid_next = rem_ids(rj0 + ru + 1, ru + 1 < rn);
                }
            };
#ifndef HMX_PLAIN_STREAM
            tile_stream(cur, F, gsum, zq, nxt, side);
#else   /* timing experiments only: 707 us per sweep at C3 against 658 for the pinned stream */
            // plain order, scheduled by the compiler: loads, finish of tile s (28 independent entries: plenty of ILP),
            // distance product of tile s+1; the partner wave of the SIMD fills the gaps
            side(std::integral_constant<int, 0>{}, cur.arg[0]);
            side(std::integral_constant<int, 2>{}, cur.arg[0]);
            side(std::integral_constant<int, 4>{}, cur.arg[0]);
            side(std::integral_constant<int, 6>{}, cur.arg[0]);
            side(std::integral_constant<int, 8>{}, cur.arg[0]);
            static_for<4>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                FinishRow<MT> W;
                W.e1 = W.us = W.a1 = W.a2 = W.a3 = 0.f;
                W.pin = 0.f;
                static_for<NVALU>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < NPART) fin_part(cur, F, W, g, k / 5, k % 5);
                    else fin_tail(cur, W, gsum, g, k - NPART);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) nxt.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            static_for<GSTEPS>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                f32x4 ya[MT];
                if constexpr (g < NF) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) ya[mt] = ld4(Ys + (size_t)(16 * mt + c) * LDY + 16 * g + 4 * q);
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) nxt.arg[mt] = MFMA16(zq.zp[g < NF ? g : 0][i2], ya[mt][i2], nxt.arg[mt]);
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < NT; ++s2) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) ya[mt][s2] = Ys[(size_t)(16 * mt + c) * LDY + 16 * NF + 4 * s2 + q];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) nxt.arg[mt] = MFMA16(zq.zt[s2], ya[mt][s2], nxt.arg[mt]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            gemm_end(nxt);
#endif
            if (i == 0) SSTAMP(3);
            if (rem_now) {
                rem_consume(rid);
                rid = rid_next;
                ++ru;
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // everything requested in the stream has landed: the addresses the next tile starts from are formed here,
            // not behind its row stores
            if (i == n - 1) end_block();
            else rsrc = rem_source(rid);
            zrow_2 = row_of(c16_3);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) cur.arg[mt] = nxt.arg[mt];
            cur.cell4 = nx_cell4;
            cur.grp = nx_grp;
            nx_cell4 = n2_cell4;
            nx_grp = group_of(n2_cell4[0]);
            zq = zn;
            j_2 = j_3;
        }
        if (n == 0) end_block();   // no tile of this block here: the hand-offs still take place
        SSTAMP(6);
    }

    if (tid == 0 && a.wait_stats) {
        atomicAdd(a.wait_stats, (unsigned long long)ws_n);
        atomicAdd(a.wait_stats + 1, (unsigned long long)ws_sum);
        atomicMax(a.wait_stats + 2, (unsigned long long)ws_max);
    }
    // ---- objective partial sums (:399, :402) ----------------------------------------------------
    km_acc = wave_sum_all(km_acc);
    ent_acc = wave_sum_all(ent_acc);
    if (lane == 0) {
        objw[2 * wv] = km_acc;
        objw[2 * wv + 1] = ent_acc;
    }
    __syncthreads();
    if (tid < 2) {
        double v = 0.0;
        for (int w = 0; w < SWEEP_WAVES; ++w) v += objw[2 * w + tid];
        if (v != 0.0) atomicAdd(&a.obj[2 * (wg & (HMX_OBJ_SLOTS - 1)) + tid], v);
    }
    if (wg != 0) {
        if (failed && tid == 0) {
            atomicExch(a.error, 1u);
            atomicAdd(&a.obj[0], __builtin_nan(""));
        }
        return;
    }

    // ---- workgroup 0 closes the sweep: O, cluster mass, cross-entropy term (:405-411) -----------
    wait_handoff(a.nblk);
    apply_handoff(a.nblk);
    if (failed && tid == 0) {
        atomicExch(a.error, 1u);
        atomicAdd(&a.obj[0], __builtin_nan(""));
    }
    for (int i = tid; i < GK; i += SWEEP_THREADS) a.O_out[i] = Ocur[i];
    double part = 0.0;
    for (int i = tid; i < K16 * a.B; i += SWEEP_THREADS) {
        const int bb = i / K16, k = i - bb * K16;
        double Tk = 0.0, Ob = 0.0;
        for (int g = 0; g < a.G; ++g) {
            const double o = Ocur[(size_t)g * K16 + k];
            Tk += o;
            bool has = false;
            for (int v = 0; v < a.V; ++v) has |= gcol[g * a.V + v] == bb;
            if (has) Ob += o;
        }
        if (bb == 0) a.T_out[k] = Tk;
        const float O = (float)Ob;
        const float Oc = fmaxf(O, 1e-8f);                               // :407
        const float Ec = fmaxf((float)Tk * prb[bb], 1e-8f);             // :408
        const float tl = tht[bb] * logf((Oc + Ec) / Ec);                // :409-410
        part += (double)(sig[k] * O * tl);
    }
    part = wave_sum_all(part);
    __syncthreads();
    if (lane == 0) objw[wv] = part;
    __syncthreads();
    if (tid == 0) {
        double v = 0.0;
        for (int w = 0; w < SWEEP_WAVES; ++w) v += objw[w];
        atomicAdd(&a.obj[2 * HMX_OBJ_SLOTS], v);
    }
}

// ---- host side ---------------------------------------------------------------------------------------
static int sweep_ks(int d) { return d <= 32 ? 8 : d <= 52 ? 13 : d <= 64 ? 16 : 0; }
int sweep_row_floats(int d) { return 4 * sweep_ks(d); }   // 32 / 52 / 64: the engine's row lengths for d <= 32 / 52 / 64
int sweep_waves() { return SWEEP_WAVES; }

size_t sweep_lds_bytes(int K16, int d, int G, int B, int V, int nblk) {
    const int ks = sweep_ks(d);
    if (!ks) return (size_t)1 << 30;
    const int ldy = (ks & 1) ? 4 * ks : 4 * ks + 4;
    const size_t GK = (size_t)G * K16;
    return ((size_t)K16 * ldy + (size_t)SWEEP_WAVES * 4 * (K16 / 16) * 64 + 2 * (size_t)K16 + 2 * GK + (size_t)K16 * B) * 4 +
           (2 * GK + K16 + 2 * SWEEP_WAVES) * 8 + (2 * (size_t)B + (size_t)G * V + G + 1 + 2 * (size_t)SWEEP_WAVES * (nblk + 1) + 16) * 4;
}

template <int MT, int KS>
static void launch_sweep_t(const SweepArgs& a, int wgs, size_t sm, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<MT, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_sweep<MT, KS>), dim3(wgs), dim3(SWEEP_THREADS), sm, s, a);
}
template <int KS>
static void launch_sweep_ks(const SweepArgs& a, int mt, int wgs, size_t sm, hipStream_t s) {
#ifdef HMX_SWEEP_ONLY_C3   /* analysis builds: only the <7, 13> instantiation */
    if (KS == 13 && mt == 7) launch_sweep_t<7, 13>(a, wgs, sm, s);
#else
    switch (mt) {
        case 1: launch_sweep_t<1, KS>(a, wgs, sm, s); break;
        case 2: launch_sweep_t<2, KS>(a, wgs, sm, s); break;
        case 3: launch_sweep_t<3, KS>(a, wgs, sm, s); break;
        case 4: launch_sweep_t<4, KS>(a, wgs, sm, s); break;
        case 5: launch_sweep_t<5, KS>(a, wgs, sm, s); break;
        case 6: launch_sweep_t<6, KS>(a, wgs, sm, s); break;
        default: launch_sweep_t<7, KS>(a, wgs, sm, s); break;
    }
#endif
}

// d = PCs; the rows of Z_cos must be sweep_row_floats(d) floats long.  -1: shape not covered.
int launch_sweep(const SweepArgs& a, int mt, int d, int wgs, hipStream_t s) {
    const size_t sm = sweep_lds_bytes(16 * mt, d, a.G, a.B, a.V, a.nblk);
    if (mt < 1 || mt > 7 || sm > 156 * 1024 || a.ldz != sweep_row_floats(d) || a.G > 64) return -1;
    switch (sweep_ks(d)) {
        case 8: launch_sweep_ks<8>(a, mt, wgs, sm, s); break;
        case 13: launch_sweep_ks<13>(a, mt, wgs, sm, s); break;
        case 16: launch_sweep_ks<16>(a, mt, wgs, sm, s); break;
        default: return -1;
    }
    return 0;
}
