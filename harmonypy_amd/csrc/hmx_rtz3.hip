// hmx_rtz3.hip -- the R^T.Z pass as a STREAMING kernel (gfx950): centroid numerators Z_cos R^T (harmony.py:443), the removal
// sums of every update block R_blk Phi_blk^T (harmony.py:491-492) and the ridge statistics Phi_Rk Z_orig^T with the exact
// O (harmony.py:550, 556-563), all from ONE pass over R and Z in storage order.
//
// Why another kernel (k_rtz2, hmx_kernels.hip, walks the round's block-major list): gathering rows of 400 / 208 bytes in a
// random order fetches 1.38x the algorithmic bytes (whole 128-byte lines), and its four waves share every tile through
// LDS with a workgroup barrier per tile and ~150 bookkeeping instructions beside 28 MFMAs.  Here:
//   * cells are read in storage order (group-sorted, rows contiguous): every byte fetched is used;
//   * a WAVE is the unit: it owns whole 16-cell tiles, brings them global -> LDS with `global_load_lds_dwordx4`
//     (1 KB per instruction, no destination registers), double buffered in its private LDS region, and multiplies a tile
//     with all MT x NT output tiles in its own accumulators -- no workgroup barrier, no ids, no shuffles in the loop;
//   * block membership does not decide the ORDER any more; it rides in the product: the B operand is the Z tile
//     (PC columns) extended by one-hot columns "cell is in block j", so the MFMAs deliver sum_{cells in block j} R[cell][k]
//     next to the centroid numerators.  16 bytes of block ids per tile (static tile order, written once per round by
//     k_tile_blocks from the round's lists) travel with the tile.  With all ids 0 column 0 is the plain column sum of R:
//     the exact O of the ridge step.
//
// MFMA fragment conventions (hmx_device.h); index maps chosen so that a lane's operands are 16-byte LDS reads:
//   k index (ks, q)      <-> cell 4q + ks of the tile             (a lane's four block ids are one dword)
//   A row m = c16, tile mt: cluster 64h + 4 c16 + j for mt = 4h + j in the full groups of four tiles,
//                           cluster 64H + r c16 + j for the r = MT % 4 remaining tiles (H = MT / 4)
//   B col n = c16, tile nt < 4: column 4 c16 + nt of [Z row (dp floats) | one-hot of blocks 0 .. 63-dp]
//                  tile nt >= 4: one-hot of block (64 - dp) + 16 (nt - 4) + c16
// k_rtz3_finish undoes the maps when it sums the per-task slabs in fp64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "hmx_internal.h"
#include "hmx_device.h"

#ifndef HMX_RTZ3_Z_NT
#define HMX_RTZ3_Z_NT 1  /* k_rtz3c: Z rows requested non-temporal like the R rows (0: experiment of round 6 -- Z_cos, 200 MB at C3, read by both
                            kernels of a round, left cacheable so that it might stay in the 256 MB memory-side cache: profiles/r06_ab_zcos_cacheable.txt) */
#endif
#ifndef HMX_RTZ3_NT
#define HMX_RTZ3_NT 1    /* the rows are read once per pass: non-temporal requests (micro-benchmark of this very pattern: 6.2 -> 6.9 TB/s) */
#endif
// (defined in FRONT of dma16: behind it -- where it stood until round 4 -- `#if HMX_RTZ3_NT` read an undefined macro, i.e. 0, and
// every request of the streaming kernels went out without the hint)
namespace {

// one 1 KB piece global -> LDS: lane l brings the 16 bytes at `base` + `voff` (= 16 l) to LDS byte address `zone` + 16 l.
// `base` is wave-uniform and travels in scalar registers (the SADDR form of the instruction): the stream's addresses cost
// no vector registers, however far ahead the compiler forms them.  Inline assembly on purpose: the
// builtin makes the compiler drain vmcnt before any later LDS read.
template <bool NT = true>
__device__ __forceinline__ void dma16(const void* base_, unsigned voff, unsigned zone_) {
    const unsigned zone = __builtin_amdgcn_readfirstlane(zone_);   // (likewise: under pressure a uniform LDS address may sit in a vector register)
    // (under register pressure the compiler may park a wave-uniform pointer in vector registers: say it again that it is
    // uniform -- folded away when the value already sits in scalar registers.  The builtin returns a signed int: through
    // `unsigned`, or a low half >= 2^31 sign-extends into the high one.)
    const unsigned long long bits = (unsigned long long)base_;
    const void* base = (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bits >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bits));
#if HMX_RTZ3_NT
    if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
    else
#endif
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void*)p);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

}  // namespace

#define RTZ3_WAVES 4
#ifndef HMX_RTZ3_ABL
#define HMX_RTZ3_ABL 0   /* timing experiments only (results become wrong): 1 no MFMAs (the stream alone), 2 no requests after the prologue (the arithmetic alone) */
#endif

#ifdef HMX_RTZ3_PROF   /* timing experiments only: s_memtime stamps per wave */
#define R3STAMP(k) do { if (lane == 0 && a.prof) a.prof[((size_t)blockIdx.x * RTZ3_WAVES + wv) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define R3ACC(k, v) do { if (lane == 0 && a.prof) a.prof[((size_t)blockIdx.x * RTZ3_WAVES + wv) * 8 + (k)] += (v); } while (0)
#define R3STAMP8(k) do { if (lane == 0 && a.prof) a.prof[((size_t)blockIdx.x * 8 + wv) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define R3ACC8(k, v) do { if (lane == 0 && a.prof) a.prof[((size_t)blockIdx.x * 8 + wv) * 8 + (k)] += (v); } while (0)
#else
#define R3STAMP8(k) do { } while (0)
#define R3ACC8(k, v) do { } while (0)
#define R3STAMP(k) do { } while (0)
#define R3ACC(k, v) do { } while (0)
#endif
template <int MT, int KS, int NTB>
__global__ __launch_bounds__(64 * RTZ3_WAVES, 2) void k_rtz3(Rtz3Args a) {
    if (a.frozen && *a.frozen) return;   // an earlier sweep of this cluster() call timed out: the state stays as that round left it
    constexpr int NT = 4 + NTB, DP = 4 * KS;
    constexpr int NR = MT, NZ = (KS + 3) / 4, NI = NR + NZ + 1;   // VMEM operations per tile: R pieces, Z pieces, block ids
    constexpr int H = MT / 4, REM = MT % 4;
    static_assert(KS <= 16 && NI < 60, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int Kp = a.Kp;
    const int buf_floats = 16 * (Kp + DP) + 4;                     // R tile | Z tile | 16 block-id bytes
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int lane16 = 16 * lane;
    const int task = blockIdx.x;
    const int t0 = a.task_t0[task], t1 = a.task_t1[task];
    const int c_first = a.task_c0[task], c_end = a.task_cend[task];
    // tiles of wave w: t0 + w + stride i (< t1).  stride = RTZ3_WAVES: a task is a contiguous run of tiles; stride = RTZ3_WAVES x
    // (tasks of the group): the group's tasks take neighbouring quads of tiles and sweep the group's rows together, like a
    // grid-stride copy (DRAM pages and TLB entries are shared by the whole chip instead of one stream per workgroup)
    const int stride = __builtin_amdgcn_readfirstlane(a.task_stride[task]);

    R3STAMP(0);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_mine = (t1 - t0 - wv + stride - 1) / stride;
    // The three streams of this wave (R rows, Z rows, block ids) as wave-uniform byte pointers to the NEXT tile to request,
    // advanced by a constant per tile: scalar registers and scalar adds only (a 64-bit product per request would be formed
    // on the vector unit and kept live across the MFMAs).
    auto uniform64 = [](unsigned long long v) {
        // (the builtin returns a signed int: through `unsigned`, or a low half >= 2^31 sign-extends into the high one)
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    };
    const int c_mine = c_first + 16 * wv;                           // first cell of this wave's first tile
    unsigned long long nr = uniform64((unsigned long long)(a.R + (size_t)c_mine * Kp));
    unsigned long long nz = uniform64((unsigned long long)(a.Z + (size_t)c_mine * DP));
    unsigned long long nb = uniform64((unsigned long long)(a.tile_blk + (size_t)16 * (t0 + wv)));
    const unsigned long long step_r = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(stride * 64 * Kp);   // 16 rows x 4 B per tile
    const unsigned long long step_z = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(stride * 64 * DP);
    const unsigned long long step_b = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(16 * stride);
    const unsigned zone0 = lds_addr(lds + (size_t)(2 * wv) * buf_floats);
    const unsigned buf_bytes = __builtin_amdgcn_readfirstlane((unsigned)buf_floats * 4u);
    const unsigned r_bytes = __builtin_amdgcn_readfirstlane(64u * (unsigned)Kp);
    // piece p of the NI requests that bring the next tile into buffer `par`: R pieces, Z pieces, the block ids
    auto issue_piece = [&](int par, int p) {
        const unsigned zb = zone0 + (par ? buf_bytes : 0u);
        if (p < NR) {
            if (p + 1 < NR || 1024 * p + lane16 < 64 * Kp) dma16((const void*)(nr + 1024ull * p), lane16, zb + 1024u * p);
        } else if (p < NR + NZ) {
            const int it = p - NR;
            if (1024 * (it + 1) <= 64 * DP || 1024 * it + lane16 < 64 * DP) dma16((const void*)(nz + 1024ull * it), lane16, zb + r_bytes + 1024u * it);
        } else {
            if (lane == 0) dma16((const void*)nb, lane16, zb + r_bytes + 64u * DP);
        }
    };
    auto advance = [&]() { nr += step_r; nz += step_z; nb += step_b; };
    if (n_mine > 0) {
#pragma unroll
        for (int p = 0; p < NI; ++p) issue_piece(0, p);
        advance();
    }
    if (n_mine > 1) {
#pragma unroll
        for (int p = 0; p < NI; ++p) issue_piece(1, p);
        advance();
    }

  if constexpr (NT <= 5) {
    // The wave's work is one stream of k-steps (4 per tile).  While the MT x NT MFMAs of a k-step are in the matrix pipe the
    // fragments of the NEXT k-step are read from LDS into the other register set -- across tile boundaries too: under the
    // last k-step of tile i the wave waits for tile i+1 (requested a whole tile earlier), reads its first fragments, and
    // then hands tile i's buffer (whose last read has long returned) to tile i+2, the NI requests riding between the MFMAs.
    // So a wave never has an MFMA-free phase after its prologue.  Why it matters: the two waves of a SIMD start together and
    // do identical work; with "read all fragments, then 4 x MT x NT MFMAs" they reached their read phases together and the
    // matrix pipe idled for both (measured: 66 % busy, 172-190 us per pass at C3 with or without the memory traffic).
    constexpr int NMF = MT * NT;                                    // MFMAs of a k-step
    // last k-step of a tile: first the NI requests of tile i+2 (one every GAP MFMAs: tile i's buffer is free, its last
    // fragments are in registers), THEN the wait for tile i+1 -- two tiles travel while the wave multiplies.  (The other
    // order -- wait first, request after -- left one tile in flight per wave and the stream at 3.7 TB/s.)
    constexpr int GAP = NMF >= 3 * NI ? 2 : 1;                      // MFMAs between two requests
    constexpr int OPEN = NI * GAP + 1 < NMF ? (NI * GAP + 1 > 2 * NMF / 3 ? NI * GAP + 1 : 2 * NMF / 3) : NMF;   // MFMAs issued before the wait
    float afr[2][MT];
    f32x4 zfr[2];
    unsigned bw = 0, bw_next = 0;
    auto read_frags = [&](const float* Rt, const float* Zt, int ks, int set) {
        const float* rr = Rt + (size_t)(4 * q + ks) * Kp;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x4 v = ld4(rr + 64 * h + 4 * c16);
#pragma unroll
            for (int j = 0; j < 4; ++j) afr[set][4 * h + j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < REM; ++j) afr[set][4 * H + j] = rr[64 * H + REM * c16 + j];
        zfr[set] = ld4(Zt + (size_t)(4 * q + ks) * DP + 4 * min(c16, KS - 1));
    };
    // tile i has landed (nothing younger is in flight at the points this is called): rows past the group's end hold other
    // cells (or the slack behind the array) and count for nothing; then its first fragments and its block ids
    auto open_tile = [&](int i, bool younger) {
        asm volatile("" ::: "memory");
#ifdef HMX_RTZ3_PROF
        const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
        if (younger) wait_vmcnt<NI>(); else wait_vmcnt<0>();        // (the requests of tile i+1 may travel on)
#ifdef HMX_RTZ3_PROF
        R3ACC(4, __builtin_amdgcn_s_memtime() - w0_);
#endif
        asm volatile("" ::: "memory");
        const int c0 = c_mine + 16 * stride * i;
        float* Rt = lds + (size_t)(2 * wv + (i & 1)) * buf_floats;
        float* Zt = Rt + 16 * Kp;
        const int n_live = min(16, c_end - c0);                     // wave-uniform; < 16 only in a group's last tile
        if (n_live < 16) {
            for (int j = n_live * Kp + lane; j < 16 * Kp; j += 64) Rt[j] = 0.f;
            for (int j = n_live * DP + lane; j < 16 * DP; j += 64) Zt[j] = 0.f;
        }
        bw_next = reinterpret_cast<const unsigned*>(Zt + 16 * DP)[q];   // block ids of cells 4q .. 4q+3
        read_frags(Rt, Zt, 0, 0);
    };
    if (n_mine > 0) {
        if (n_mine > 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();     // tile 0 (tile 1 may still travel)
        asm volatile("" ::: "memory");
        const int n_live = min(16, c_end - c_mine);
        float* Rt = lds + (size_t)(2 * wv) * buf_floats;
        float* Zt = Rt + 16 * Kp;
        if (n_live < 16) {
            for (int j = n_live * Kp + lane; j < 16 * Kp; j += 64) Rt[j] = 0.f;
            for (int j = n_live * DP + lane; j < 16 * DP; j += 64) Zt[j] = 0.f;
        }
        bw_next = reinterpret_cast<const unsigned*>(Zt + 16 * DP)[q];
        read_frags(Rt, Zt, 0, 0);
    }
    R3STAMP(1);
    for (int i = 0; i < n_mine; ++i) {
        float* Rt = lds + (size_t)(2 * wv + (i & 1)) * buf_floats;
        float* Zt = Rt + 16 * Kp;
        bw = bw_next;
        const bool next = i + 1 < n_mine;                           // wave-uniform
        const bool more = i + 2 < n_mine && !(HMX_RTZ3_ABL & 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int set = ks & 1;
            if (ks < 3) read_frags(Rt, Zt, ks + 1, set ^ 1);
            const int bid = (bw >> (8 * ks)) & 255;
            float bfr[NT];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bfr[nt] = (c16 < KS) ? zfr[set][nt] : ((bid == 4 * c16 + nt - DP) ? 1.f : 0.f);
#pragma unroll
            for (int e = 0; e < NTB; ++e) bfr[4 + e] = (bid == (64 - DP) + 16 * e + c16) ? 1.f : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#if HMX_RTZ3_ABL & 1
                    acc[mt][nt][0] += afr[set][mt] * bfr[nt];
#else
                    acc[mt][nt] = MFMA16(afr[set][mt], bfr[nt], acc[mt][nt]);
#endif
                    if (ks == 3) {
                        const int m = mt * NT + nt + 1;             // MFMAs of this k-step issued so far
                        if (m % GAP == 0 && m / GAP - 1 < NI && m < OPEN) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue_piece(i & 1, m / GAP - 1);   // (the reads of tile i's buffer returned before this k-step began)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (m == OPEN) {
                            __builtin_amdgcn_sched_barrier(0);
                            if ((OPEN - 1) / GAP < NI) {            // (few MFMAs per k-step: the remaining requests go out here)
#pragma unroll
                                for (int p = (OPEN - 1) / GAP; p < NI; ++p)
                                    if (more) issue_piece(i & 1, p);
                            }
                            if (next) open_tile(i + 1, more);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
        }
        if (more) advance();
    }

  } else {
    // Six column tiles (29 to 44 update blocks, or 17+ with 64-float rows): 168 accumulators leave no room for a second
    // fragment set.  Plain form: wait for the tile, read all its fragments, hand the buffer on, multiply; the partner wave
    // of the SIMD covers the read phase when it can.
    for (int i = 0; i < n_mine; ++i) {
        asm volatile("" ::: "memory");
#if HMX_RTZ3_ABL & 2
        wait_vmcnt<0>();
#else
        if (i + 1 < n_mine) wait_vmcnt<NI>(); else wait_vmcnt<0>();
#endif
        asm volatile("" ::: "memory");
        const int c0 = c_mine + 16 * stride * i;
        float* Rt = lds + (size_t)(2 * wv + (i & 1)) * buf_floats;
        float* Zt = Rt + 16 * Kp;
        const int n_live = min(16, c_end - c0);
        if (n_live < 16) {
            for (int j = n_live * Kp + lane; j < 16 * Kp; j += 64) Rt[j] = 0.f;
            for (int j = n_live * DP + lane; j < 16 * DP; j += 64) Zt[j] = 0.f;
        }
        const unsigned bw = reinterpret_cast<const unsigned*>(Zt + 16 * DP)[q];
        float afr[4][MT];
        f32x4 zfr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* rr = Rt + (size_t)(4 * q + ks) * Kp;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const f32x4 v = ld4(rr + 64 * h + 4 * c16);
#pragma unroll
                for (int j = 0; j < 4; ++j) afr[ks][4 * h + j] = v[j];
            }
#pragma unroll
            for (int j = 0; j < REM; ++j) afr[ks][4 * H + j] = rr[64 * H + REM * c16 + j];
            zfr[ks] = ld4(Zt + (size_t)(4 * q + ks) * DP + 4 * min(c16, KS - 1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the buffer is in registers: it may be overwritten
        if (i + 2 < n_mine && !(HMX_RTZ3_ABL & 2)) {
#pragma unroll
            for (int p = 0; p < NI; ++p) issue_piece(i & 1, p);
            advance();
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int bid = (bw >> (8 * ks)) & 255;
            float bfr[NT];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bfr[nt] = (c16 < KS) ? zfr[ks][nt] : ((bid == 4 * c16 + nt - DP) ? 1.f : 0.f);
#pragma unroll
            for (int e = 0; e < NTB; ++e) bfr[4 + e] = (bid == (64 - DP) + 16 * e + c16) ? 1.f : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(afr[ks][mt], bfr[nt], acc[mt][nt]);
        }
    }
  }

    R3STAMP(2);
    // ---- the four waves' accumulators meet in the task's slab: [output tile][lane][r], i.e. a lane's four values of a tile
    //      are 16 contiguous bytes.  Two waves at a time store their accumulators to two LDS regions (ds_write_b128, one per
    //      tile; one region when the LDS holds only one), then all threads add the regions into the slab (the second pair on
    //      top of the first: a read-modify-write of the thread's own 16-byte pieces, L2-resident).  No accumulator is read
    //      back into registers.  (The first version added one wave after the other with 4-byte LDS read-modify-writes:
    //      48 k cycles per wave, 12 % of the pass.)
    __syncthreads();                                                // every wave is done with its buffers (all requests landed)
    constexpr int PER = MT * NT * 256;
    const int nreg = (2 * PER <= 2 * RTZ3_WAVES * buf_floats) ? 2 : 1;   // workgroup-uniform
    float* slab = a.slab + (size_t)task * PER;
#pragma unroll 1
    for (int r0 = 0; r0 < RTZ3_WAVES; r0 += nreg) {
        if (wv >= r0 && wv < r0 + nreg) {
            float* reg = lds + (size_t)(wv - r0) * PER + 4 * lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) st4(reg + (mt * NT + nt) * 256, acc[mt][nt]);
        }
        __syncthreads();
        for (int j = tid; j < PER / 4; j += 64 * RTZ3_WAVES) {
            f32x4 v = ld4(lds + 4 * j);
            if (nreg == 2) v += ld4(lds + PER + 4 * j);
            if (r0 > 0) v += ld4(slab + 4 * j);
            st4(slab + 4 * j, v);
        }
        __syncthreads();                                            // (the regions are overwritten by the next pair)
    }
    R3STAMP(3);
    R3ACC(5, (unsigned long long)n_mine);
}

// ------------------------------------------------------------------------------------------
// k_rtz3c: k_rtz3 on the bf16 matrix pipe (hmx_device.h: every fp32 operand as the exact sum of three bf16 terms).  The k index
// runs over cells, so BOTH operands stream and both are split in registers; a k-step of v_mfma_f32_16x16x32_bf16 is 32 cells =
// a PAIR of the wave's tiles (any two of them), lane (c16, q) supplies the cells 8 (q & 1) .. + 8 of tile q >> 1.  Per pair and
// wave MT x (4 x 6 + 3 NTB) MFMAs of 16 cycles (the one-hot block columns are exact in bf16: one plane, three products) where
// k_rtz3 issues 2 x 4 x MT x (4 + NTB) of 32.  Same requests per tile, same slabs and finish kernel as k_rtz3.
// EIGHT waves per workgroup, two per SIMD, TWO tile buffers (one pair) each -- one workgroup per CU (156 KB at K = 100, d = 50).
// A wave requests a pair, waits for it, multiplies it, and only then requests its next pair: no double buffering inside the
// wave; the overlap comes from the SIMD's other wave, which multiplies while this one requests (24 LDS-DMA instructions of
// ~125 cycles of issue each) and waits.  The first cut of round 5 (k_rtz3b: four waves, two pair buffers each, a wave alone
// on its SIMD) left the matrix pipe idle during exactly those phases: 141.8 us per 1 M-cell pass at C3 against 127.5 here
// (stream alone 87-98 us, arithmetic alone 82-87 us: profiles/r04_micro_rtz_bf16_pipe.txt, r05_ab_rtz3c_eight_waves.txt).
// The tiles of a task are dealt to the eight waves as tile t0 + w + stride i (the host cuts the tasks with quads of eight:
// rtz3_quad).  Developed and float64-checked as a standalone kernel first: scripts/micro/rtz_bf3_tasks.hip.
// ------------------------------------------------------------------------------------------
#define RTZ3C_WAVES 8
template <int MT, int KS, int NTB>
__global__ __launch_bounds__(64 * RTZ3C_WAVES, 1) void k_rtz3c(Rtz3Args a) {
    if (a.frozen && *a.frozen) return;
    constexpr int NT = 4 + NTB, DP = 4 * KS;
    constexpr int NR = MT, NZ = (KS + 3) / 4, NI = NR + NZ + 1;     // VMEM operations per tile: R pieces, Z pieces, block ids
    constexpr int H = MT / 4, REM = MT % 4;
    static_assert(NTB <= 1 && KS <= 16 && 2 * NI < 64, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int Kp = a.Kp;
    const int buf_floats = 16 * (Kp + DP) + 4;                     // R tile | Z tile | 16 block-id bytes
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int lane16 = 16 * lane;
    const int task = blockIdx.x;
    const int t0 = a.task_t0[task], t1 = a.task_t1[task];
    const int c_first = a.task_c0[task], c_end = a.task_cend[task];
    const int stride = __builtin_amdgcn_readfirstlane(a.task_stride[task]);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_tiles = (t1 - t0 - wv + stride - 1) / stride;      // this wave's tiles: t0 + wv + stride i (< t1)
    const int n_mine = n_tiles > 0 ? (n_tiles + 1) / 2 : 0;        // ... taken two at a time
    const int c_mine = c_first + 16 * wv;
    const unsigned zone0 = lds_addr(lds + (size_t)(2 * wv) * buf_floats);
    const unsigned buf_bytes = __builtin_amdgcn_readfirstlane((unsigned)buf_floats * 4u);
    const unsigned r_bytes = __builtin_amdgcn_readfirstlane(64u * (unsigned)Kp);
    auto request = [&](int ti, int b) {                             // the NI requests of the wave's tile `ti` into its buffer b (0, 1)
        const size_t cell0 = (size_t)c_mine + (size_t)16 * stride * ti;
        const unsigned char* r = reinterpret_cast<const unsigned char*>(a.R + cell0 * Kp);
        const unsigned char* z = reinterpret_cast<const unsigned char*>(a.Z + cell0 * DP);
        const unsigned char* id = a.tile_blk + (size_t)16 * (t0 + wv + (size_t)stride * ti);
        const unsigned zb = zone0 + (unsigned)b * buf_bytes;
#pragma unroll
        for (int p = 0; p < NR; ++p)
            if (p + 1 < NR || 1024 * p + lane16 < 64 * Kp) dma16(r + 1024 * p, lane16, zb + 1024u * p);
#pragma unroll
        for (int it = 0; it < NZ; ++it)
            if (1024 * (it + 1) <= 64 * DP || 1024 * it + lane16 < 64 * DP) dma16<(HMX_RTZ3_Z_NT != 0)>(z + 1024 * it, lane16, zb + r_bytes + 1024u * it);
        if (lane == 0) dma16(id, lane16, zb + r_bytes + 64u * DP);
    };
    auto request_pair = [&](int i) {                                // (a missing second tile: the first one again, never read)
        request(2 * i, 0);
        request(2 * i + 1 < n_tiles ? 2 * i + 1 : 2 * i, 1);
    };
    if (n_mine > 0) request_pair(0);

    for (int i = 0; i < n_mine; ++i) {
        asm volatile("" ::: "memory");
        wait_vmcnt<0>();                                            // pair i has landed (nothing else of this wave travels)
        asm volatile("" ::: "memory");
        const bool has2 = 2 * i + 1 < n_tiles;                      // wave-uniform
        float* pb = lds + (size_t)(2 * wv) * buf_floats;
        // rows past the group's end hold other cells (or the slack behind the arrays): they count for nothing
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int c0 = c_mine + 16 * stride * (2 * i + u);
            const int n_live = (u == 0 || has2) ? min(16, c_end - c0) : 0;
            if (n_live < 16) {
                float* Rt = pb + (size_t)u * buf_floats;
                float* Zt = Rt + 16 * Kp;
                for (int j = n_live * Kp + lane; j < 16 * Kp; j += 64) Rt[j] = 0.f;
                for (int j = n_live * DP + lane; j < 16 * DP; j += 64) Zt[j] = 0.f;
                if (n_live == 0 && lane < 4) reinterpret_cast<unsigned*>(Zt + 16 * DP)[lane] = 0xFFFFFFFFu;   // no block
            }
        }
        // lane (c16, q): k slot j <-> cell 8 (q & 1) + j of tile q >> 1 of the pair
        const float* R = pb + (size_t)(q >> 1) * buf_floats + 8 * (q & 1) * Kp;
        const float* Z = pb + (size_t)(q >> 1) * buf_floats + 16 * Kp + 8 * (q & 1) * DP;
        const unsigned char* ids = reinterpret_cast<const unsigned char*>(pb + (size_t)(q >> 1) * buf_floats + 16 * (Kp + DP)) + 8 * (q & 1);
        u32x4 bh[4], bm[4], bl[4], oh = {0u, 0u, 0u, 0u};           // B planes of the four PC-column tiles, the one-hot tile's only plane
        {
            // the lane's eight block ids stay packed in two registers and are extracted where they are compared (eight
            // unpacked ids live across the split pushed the 7 x 5 instance over its 256 registers)
            const u32x2 idw = *reinterpret_cast<const u32x2*>(ids);
            auto bidj = [&](int j) { return (int)((idw[j >> 2] >> (8 * (j & 3))) & 255u); };
#pragma unroll
            for (int half = 0; half < 2; ++half) {                  // two column tiles at a time: 8-byte reads, 16 raw registers live
                f32x2 z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = *reinterpret_cast<const f32x2*>(Z + j * DP + 4 * min(c16, KS - 1) + 2 * half);
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int nt = 2 * half + n2;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        f32x2 x;
                        x.x = (c16 < KS) ? z[2 * p][n2] : ((bidj(2 * p) == 4 * c16 + nt - DP) ? 1.f : 0.f);
                        x.y = (c16 < KS) ? z[2 * p + 1][n2] : ((bidj(2 * p + 1) == 4 * c16 + nt - DP) ? 1.f : 0.f);
                        unsigned h, m, l;
                        bf16_split3(x, h, m, l);
                        bh[nt][p] = h; bm[nt][p] = m; bl[nt][p] = l;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (NTB > 0) {
#pragma unroll
                for (int p = 0; p < 4; ++p)                          // bf16(1.0) = 0x3F80
                    oh[p] = ((bidj(2 * p) == (64 - DP) + c16) ? 0x3F80u : 0u) | ((bidj(2 * p + 1) == (64 - DP) + c16) ? 0x3F800000u : 0u);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto one_mt = [&](int mt, const float (&av)[8]) {
            u32x4 ah, am, al;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned h, m, l;
                bf16_split3((f32x2){av[2 * p], av[2 * p + 1]}, h, m, l);
                ah[p] = h; am[p] = m; al[p] = l;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {                         // smallest terms first
                acc[mt][nt] = MFMA_BF16(al, bh[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(ah, bl[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(am, bm[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(am, bh[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(ah, bm[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(ah, bh[nt], acc[mt][nt]);
            }
            if (NTB > 0) {
                acc[mt][NT - 1] = MFMA_BF16(al, oh, acc[mt][NT - 1]);
                acc[mt][NT - 1] = MFMA_BF16(am, oh, acc[mt][NT - 1]);
                acc[mt][NT - 1] = MFMA_BF16(ah, oh, acc[mt][NT - 1]);
            }
        };
        if constexpr (MT * NT >= 35) {
            // 140 accumulators: the values of ONE cluster tile at a time (4-byte reads; with two tiles' values live -- the 8-byte
            // reads below -- the instance spilled seven registers)
#pragma unroll
            for (int mt = 0; mt < 4 * H; ++mt) {
                float av[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = R[j * Kp + 64 * (mt >> 2) + 4 * c16 + (mt & 3)];
                one_mt(mt, av);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int half = 0; half < 2; ++half) {                  // k_rtz3's groups of four cluster tiles, two at a time (8-byte reads)
                f32x2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x2*>(R + j * Kp + 64 * h + 4 * c16 + 2 * half);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float av[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[j] = v[j][jj];
                    one_mt(4 * h + 2 * half + jj, av);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < REM; ++jj) {
            float av[8];
            const int col = 64 * H + REM * c16 + jj;                 // clusters past the row do not exist: no read past it
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                av[j] = R[j * Kp + min(col, Kp - 1)];
                if (col >= Kp) av[j] = 0.f;
            }
            one_mt(4 * H + jj, av);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the pair's buffers are in registers: they take the wave's next pair
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 1 < n_mine) request_pair(i + 1);
    }

    // ---- the eight waves' accumulators meet in the task's slab (as in k_rtz3, four waves at a time when the LDS holds four regions) ----
    __syncthreads();                                                // every wave is done with its buffers (all requests landed)
    constexpr int PER = MT * NT * 256;
    const int room = (2 * RTZ3C_WAVES * buf_floats) / PER;          // regions the LDS holds (workgroup-uniform)
    const int nreg = room >= 4 ? 4 : room >= 2 ? 2 : 1;
    float* slab = a.slab + (size_t)task * PER;
#pragma unroll 1
    for (int r0 = 0; r0 < RTZ3C_WAVES; r0 += nreg) {
        if (wv >= r0 && wv < r0 + nreg) {
            float* reg = lds + (size_t)(wv - r0) * PER + 4 * lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) st4(reg + (mt * NT + nt) * 256, acc[mt][nt]);
        }
        __syncthreads();
        for (int j = tid; j < PER / 4; j += 64 * RTZ3C_WAVES) {
            f32x4 v = ld4(lds + 4 * j);
            for (int r = 1; r < nreg; ++r) v += ld4(lds + (size_t)r * PER + 4 * j);
            if (r0 > 0) v += ld4(slab + 4 * j);
            st4(slab + 4 * j, v);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// k_rtzw: the same streaming pass for shapes whose output does not fit one wave's registers (K up to 208, d up to 208:
// BASELINE configs[4], the f32-MFMA-bound regime).  Eight waves share every 16-cell tile: the tile's R rows, Z rows and block
// ids travel global -> LDS (each wave requests its share of the 1 KB pieces) into a ring of four tile buffers, three tiles
// ahead; one workgroup barrier per tile says "tile i is complete, nobody reads tile i-1 any more"; wave w owns the column
// tiles w and w + 8 of the output and ALL MT cluster tiles of them (2 x MT accumulators), so nothing is reduced across
// waves -- every wave stores its own output tiles.  Cluster rows use k_rtz3's permuted map (16-byte A reads); columns are
// plain: PC tile nt holds columns 16 nt .., the columns d .. dp-1 the row padding leaves free in the last PC tile carry
// the first one-hot block columns, whole extra tiles the rest.
// ------------------------------------------------------------------------------------------
#define RTZW_WAVES 8
#define RTZW_NBUF 4
template <int MT>
__global__ __launch_bounds__(64 * RTZW_WAVES, 1) void k_rtzw(Rtz3Args a) {
    constexpr int H = MT / 4, REM = MT % 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int Kp = a.Kp, DP = a.dp, d = a.d, NT = a.nt, NTP = DP >> 4;
    const int buf_floats = 16 * (Kp + DP) + 4;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int lane16 = 16 * lane;
    const int task = blockIdx.x;
    const int t0 = a.task_t0[task], t1 = a.task_t1[task];
    const int c_first = a.task_c0[task], c_end = a.task_cend[task];
    const int stride = __builtin_amdgcn_readfirstlane(a.task_stride[task]);   // tiles between two tiles of this workgroup
    const int n_tiles = (t1 - t0 + stride - 1) / stride;

    // this wave's output column tiles
    const int nt0 = wv, nt1 = wv + RTZW_WAVES;
    const bool has0 = nt0 < NT, has1 = nt1 < NT;                   // wave-uniform
    f32x4 acc0[MT], acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc0[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // requests: a tile is n_pr pieces of R (64 Kp bytes), n_pz of Z (64 dp bytes) and one of block ids; piece p goes to wave p % 8
    const int n_pr = (64 * Kp + 1023) / 1024, n_pz = (64 * DP + 1023) / 1024, n_p = n_pr + n_pz + 1;
    const int npw = (n_p - wv + RTZW_WAVES - 1) / RTZW_WAVES;     // requests of this wave per tile (wave-uniform)
    auto uniform64 = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    };
    const unsigned zone0 = lds_addr(lds);
    auto issue = [&](int i) {                                       // this wave's pieces of tile i of the task
        const int c0 = c_first + 16 * stride * i;
        const unsigned zb = zone0 + (unsigned)(i % RTZW_NBUF) * (unsigned)buf_floats * 4u;
        const unsigned long long rs = uniform64((unsigned long long)(a.R + (size_t)c0 * Kp));
        const unsigned long long zs = uniform64((unsigned long long)(a.Z + (size_t)c0 * DP));
        const unsigned long long bs = uniform64((unsigned long long)(a.tile_blk + (size_t)16 * (t0 + stride * i)));
        for (int p = wv; p < n_p; p += RTZW_WAVES) {
            if (p < n_pr) {
                if (1024 * p + lane16 < 64 * Kp) dma16((const void*)(rs + 1024ull * p), lane16, zb + 1024u * p);
            } else if (p < n_pr + n_pz) {
                const int it = p - n_pr;
                if (1024 * it + lane16 < 64 * DP) dma16((const void*)(zs + 1024ull * it), lane16, zb + 64u * Kp + 1024u * it);
            } else {
                if (lane == 0) dma16((const void*)bs, lane16, zb + 64u * (Kp + DP));
            }
        }
    };
    R3STAMP8(0);
    for (int i = 0; i < RTZW_NBUF - 1 && i < n_tiles; ++i) issue(i);

    // One stream of k-steps per wave, as in k_rtz3: the fragments of the next k-step are read while the 2 x MT MFMAs of the
    // current one are in the matrix pipe; inside the last k-step of tile i the wave waits for its pieces of tile i+1,
    // meets the others at the barrier ("tile i+1 is complete, nobody reads tile i any more" -- every wave's last reads of
    // tile i returned before its last k-step began), hands tile i's buffer to tile i+4 and reads tile i+1's first
    // fragments.  No branches in the stream: a wave without a second (or first) column tile multiplies zeros -- the SIMD
    // that hosts it is not the busiest one anyway, and straight-line code is what lets the compiler overlap reads and MFMAs.
    // (First version: barrier, requests, then per k-step "read fragments, multiply" behind `if (has tile)` branches: the
    // eight waves left every barrier in lock-step and paid every LDS latency together -- 1.43 ms per pass at configs[4],
    // slower than the kernel it replaces.)
    // raw fragments of a k-step (two register sets) and what turns them into operands at USE time: the row-is-live factor,
    // the PC-column masks and the one-hot values.  Masking at read time put a wait for the LDS data right behind every
    // read -- and an in-order wave cannot issue the MFMAs queued behind that wait.
    float afr[2][MT];
    float zq[2][2], mq[2][2], oq[2][2], lmq[2];
    const int spare = DP - d;
    auto wait_mine = [&](int i) {                                   // this wave's pieces of tile i have landed
        asm volatile("" ::: "memory");
#ifdef HMX_RTZ3_PROF
        const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
        const int younger = min(RTZW_NBUF - 2, n_tiles - 1 - i) * npw;       // wave-uniform, one of a few values
        if (younger >= 8) wait_vmcnt<8>(); else if (younger == 6) wait_vmcnt<6>(); else if (younger == 4) wait_vmcnt<4>();
        else if (younger == 3) wait_vmcnt<3>(); else if (younger == 2) wait_vmcnt<2>(); else if (younger == 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
#ifdef HMX_RTZ3_PROF
        const unsigned long long w1_ = __builtin_amdgcn_s_memtime();
        R3ACC8(4, w1_ - w0_);
        wg_barrier_lds();
        R3ACC8(6, __builtin_amdgcn_s_memtime() - w1_);
#else
        wg_barrier_lds();                                           // ... and everybody's: the tile is complete; nobody reads the tile before it any more
#endif
        asm volatile("" ::: "memory");
    };
    auto read_frags = [&](int i, int ks, int set, unsigned bwv) {
        const int c0 = c_first + 16 * stride * i;
        const float* Rt = lds + (size_t)(i % RTZW_NBUF) * buf_floats;
        const float* Zt = Rt + 16 * Kp;
        const int cell = 4 * q + ks;
        const bool live = cell < c_end - c0;                        // rows past the group's end count for nothing: a factor, not
        lmq[set] = live ? 1.f : 0.f;                                // a select around the read (the rows in LDS are finite)
        const float* rr = Rt + (size_t)cell * Kp;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x4 v = ld4(rr + 64 * h + 4 * c16);
#pragma unroll
            for (int j = 0; j < 4; ++j) afr[set][4 * h + j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < REM; ++j) afr[set][4 * H + j] = rr[64 * H + REM * c16 + j];
        const int bid = (bwv >> (8 * ks)) & 255;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // B operand of this wave's column tile u for the lane's column c16: a PC column (Z value), a one-hot block column
            // in the row padding of the last PC tile or in an extra tile, or nothing: z * mask + one-hot
            const int nt = u ? nt1 : nt0;
            const int col = 16 * min(nt, NTP - 1) + c16;            // (clamped: a read that is not used stays inside the tile)
            zq[set][u] = Zt[(size_t)cell * DP + col];
            mq[set][u] = (nt < NTP && col < d && live) ? 1.f : 0.f;
            const int blk_col = nt < NTP ? col - d : spare + 16 * (nt - NTP) + c16;   // block whose one-hot column this is (negative: none)
            oq[set][u] = (nt < NT && blk_col == bid && blk_col >= 0) ? 1.f : 0.f;
        }
    };
    unsigned bw = 0, bw_next = 0;
    if (n_tiles > 0) {
        wait_mine(0);
        if (RTZW_NBUF - 1 < n_tiles) issue(RTZW_NBUF - 1);
        bw_next = reinterpret_cast<const unsigned*>(lds + 16 * (Kp + DP))[q];
        read_frags(0, 0, 0, bw_next);
    }
    R3STAMP8(1);
    for (int i = 0; i < n_tiles; ++i) {
        bw = bw_next;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int set = ks & 1;
            // operands of this k-step from the fragments read a whole k-step ago
            float am[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) am[mt] = afr[set][mt] * lmq[set];
            const float b0 = fmaf(zq[set][0], mq[set][0], oq[set][0]), b1 = fmaf(zq[set][1], mq[set][1], oq[set][1]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks < 3) read_frags(i, ks + 1, set ^ 1, bw);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc0[mt] = MFMA16(am[mt], b0, acc0[mt]);
                if (ks == 3 && mt == MT / 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (i + 1 < n_tiles) {
                        wait_mine(i + 1);                           // tile i+1 is complete; nobody reads tile i any more
                        if (i + RTZW_NBUF < n_tiles) issue(i + RTZW_NBUF);
                        bw_next = reinterpret_cast<const unsigned*>(lds + (size_t)((i + 1) % RTZW_NBUF) * buf_floats + 16 * (Kp + DP))[q];
                        read_frags(i + 1, 0, 0, bw_next);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc1[mt] = MFMA16(am[mt], b1, acc1[mt]);
            __builtin_amdgcn_sched_barrier(0);                      // the next k-step's operands are formed after these MFMAs, not before
        }
    }
    R3STAMP8(2);
    R3ACC8(5, (unsigned long long)n_tiles);
    // every wave stores its own output tiles: slab [mt][nt][lane][r]
    float* slab = a.slab + (size_t)task * ((size_t)MT * NT * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (has0) st4(slab + ((size_t)(mt * NT + nt0) * 64 + lane) * 4, acc0[mt]);
        if (has1) st4(slab + ((size_t)(mt * NT + nt1) * 64 + lane) * 4, acc1[mt]);
    }
}

// ------------------------------------------------------------------------------------------
// k_rtzw2b: the wide streaming pass (K > 112, seven to fourteen column tiles) on the bf16 matrix pipe (hmx_device.h: every fp32
// operand as the exact sum of three bf16 terms, six products, fp32 accumulation).  configs[4] is bound by the f32-input MFMA
// outright (8 x 32 cycles per 32 cells and output tile; here 6 x 16), so this is where the split moves the BOUND, not only
// the constant.  The k index runs over cells: a k-step of v_mfma_f32_16x16x32_bf16 is a PAIR of the task's tiles, lane (c16, q)
// supplies cells 8 (q & 1) .. + 8 of tile q >> 1 (as in k_rtz3c), both operands are split in registers.  Tasks, slabs
// [mt][nt][lane][r] and finish kernel are k_rtzw's; the MT x NT output tiles are split 2 x 2 over FOUR waves (row half x
// column half: up to 7 x 7 tiles = 196 accumulators; k_rtzw's eight waves share one tile and hold 2 x MT).  Further:
//   * one workgroup per CU (512 registers per lane: 196 accumulators + the 84 registers of the column half's B planes);
//   * the tiles travel global -> registers -> LDS with ORDINARY loads (non-temporal), half a pair at a time, under the
//     multiply of the pair before: the compiler counts the waits, rows past the group's end (and the missing second tile
//     of an odd count) are written as zeros -- no live-row factor, no select in the loop -- and an LDS-DMA request's
//     ~200 cycles of issue are not paid by a wave that has no partner on its SIMD;
//   * A values come as 16-byte LDS reads (four cluster tiles of the permuted row map per read);
//   * every column tile takes the same six products -- a PC tile, the tile whose padding carries the first one-hot block
//     columns, a pure one-hot tile (its m and l planes are zero) or a tile past NT (all zero): no branch in the multiply.
// Measured at the configs[4] shard: 866 us per pass (f32-input kernels: 1 144); the pass issues ~2 500 instructions per pair
// of tiles and wave beside its 294 MFMAs (splits, staging predicates, waits) -- that count, not the matrix pipe, HBM or the
// staging distance, is what bounds it (NOTES.md, round 5: an eight-wave 2 x 4 version and a deeper staging were slower).
// ------------------------------------------------------------------------------------------
#define RTZWB_WAVES 4
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F& f) {                    // f(integral_constant<int, I>) for I .. N-1, unrolled by construction
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// ZF (round 6): Z_cos does not change during the rounds of a Harmony iteration, so its three bf16 planes are split ONCE per
// iteration (k_zplanes) and stored per 16-cell static tile in the B-FRAGMENT order of this kernel -- [column tile][plane h, m, l]
// [lane (q & 1, c16)][8 bf16 = cells 8 (q & 1) .. + 8] = 512 bytes per (tile, column tile, plane).  The pass then stages those
// instead of fp32 rows (+50 % of the Z bytes: 1 248 instead of 832 per cell at 200 PCs, still far below the matrix floor), a
// wave's B planes are three conflict-free 16-byte LDS reads per column tile, and the serial prologue of every pair -- 56
// four-way conflicted 4-byte reads and 28 splits per wave with the matrix pipe idle -- is gone; only the one-hot block
// columns (the last one or two column tiles) are still formed per pair.  The ridge statistics (Z_orig) keep ZF = false.
template <int MT, int NTH, bool ZF>
__global__ __launch_bounds__(64 * RTZWB_WAVES, 1) void k_rtzw2b(Rtz3Args a) {
    if (a.frozen && *a.frozen) return;
    constexpr int MTA = (MT + 1) / 2;
    constexpr int H = MT / 4, REM = MT % 4;
    constexpr int NRP = (64 * MT + 255) / 256;                       // 16-byte pieces of an R tile per thread (16 rows x Kp <= 16 MT floats)
    constexpr int NZP = ZF ? 5 : 4;                                  // ... of a Z tile (dp <= 208: 832 pieces of fp32 rows, or 13 x 96 pieces of planes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int Kp = a.Kp, DP = a.dp, d = a.d, NT = a.nt, NTP = DP >> 4;
    const int zt_floats = ZF ? NTP * 3 * 128 : 16 * DP;              // a tile's Z part: fragments of three planes, or fp32 rows
    const int buf_floats = 256 * MT + zt_floats + 4;                 // R tile (rows at stride Kp, padded to MT KB) | Z tile | 16 block-id bytes
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int task = blockIdx.x;
    const int t0 = a.task_t0[task], t1 = a.task_t1[task];
    const int c_first = a.task_c0[task], c_end = a.task_cend[task];
    const int stride = __builtin_amdgcn_readfirstlane(a.task_stride[task]);
    const int n_tiles = (t1 - t0 + stride - 1) / stride;
    const int n_pairs = (n_tiles + 1) / 2;
    if (n_tiles <= 0) return;                                         // (workgroup-uniform; the slab of such a task is never read: no task is built without a tile)
    const int rh = wv >> 1, ch = wv & 1;                             // this wave's quarter of the output: row half, column half
    const int nt_lo = ch * NTH;
    const int spare = DP - d;

    // ---- staging: tile `ti` of the task (clamped: a missing tile is loaded from the last one and written as zeros) ----
    f32x4 sr[NRP], sz[NZP], sid;
    auto load_tile = [&](int ti) {
        const int tc = min(ti, n_tiles - 1);
        const size_t cell0 = (size_t)c_first + (size_t)16 * stride * tc;
        const float* rsrc = a.R + cell0 * Kp;
        const float* zsrc = ZF ? reinterpret_cast<const float*>(a.Zf) + ((size_t)t0 + (size_t)stride * tc) * zt_floats : a.Z + cell0 * DP;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NRP; ++j) {
            const int p = tid + 256 * j;
            sr[j] = (p < 4 * Kp) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rsrc) + p) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NZP; ++j) {
            const int p = tid + 256 * j;
            sz[j] = (4 * p < zt_floats) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(zsrc) + p) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        sid = (tid == 0) ? *reinterpret_cast<const f32x4*>(a.tile_blk + (size_t)16 * (t0 + (size_t)stride * tc)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
    };
    auto store_tile = [&](int ti, int b) {                            // ... into tile buffer b (0..3)
        const int cell0 = c_first + 16 * stride * min(ti, n_tiles - 1);
        const int n_live = ti < n_tiles ? min(16, c_end - cell0) : 0; // rows of the tile inside the group (workgroup-uniform)
        float* Rt = lds + (size_t)b * buf_floats;
        float* Zt = Rt + 256 * MT;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NRP; ++j) {
            const int p = tid + 256 * j;
            if (p < 4 * Kp) st4(Rt + 4 * p, (4 * p < n_live * Kp) ? sr[j] : (f32x4){0.f, 0.f, 0.f, 0.f});
        }
#pragma unroll
        for (int j = 0; j < NZP; ++j) {
            const int p = tid + 256 * j;
            // (planes: the rows past the group's end are zero in the stored fragments; a missing tile is written as zeros)
            if (4 * p < zt_floats) st4(Zt + 4 * p, (ZF ? n_live > 0 : 4 * p < n_live * DP) ? sz[j] : (f32x4){0.f, 0.f, 0.f, 0.f});
        }
        if (tid == 0) st4(Zt + zt_floats, sid);
        __builtin_amdgcn_sched_barrier(0);
    };

    auto body = [&](auto rh_c) {
        constexpr int RH = decltype(rh_c)::value;
        constexpr int LO = RH * MTA, HI = RH ? MT : MTA;              // this wave's row tiles
        f32x4 acc[MTA][NTH];
#pragma unroll
        for (int t = 0; t < MTA; ++t)
#pragma unroll
            for (int u = 0; u < NTH; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

        load_tile(0);
        store_tile(0, 0);
        load_tile(1);
        store_tile(1, 1);

        for (int i = 0; i < n_pairs; ++i) {
            wg_barrier_lds();                                         // pair i is complete in LDS; nobody reads pair i-1 any more
            const float* pb = lds + (size_t)(2 * (i & 1)) * buf_floats;
            const int nb = 2 * ((i + 1) & 1);                         // the tile buffers of pair i+1 (pair i-1's)
            load_tile(2 * i + 2);                                     // travels under the B planes and the first row tiles
            // lane (c16, q): k slot j <-> cell 8 (q & 1) + j of tile q >> 1 of the pair
            const float* Rl = pb + (size_t)(q >> 1) * buf_floats + 8 * (q & 1) * Kp;
            const float* Zl = Rl + 256 * MT - 8 * (q & 1) * Kp + 8 * (q & 1) * DP;
            const unsigned char* ids = reinterpret_cast<const unsigned char*>(pb + (size_t)(q >> 1) * buf_floats + 256 * MT + zt_floats) + 8 * (q & 1);
            int bid[8];
            {
                const u32x2 w = *reinterpret_cast<const u32x2*>(ids);
#pragma unroll
                for (int j = 0; j < 8; ++j) bid[j] = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
            }
            // ---- B planes of the column half: value = PC column (zero in the row padding) + one-hot of the block column ----
            u32x4 bh[NTH], bm[NTH], bl[NTH];
            if constexpr (ZF) {
                // the lane's fragment of tile q >> 1: 16 bytes per (column tile, plane), lanes of a quarter wave side by side
                const unsigned* zf = reinterpret_cast<const unsigned*>(pb + (size_t)(q >> 1) * buf_floats + 256 * MT) + 4 * (16 * (q & 1) + c16);
#pragma unroll
                for (int u = 0; u < NTH; ++u) {
                    const int nt = nt_lo + u;                         // wave-uniform
                    u32x4 h = (u32x4){0u, 0u, 0u, 0u}, m = h, l = h;
                    if (nt < NTP) {
                        const unsigned* f = zf + (size_t)nt * 3 * 128;
                        h = ld4u(f); m = ld4u(f + 128); l = ld4u(f + 256);
                    }
                    if (nt >= NTP - 1 && nt < NT) {                   // one-hot block columns: 1.0 = 0x3F80 in the h plane of a column whose PC planes are zero
                        const int blk_col = nt < NTP ? 16 * nt + c16 - d : spare + 16 * (nt - NTP) + c16;
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            h[p] |= (bid[2 * p] == blk_col ? 0x3F80u : 0u) | (bid[2 * p + 1] == blk_col ? 0x3F800000u : 0u);
                    }
                    bh[u] = h; bm[u] = m; bl[u] = l;
                }
            } else
#pragma unroll
            for (int u = 0; u < NTH; ++u) {
                const int nt = nt_lo + u;                             // wave-uniform
                const bool pc = nt < NTP;
                const float* zr = Zl + 16 * min(nt, NTP - 1) + c16;   // (clamped: an unused read stays inside the tile)
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = zr[j * DP];
                if (!pc) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = 0.f;
                }
                if (nt >= NTP - 1 && nt < NT) {                       // block whose one-hot column this lane's column is (negative: none)
                    const int blk_col = pc ? 16 * nt + c16 - d : spare + 16 * (nt - NTP) + c16;
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] += (bid[j] == blk_col) ? 1.f : 0.f;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    unsigned h, m, l;
                    bf16_split3((f32x2){x[2 * p], x[2 * p + 1]}, h, m, l);
                    bh[u][p] = h; bm[u][p] = m; bl[u][p] = l;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- row tiles, software-pipelined: while the 6 NTH products of tile t are in the matrix pipe, the A values of tile
            //      t+1 are read (16-byte reads: the four tiles of a group of the permuted row map share them) and split.  One
            //      scheduling region per tile; the hints ask for one VALU instruction behind every MFMA (a wave without a
            //      partner on its SIMD hides the split only inside its own instruction stream).  Products: smallest terms
            //      first, consecutive MFMAs go to different accumulators.
            f32x4 v[8];
            u32x4 pa[2][3];                                           // planes (h, m, l) of the current / the next tile
            auto fetch_split = [&](auto tc, u32x4 (&pl)[3]) {         // tile LO + T of the wave's half -> planes
                constexpr int T = decltype(tc)::value;
                constexpr int mt = LO + T;
                float av[8];
                if constexpr (mt < 4 * H) {
                    if constexpr (T == 0 || (mt & 3) == 0) {          // first tile of its group in this half: the group's values
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = ld4(Rl + j * Kp + 64 * (mt >> 2) + 4 * c16);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[j] = v[j][mt & 3];
                } else {
                    const int col = 64 * H + REM * c16 + (mt - 4 * H);   // clusters past the row do not exist: no read past it
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        av[j] = Rl[j * Kp + min(col, Kp - 1)];
                        if (col >= Kp) av[j] = 0.f;
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    unsigned h, m, l;
                    bf16_split3((f32x2){av[2 * p], av[2 * p + 1]}, h, m, l);
                    pl[0][p] = h; pl[1][p] = m; pl[2][p] = l;
                }
            };
            auto products = [&](int t, const u32x4 (&pl)[3]) {
#pragma unroll
                for (int u = 0; u < NTH; ++u) acc[t][u] = MFMA_BF16(pl[2], bh[u], acc[t][u]);
#pragma unroll
                for (int u = 0; u < NTH; ++u) acc[t][u] = MFMA_BF16(pl[0], bl[u], acc[t][u]);
#pragma unroll
                for (int u = 0; u < NTH; ++u) acc[t][u] = MFMA_BF16(pl[1], bm[u], acc[t][u]);
#pragma unroll
                for (int u = 0; u < NTH; ++u) acc[t][u] = MFMA_BF16(pl[1], bh[u], acc[t][u]);
#pragma unroll
                for (int u = 0; u < NTH; ++u) acc[t][u] = MFMA_BF16(pl[0], bm[u], acc[t][u]);
#pragma unroll
                for (int u = 0; u < NTH; ++u) acc[t][u] = MFMA_BF16(pl[0], bh[u], acc[t][u]);
            };
            fetch_split(std::integral_constant<int, 0>{}, pa[0]);
            __builtin_amdgcn_sched_barrier(0);
            auto tile_step = [&](auto tc) {
                constexpr int T = decltype(tc)::value;
                if constexpr (T + 1 < HI - LO) fetch_split(std::integral_constant<int, T + 1>{}, pa[(T + 1) & 1]);
                products(T, pa[T & 1]);
                if constexpr (T + 1 < HI - LO) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the LDS reads of the next tile's values first
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // ... their latency under the first products
#pragma unroll
                    for (int r = 0; r < 6 * NTH - 6; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (T + 1 == (HI - LO + 1) / 2) {           // half way: the first tile of pair i+1 has landed
                    store_tile(2 * i + 2, nb);
                    load_tile(2 * i + 3);
                }
            };
            static_for<0, HI - LO>(tile_step);
            store_tile(2 * i + 3, nb + 1);
        }
        // every wave stores its own output tiles: slab [mt][nt][lane][r]
        float* slab = a.slab + (size_t)task * ((size_t)MT * NT * 256);
#pragma unroll
        for (int t = 0; t < MTA; ++t)
#pragma unroll
            for (int u = 0; u < NTH; ++u) {
                const int mt = LO + t, nt = nt_lo + u;
                if (mt < HI && nt < NT) st4(slab + ((size_t)(mt * NT + nt) * 64 + lane) * 4, acc[t][u]);
            }
    };
    if (rh == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------
// k_rtz3_finish: the per-task slabs summed in fp64, one workgroup per cluster k; undoes the index maps of k_rtz3.
//   mode 0 (k-means round): Ysum[k][pc] over all tasks (the centroid numerators, :443), Sold[blk][g][k] over the tasks of
//          group g (the removal sums, :491-492); with Yout the row is normalised on the spot (:444) -- no collective
//          is due in between on a single engine -- and the workgroups share the fills the sweep kernel needs.
//   mode 1 (ridge): Sr[g][k][pc] (:556-563) and the exact Oxr[g][k] (:550) over the tasks of group g.
// ------------------------------------------------------------------------------------------
#define RTZ3_FIN_THREADS 1024   /* many short chains of dependent-free loads: the slab reads are latency-bound */
__global__ __launch_bounds__(RTZ3_FIN_THREADS) void k_rtz3_finish(Rtz3FinishArgs a) {
    if (a.frozen && *a.frozen) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tab = reinterpret_cast<double*>(smem);                  // G x NV
    const int NT = a.wide ? a.NT : 4 + a.NTB, NV = 16 * NT, DP = 4 * a.KS;
    const int k = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < a.G * NV; i += RTZ3_FIN_THREADS) tab[i] = 0.0;
    if (a.zero_p) {                                                 // fill duty: this workgroup's slice
        const size_t per = (a.zero_n + gridDim.x - 1) / gridDim.x;
        const size_t z0 = (size_t)k * per, z1 = min(z0 + per, a.zero_n);
        for (size_t i = z0 + tid; i < z1; i += RTZ3_FIN_THREADS) a.zero_p[i] = 0.0;
    }
    if (a.zero2_p)
        for (size_t i = (size_t)k * RTZ3_FIN_THREADS + tid; i < a.zero2_n; i += (size_t)gridDim.x * RTZ3_FIN_THREADS) a.zero2_p[i] = 0.0;
    if (a.copy_dst)
        for (int i = k * RTZ3_FIN_THREADS + tid; i < a.copy_n; i += gridDim.x * RTZ3_FIN_THREADS) a.copy_dst[i] = a.copy_src[i];
    __syncthreads();
    // where cluster k sits in a slab: tile mt, row m of the tile
    const int Hq = a.MT / 4, rem = a.MT % 4;
    int mt, m;
    if (k < 64 * Hq) { mt = 4 * (k / 64) + (k & 3); m = (k & 63) >> 2; }
    else { const int x = k - 64 * Hq; m = x / rem; mt = 4 * Hq + x % rem; }
    const int per = a.MT * NT * 256;
    const int nslice = RTZ3_FIN_THREADS / NV;
    const int v = tid % NV, sl = tid / NV;                          // value (nt, n) of the row, slice of the tasks
    if (sl < nslice) {
        const int nt = v >> 4, n = v & 15;
        const size_t off = (size_t)(mt * NT + nt) * 256 + (size_t)(16 * (m >> 2) + n) * 4 + (m & 3);   // [tile][lane][r]: row m = 4 q + r, column n
        int g = -1;
        double acc = 0.0;
        // the slab reads are latency-bound: sixteen independent loads in flight per thread (clamped, not predicated: a
        // select on the loaded value keeps them one batch), then folded in task order (tasks are sorted by group)
        constexpr int BATCH = 16;
        for (int w0 = sl; w0 < a.ntasks; w0 += BATCH * nslice) {
            float val[BATCH];
            int grp[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int w = min(w0 + u * nslice, a.ntasks - 1);
                val[u] = a.slab[(size_t)w * per + off];
                grp[u] = a.task_grp[w];
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (w0 + u * nslice >= a.ntasks) break;
                if (grp[u] != g) {
                    if (g >= 0 && acc != 0.0) atomicAdd(&tab[g * NV + v], acc);
                    g = grp[u];
                    acc = 0.0;
                }
                acc += (double)val[u];
            }
        }
        if (g >= 0 && acc != 0.0) atomicAdd(&tab[g * NV + v], acc);
    }
    __syncthreads();
    // column of a PC / of a block's one-hot in the row's NV values (v = 16 nt + n): k_rtz3's permuted tiles, or k_rtzw's plain ones
    auto col_pc = [&](int pc) { return a.wide ? pc : 16 * (pc & 3) + (pc >> 2); };
    auto col_blk = [&](int j) {
        if (a.wide) {                                               // spare columns d .. DP-1 of the last PC tile first, then whole tiles
            const int spare = DP - a.d;
            return j < spare ? a.d + j : DP + (j - spare);
        }
        if (j < 64 - DP) { const int c = DP + j; return 16 * (c & 3) + (c >> 2); }
        const int x = j - (64 - DP);
        return 16 * (4 + x / 16) + (x & 15);
    };
    if (a.mode == 1) {
        for (int i = tid; i < a.G * a.ld; i += RTZ3_FIN_THREADS) {
            const int g = i / a.ld, j = i - g * a.ld;
            a.Sr[((size_t)g * a.K16 + k) * a.ld + j] = (k < a.K && j < a.d) ? tab[g * NV + col_pc(j)] : 0.0;
        }
        for (int g = tid; g < a.G; g += RTZ3_FIN_THREADS) a.Oxr[(size_t)g * a.K16 + k] = (k < a.K) ? tab[g * NV + col_blk(0)] : 0.0;
        return;
    }
    for (int i = tid; i < a.nblk * a.G; i += RTZ3_FIN_THREADS) {
        const int b = i / a.G, g = i - b * a.G;
        a.Sold[((size_t)b * a.G + g) * a.K16 + k] = (k < a.K) ? tab[g * NV + col_blk(b)] : 0.0;
    }
    if (tid < 64) {                                                 // one wave: the row's numerators, then (:444) its unit form
        const int lane = tid;
        float vals[4];
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            double s = 0.0;
            if (k < a.K && j < a.d)
                for (int g = 0; g < a.G; ++g) s += tab[g * NV + col_pc(j)];
            if (j < a.ld) a.Ysum[(size_t)k * a.ld + j] = s;
            vals[u] = (float)s;                                     // fp64 sums are rounded to fp32 first (:443)
            ss += vals[u] * vals[u];
        }
        if (a.Yout) {
            for (int msk = 32; msk >= 1; msk >>= 1) ss += __shfl_xor(ss, msk, 64);
            const float nrm = sqrtf(ss);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = lane + 64 * u;
                if (j < a.ld) a.Yout[(size_t)k * a.ld + j] = (k < a.K && j < a.d) ? vals[u] / nrm : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_tile_blocks: block id of every cell in STATIC tile order (16 bytes per static tile), from a round's block-major list
// (cells, tile groups, block tile offsets: built on the host from torch.randperm or on the device from the keyed
// bijection, harmony.py:471-484).  Static position of internal cell c of group g: 16 * s_tile_start[g] + (c - gstart[g]).
// Positions of the static padding are never written (they stay 255: no block); k_rtz3 zeroes those rows anyway.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_blocks(const int* __restrict__ cells, const int* __restrict__ tile_grp,
                                                     const int* __restrict__ blk_start, int nblk, const int* __restrict__ gstart,
                                                     const int* __restrict__ s_tile_start, unsigned char* __restrict__ tile_blk) {
    __shared__ int bs[64 + 2];
    for (int i = threadIdx.x; i <= nblk; i += 256) bs[i] = blk_start[i];
    __syncthreads();
    const int n_pos = 16 * bs[nblk];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n_pos; p += gridDim.x * 256) {
        const int c = cells[p];
        if (c < 0) continue;
        const int t = p >> 4;
        int b = 0;
        while (b + 1 < nblk && t >= bs[b + 1]) ++b;
        const int g = tile_grp[t];
        tile_blk[(size_t)16 * s_tile_start[g] + (c - gstart[g])] = (unsigned char)b;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int rtz3_ntb(int dp, int nblk) { return std::max(0, (nblk - (64 - dp) + 15) / 16); }
bool rtz3_ok(int mt, int dp, int nblk, int G) {
    return mt >= 1 && mt <= 7 && (dp == 32 || dp == 52 || dp == 64) && rtz3_ntb(dp, nblk) <= 2 && G <= 64 && nblk <= 64;
}
int rtz3_slab_floats(int mt, int dp, int nblk) { return mt * (4 + rtz3_ntb(dp, nblk)) * 256; }
size_t rtz3_lds_bytes(int Kp, int dp) { return (size_t)2 * RTZ3_WAVES * (16 * (Kp + dp) + 4) * sizeof(float); }

template <int MT, int KS, int NTB>
static void launch_rtz3_t(const Rtz3Args& a, size_t sm, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rtz3<MT, KS, NTB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
        if (getenv("HMX_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_rtz3<MT, KS, NTB>), 64 * RTZ3_WAVES, sm);
            fprintf(stderr, "[hmx] k_rtz3<%d,%d,%d>: %d tasks, %zu bytes of LDS per workgroup, %d workgroups per CU\n", MT, KS, NTB, a.ntasks, sm, nb);
        }
    }
#ifdef HMX_RTZ3_PROF
    static unsigned long long* prof = nullptr;
    static int calls = 0;
    Rtz3Args b = a;
    if (!prof) (void)hipMalloc(reinterpret_cast<void**>(&prof), (size_t)4096 * RTZ3_WAVES * 8 * 8);
    (void)hipMemsetAsync(prof, 0, (size_t)a.ntasks * RTZ3_WAVES * 8 * 8, s);
    b.prof = prof;
    hipLaunchKernelGGL((k_rtz3<MT, KS, NTB>), dim3(a.ntasks), dim3(64 * RTZ3_WAVES), sm, s, b);
    if (++calls == 30) {
        std::vector<unsigned long long> h((size_t)a.ntasks * RTZ3_WAVES * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t_min = ~0ull, t_max = 0;
        double pro = 0, loop = 0, epi = 0, wait = 0, tiles = 0, start_spread_max = 0;
        const size_t nw = (size_t)a.ntasks * RTZ3_WAVES;
        for (size_t w = 0; w < nw; ++w) { t_min = std::min(t_min, h[w * 8]); t_max = std::max(t_max, h[w * 8 + 3]); }
        for (size_t w = 0; w < nw; ++w) {
            const unsigned long long* r = &h[w * 8];
            pro += (double)(r[1] - r[0]); loop += (double)(r[2] - r[1]); epi += (double)(r[3] - r[2]); wait += (double)r[4]; tiles += (double)r[5];
            start_spread_max = std::max(start_spread_max, (double)(r[0] - t_min));
        }
        fprintf(stderr, "[k_rtz3 prof] <%d,%d,%d> %d tasks: kernel span %.0f cycles; per wave: prologue %.0f, loop %.0f (%.0f per tile, %.1f tiles), of which waiting for tiles %.0f (%.0f per tile), epilogue %.0f; latest wave start +%.0f\n",
                MT, KS, NTB, a.ntasks, (double)(t_max - t_min), pro / nw, loop / nw, loop / tiles, tiles / nw, wait / nw, wait / tiles, epi / nw, start_spread_max);
    }
    return;
#endif
    hipLaunchKernelGGL((k_rtz3<MT, KS, NTB>), dim3(a.ntasks), dim3(64 * RTZ3_WAVES), sm, s, a);
}
template <int KS, int NTB>
static void launch_rtz3_m(const Rtz3Args& a, int mt, size_t sm, hipStream_t s) {
    switch (mt) {
        case 1: launch_rtz3_t<1, KS, NTB>(a, sm, s); break;
        case 2: launch_rtz3_t<2, KS, NTB>(a, sm, s); break;
        case 3: launch_rtz3_t<3, KS, NTB>(a, sm, s); break;
        case 4: launch_rtz3_t<4, KS, NTB>(a, sm, s); break;
        case 5: launch_rtz3_t<5, KS, NTB>(a, sm, s); break;
        case 6: launch_rtz3_t<6, KS, NTB>(a, sm, s); break;
        default: launch_rtz3_t<7, KS, NTB>(a, sm, s); break;
    }
}
template <int KS>
static void launch_rtz3_k(const Rtz3Args& a, int mt, int ntb, size_t sm, hipStream_t s) {
    switch (ntb) {
        case 0: launch_rtz3_m<KS, 0>(a, mt, sm, s); break;
        case 1: launch_rtz3_m<KS, 1>(a, mt, sm, s); break;
        default: launch_rtz3_m<KS, 2>(a, mt, sm, s); break;
    }
}

// ---- k_rtz3c (bf16 pipe): at most one extra one-hot tile (140 accumulators), sixteen tile buffers in one CU's LDS
size_t rtz3b_lds_bytes(int Kp, int dp) { return (size_t)4 * RTZ3_WAVES * (16 * (Kp + dp) + 4) * sizeof(float); }
bool rtz3b_ok(int mt, int dp, int nblk, int Kp) {
    return rtz3_ntb(dp, nblk) <= 1 && std::max(rtz3b_lds_bytes(Kp, dp), (size_t)rtz3_slab_floats(mt, dp, nblk) * sizeof(float)) <= 160 * 1024;
}
int rtz3_quad(int mt, int dp, int nblk, int Kp, bool allow_bf16) {   // tiles a workgroup takes side by side (the host cuts the tasks with it): k_rtz3c's eight waves, k_rtz3's four
    return (allow_bf16 && rtz3b_ok(mt, dp, nblk, Kp)) ? 8 : 4;
}
template <int MT, int KS, int NTB>
static void launch_rtz3c_t(const Rtz3Args& a, size_t sm, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rtz3c<MT, KS, NTB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_rtz3c<MT, KS, NTB>), dim3(a.ntasks), dim3(64 * RTZ3C_WAVES), sm, s, a);
}
template <int KS, int NTB>
static void launch_rtz3c_m(const Rtz3Args& a, int mt, size_t sm, hipStream_t s) {
    switch (mt) {
        case 1: launch_rtz3c_t<1, KS, NTB>(a, sm, s); break;
        case 2: launch_rtz3c_t<2, KS, NTB>(a, sm, s); break;
        case 3: launch_rtz3c_t<3, KS, NTB>(a, sm, s); break;
        case 4: launch_rtz3c_t<4, KS, NTB>(a, sm, s); break;
        case 5: launch_rtz3c_t<5, KS, NTB>(a, sm, s); break;
        case 6: launch_rtz3c_t<6, KS, NTB>(a, sm, s); break;
        default: launch_rtz3c_t<7, KS, NTB>(a, sm, s); break;
    }
}
template <int KS>
static void launch_rtz3c_k(const Rtz3Args& a, int mt, int ntb, size_t sm, hipStream_t s) {
    if (ntb == 0) launch_rtz3c_m<KS, 0>(a, mt, sm, s); else launch_rtz3c_m<KS, 1>(a, mt, sm, s);
}

// `nblk` decides the one-hot columns (1 for the ridge / centroid-only passes: column 0 = the plain column sums)
int launch_rtz3(const Rtz3Args& a, int mt, int dp, int nblk, hipStream_t s, bool allow_bf16, int quad) {
    if (!rtz3_ok(mt, dp, nblk, 1) || a.ntasks <= 0) return -1;
    const int ntb = rtz3_ntb(dp, nblk);
    if (allow_bf16 && quad == 8 && rtz3b_ok(mt, dp, nblk, a.Kp)) {   // the tasks were cut for eight waves: k_rtz3c
        const size_t smb = std::max(rtz3b_lds_bytes(a.Kp, dp), (size_t)rtz3_slab_floats(mt, dp, nblk) * sizeof(float));
        switch (dp) {
            case 32: launch_rtz3c_k<8>(a, mt, ntb, smb, s); break;
            case 52: launch_rtz3c_k<13>(a, mt, ntb, smb, s); break;
            default: launch_rtz3c_k<16>(a, mt, ntb, smb, s); break;
        }
        return 1;
    }
    if (quad != 4) return -1;                                       // (tasks cut for eight waves cannot run on k_rtz3's four)
    const size_t sm = std::max(rtz3_lds_bytes(a.Kp, dp), (size_t)rtz3_slab_floats(mt, dp, nblk) * sizeof(float));
    switch (dp) {
        case 32: launch_rtz3_k<8>(a, mt, ntb, sm, s); break;
        case 52: launch_rtz3_k<13>(a, mt, ntb, sm, s); break;
        default: launch_rtz3_k<16>(a, mt, ntb, sm, s); break;
    }
    return 0;
}

void launch_rtz3_finish(const Rtz3FinishArgs& a, hipStream_t s) {
    const size_t sm = (size_t)a.G * 16 * (a.wide ? a.NT : 4 + a.NTB) * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rtz3_finish), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_rtz3_finish, dim3(a.K16), dim3(RTZ3_FIN_THREADS), sm, s, a);
}

// ---- wide shapes (k_rtzw)
int rtzw_nt(int dp, int d, int nblk) { return dp / 16 + std::max(0, (nblk - (dp - d) + 15) / 16); }
bool rtzw_ok(int mt, int dp, int d, int nblk, int G) {
    return mt >= 1 && mt <= 13 && dp % 16 == 0 && dp <= 208 && (mt > 7 || dp > 64) && rtzw_nt(dp, d, nblk) <= 2 * RTZW_WAVES &&
           nblk <= 64 && (size_t)G * 16 * rtzw_nt(dp, d, nblk) * sizeof(double) <= 150 * 1024;
}
int rtzw_slab_floats(int mt, int dp, int d, int nblk) { return mt * rtzw_nt(dp, d, nblk) * 256; }

template <int MT>
static void launch_rtzw_t(const Rtz3Args& a, size_t sm, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rtzw<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
#ifdef HMX_RTZ3_PROF
    static unsigned long long* prof = nullptr;
    static int calls = 0;
    Rtz3Args b = a;
    if (!prof) (void)hipMalloc(reinterpret_cast<void**>(&prof), (size_t)4096 * 8 * 8 * 8);
    (void)hipMemsetAsync(prof, 0, (size_t)a.ntasks * 8 * 8 * 8, s);
    b.prof = prof;
    hipLaunchKernelGGL((k_rtzw<MT>), dim3(a.ntasks), dim3(64 * RTZW_WAVES), sm, s, b);
    if (++calls == 12) {
        std::vector<unsigned long long> h((size_t)a.ntasks * 8 * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, wait = 0, bar = 0, tiles = 0;
        const size_t nw = (size_t)a.ntasks * 8;
        for (size_t w = 0; w < nw; ++w) {
            const unsigned long long* r = &h[w * 8];
            pro += (double)(r[1] - r[0]); loop += (double)(r[2] - r[1]); wait += (double)r[4]; bar += (double)r[6]; tiles += (double)r[5];
        }
        fprintf(stderr, "[k_rtzw prof] <%d> %d tasks: per wave: prologue %.0f, loop %.0f (%.0f per tile, %.1f tiles), waiting for own pieces %.0f per tile, at the barrier %.0f per tile\n",
                MT, a.ntasks, pro / nw, loop / nw, loop / tiles, tiles / nw, wait / tiles, bar / tiles);
    }
    return;
#endif
    hipLaunchKernelGGL((k_rtzw<MT>), dim3(a.ntasks), dim3(64 * RTZW_WAVES), sm, s, a);
}

// ---- Z_cos as bf16 planes in k_rtzw2b's B-fragment order (see the kernel): one 512-byte piece per (static tile, column tile, plane)
size_t rtzw_zf_tile_words(int dp) { return (size_t)(dp >> 4) * 3 * 128; }
bool rtzw2b_zf_ok(int mt, int dp) { return (size_t)4 * (256 * mt + rtzw_zf_tile_words(dp) + 4) * sizeof(float) <= 160 * 1024; }
__global__ __launch_bounds__(256) void k_zplanes(const float* __restrict__ Z, int DP, int n_tiles, const int* __restrict__ tile_grp,
                                                 const int* __restrict__ gstart, const int* __restrict__ s_tile_start, unsigned* __restrict__ Zf) {
    const int NTP = DP >> 4;
    for (int T = blockIdx.x; T < n_tiles; T += gridDim.x) {
        const int g = tile_grp[T];
        const int cell0 = gstart[g] + 16 * (T - s_tile_start[g]);     // static tiles hold consecutive cells of one group
        const int n_live = min(16, gstart[g + 1] - cell0);
        unsigned* out = Zf + (size_t)T * NTP * 3 * 128;
        for (int it = threadIdx.x; it < NTP * 32; it += 256) {
            const int nt = it >> 5, l32 = it & 31, half = l32 >> 4, c16 = l32 & 15;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * half + j;
                x[j] = (r < n_live) ? Z[(size_t)(cell0 + r) * DP + 16 * nt + c16] : 0.f;   // rows past the group's end: zeros
            }
            u32x4 h, m, l;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned hh, mm, ll;
                bf16_split3((f32x2){x[2 * p], x[2 * p + 1]}, hh, mm, ll);
                h[p] = hh; m[p] = mm; l[p] = ll;
            }
            unsigned* o = out + ((size_t)nt * 3 * 32 + l32) * 4;
            *reinterpret_cast<u32x4*>(o) = h;
            *reinterpret_cast<u32x4*>(o + 128) = m;
            *reinterpret_cast<u32x4*>(o + 256) = l;
        }
    }
}
void launch_zplanes(const float* Z, int dp, int n_tiles, const int* tile_grp, const int* gstart, const int* s_tile_start, unsigned* Zf, hipStream_t s) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(k_zplanes, dim3(std::min(n_tiles, 256 * 16)), dim3(256), 0, s, Z, dp, n_tiles, tile_grp, gstart, s_tile_start, Zf);
}

// ---- k_rtzw2b: K > 112, seven to fourteen column tiles; four tile buffers in one CU's LDS
bool rtzw2b_ok(int mt, int dp, int d, int nblk) {
    if (!rtzw_ok(mt, dp, d, nblk, 1) || mt < 8 || mt > 13) return false;
    const int nth = (rtzw_nt(dp, d, nblk) + 1) / 2;
    return nth >= 4 && nth <= 7 && (size_t)4 * (256 * mt + 16 * dp + 4) * sizeof(float) <= 160 * 1024;
}
template <int MT, int NTH, bool ZF>
static void launch_rtzw2b_z(const Rtz3Args& a, size_t sm, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rtzw2b<MT, NTH, ZF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_rtzw2b<MT, NTH, ZF>), dim3(a.ntasks), dim3(64 * RTZWB_WAVES), sm, s, a);
}
template <int MT, int NTH>
static void launch_rtzw2b_t(const Rtz3Args& a, size_t sm, hipStream_t s) {
    if (a.Zf) launch_rtzw2b_z<MT, NTH, true>(a, sm, s); else launch_rtzw2b_z<MT, NTH, false>(a, sm, s);
}
template <int MT>
static void launch_rtzw2b_m(const Rtz3Args& a, int nth, size_t sm, hipStream_t s) {
    switch (nth) {
        case 4: launch_rtzw2b_t<MT, 4>(a, sm, s); break;
        case 5: launch_rtzw2b_t<MT, 5>(a, sm, s); break;
        case 6: launch_rtzw2b_t<MT, 6>(a, sm, s); break;
        default: launch_rtzw2b_t<MT, 7>(a, sm, s); break;
    }
}
static void launch_rtzw2b(const Rtz3Args& a, int mt, hipStream_t s) {
    const int nth = (a.nt + 1) / 2;
    const size_t sm = (size_t)4 * (256 * mt + (a.Zf ? rtzw_zf_tile_words(a.dp) : 16 * a.dp) + 4) * sizeof(float);
    switch (mt) {
        case 8: launch_rtzw2b_m<8>(a, nth, sm, s); break;
        case 9: launch_rtzw2b_m<9>(a, nth, sm, s); break;
        case 10: launch_rtzw2b_m<10>(a, nth, sm, s); break;
        case 11: launch_rtzw2b_m<11>(a, nth, sm, s); break;
        case 12: launch_rtzw2b_m<12>(a, nth, sm, s); break;
        default: launch_rtzw2b_m<13>(a, nth, sm, s); break;
    }
}

// returns 1 when the bf16-pipe kernel (k_rtzw2b) ran, 0 for the f32-input kernel k_rtzw, -1 unsupported
int launch_rtzw(const Rtz3Args& a_in, int mt, int dp, int d, int nblk, hipStream_t s, bool allow_bf16) {
    if (!rtzw_ok(mt, dp, d, nblk, 1) || a_in.ntasks <= 0) return -1;
    Rtz3Args a = a_in;
    a.dp = dp; a.d = d; a.nt = rtzw_nt(dp, d, nblk);
    if (allow_bf16 && rtzw2b_ok(mt, dp, d, nblk)) {
        launch_rtzw2b(a, mt, s);
        return 1;
    }
    const size_t sm = (size_t)RTZW_NBUF * (16 * (a.Kp + dp) + 4) * sizeof(float);
    if (sm > 160 * 1024) return -1;
    switch (mt) {
        case 1: launch_rtzw_t<1>(a, sm, s); break;   case 2: launch_rtzw_t<2>(a, sm, s); break;
        case 3: launch_rtzw_t<3>(a, sm, s); break;   case 4: launch_rtzw_t<4>(a, sm, s); break;
        case 5: launch_rtzw_t<5>(a, sm, s); break;   case 6: launch_rtzw_t<6>(a, sm, s); break;
        case 7: launch_rtzw_t<7>(a, sm, s); break;   case 8: launch_rtzw_t<8>(a, sm, s); break;
        case 9: launch_rtzw_t<9>(a, sm, s); break;   case 10: launch_rtzw_t<10>(a, sm, s); break;
        case 11: launch_rtzw_t<11>(a, sm, s); break; case 12: launch_rtzw_t<12>(a, sm, s); break;
        default: launch_rtzw_t<13>(a, sm, s); break;
    }
    return 0;
}

void launch_tile_blocks(const int* cells, const int* tile_grp, const int* blk_start, int nblk, int64_t n_pos_upper, const int* gstart,
                        const int* s_tile_start, unsigned char* tile_blk, hipStream_t s) {
    const int wgs = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n_pos_upper + 1023) / 1024));
    hipLaunchKernelGGL(k_tile_blocks, dim3(wgs), dim3(256), 0, s, cells, tile_grp, blk_start, nblk, gstart, s_tile_start, tile_blk);
}
