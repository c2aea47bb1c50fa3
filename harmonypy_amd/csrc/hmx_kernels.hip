// hmx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Harmony iteration engine.
//
// Everything here is written for wave64 + v_mfma_f32_16x16x4_f32 (exact fp32 MFMA).
// Data layout in HBM is cell-major: one row per cell (Z: dp floats, R: Kp floats), so a
// cell's operands for the MFMA fragments are contiguous 16..64-byte pieces of its row.
//
// Fragment conventions (cdna_hip_programming.md §3):
//   A operand: lane l holds A[i = l&15][k = l>>4]
//   B operand: lane l holds B[k = l>>4][j = l&15]
//   C/D      : lane l, reg r holds C[row = 4*(l>>4) + r][col = l&15]
// Throughout: c16 = lane & 15, q = lane >> 4.
//
// A *tile* is 16 list positions that share one batch group; lists are padded with -1.
//
// Reference lines are harmonypy/harmony.py (v0.2.0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "hmx_internal.h"
#include "hmx_device.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// Row normalisation: Z_cos = Z / ||Z||_2 per cell (harmony.py:238, 569)
// one wave per 4 rows (16 lanes per row)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_normalize_rows(const float* __restrict__ Z, float* __restrict__ Zc,
                                                        int64_t N, int dp) {
    const int lane16 = threadIdx.x & 15;
    int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= N) return;  // whole 16-lane group leaves together
    const float* z = Z + row * dp;
    float ss = 0.f;
    for (int j = lane16; j < dp; j += 16) ss += z[j] * z[j];
    ss = wave_sum_c16(ss);
    const float nrm = sqrtf(ss);
    float* o = Zc + row * dp;
    for (int j = lane16; j < dp; j += 16) o[j] = z[j] / nrm;
}

// ------------------------------------------------------------------------------------------
// Centroid rows -> unit length (harmony.py:377, 444).  One 64-lane wave per cluster.
// src: K16 x ldy raw sums (float), dst: K16 x ldy (rows >= K and cols >= d stay zero)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_y_normalize(const T* __restrict__ src, float* __restrict__ dst, int K,
                                                    int d, int ldy) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x;
    float ss = 0.f;
    if (k < K)
        for (int j = lane; j < d; j += 64) {
            const float v = (float)src[(size_t)k * ldy + j];   // fp64 sums are rounded to fp32 first (:443)
            ss += v * v;
        }
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    const float nrm = sqrtf(ss);
    for (int j = lane; j < ldy; j += 64) dst[(size_t)k * ldy + j] = (k < K && j < d) ? (float)src[(size_t)k * ldy + j] / nrm : 0.f;
}

// ------------------------------------------------------------------------------------------
// Assignment kernel: distance GEMM + soft assignment (+ diversity penalty)
//   harmony.py:380-385 (PENALTY = false, init_cluster)
//   harmony.py:447, 466-468, 500-503 and the R-dependent sums of :399, :402 (PENALTY = true)
//
// C[cluster][cell] = Y (K16 x d) . Zcos^T: A = Y (rows from L1/L2, 16 B per lane),
// B = Zcos rows gathered straight from HBM in fragment layout (16 B per lane: the four
// q-lanes of a cell read 64 contiguous bytes of its row).  A lane ends up with the
// clusters {16mt + 4q + r} of cell c16, so the softmax needs two xor-shuffles only.
// ------------------------------------------------------------------------------------------
template <int MT, int NT, bool PENALTY>
__global__ __launch_bounds__(256) void k_assign(AssignArgs a) {
    const int lane = threadIdx.x & 63;
    const int c16 = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int mtn = a.mt;  // live cluster tiles (<= MT)

    // contiguous tile range of this wave, in multiples of NT tiles; the block's tile range
    // lives in device memory (it is produced on the device when the update order is)
    const int tile_begin = a.blk_start ? a.blk_start[a.blk] : a.tile_begin;
    const int tile_end = a.blk_start ? a.blk_start[a.blk + 1] : a.tile_end;
    const int ntiles = tile_end - tile_begin;
    int per = (ntiles + nwaves - 1) / nwaves;
    per = ((per + NT - 1) / NT) * NT;
    const int t0 = min(tile_begin + wave * per, tile_end);
    const int t1 = min(t0 + per, tile_end);
    // Every wave of the workgroup walks `per / NT` steps (empty ones included): the centroid columns of a
    // k-step are staged ONCE per workgroup in LDS (coalesced 16-byte loads, double buffered) instead of being
    // fetched from L2 by every wave -- at K = 200, d = 200 the table is 160 KB, far beyond the L1.
    __shared__ __attribute__((aligned(16))) float Ysh[2][16 * MT][20];
    const int tid = threadIdx.x;

    float sg[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * mt + 4 * q + r;
            sg[mt][r] = (mt < mtn && k < a.K) ? a.sigma[k] : 0.f;  // 0 => exp(-2/0) = 0: padded cluster
        }

    float sacc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[mt][r] = 0.f;
    int cur_g = -1;
    double km_acc = 0.0, ent_acc = 0.0;

    const int kb_full = a.dp >> 4;           // full 16-column blocks of a Z row
    const int tail = (a.dp - 16 * kb_full) >> 2;  // remaining 4-column steps

    auto flush = [&](int g) {
        if (g < 0) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt < mtn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = wave_sum_c16(sacc[mt][r]);
                    const int k = 16 * mt + 4 * q + r;
                    if (c16 == 0 && k < a.K && s != 0.f) atomicAdd(&a.S_out[(size_t)g * a.K16 + k], (double)s);
                    sacc[mt][r] = 0.f;
                }
            }
        }
    };

    int stage = 0;
    for (int step = 0; step < per; step += NT) {
        const int t = t0 + step;
        int cell[NT], grp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int tt = t + nt;
            cell[nt] = (tt < t1) ? a.cells[(size_t)tt * 16 + c16] : -1;
            grp[nt] = (tt < t1) ? a.tile_grp[tt] : -1;
        }
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- distance GEMM over the PC dimension --------------------------------------
        // software pipeline over the 16-column k-steps: the centroid columns and the Z_cos pieces of step
        // kb+1 travel (registers) while step kb multiplies from LDS
        constexpr int YPT = (16 * MT * 4 + 255) / 256;       // 16-byte pieces of a centroid chunk per thread
        f32x4 ynext[YPT];
        f32x4 bnext[NT];
        auto fetch_step = [&](int kb) {
#pragma unroll
            for (int p2 = 0; p2 < YPT; ++p2) {
                const int i = tid + 256 * p2;
                const int row = i >> 2, c4 = i & 3;
                ynext[p2] = (i < 16 * mtn * 4) ? ld4(a.Y + (size_t)row * a.ldy + 16 * kb + 4 * c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bnext[nt] = (cell[nt] >= 0) ? ld4(a.Zcos + (size_t)cell[nt] * a.dp + 16 * kb + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        if (kb_full > 0) fetch_step(0);
        for (int kb = 0; kb < kb_full; ++kb) {
            f32x4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = bnext[nt];
#pragma unroll
            for (int p2 = 0; p2 < YPT; ++p2) {
                const int i = tid + 256 * p2;
                if (i < 16 * mtn * 4) st4(&Ysh[stage][i >> 2][4 * (i & 3)], ynext[p2]);
            }
            if (kb + 1 < kb_full) fetch_step(kb + 1);
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < mtn) {
                    const f32x4 ya = ld4(&Ysh[stage][16 * mt + c16][4 * q]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(ya[i], b[nt][i], acc[mt][nt]);
                }
            }
            stage ^= 1;   // the other buffer is free: every wave passed the barrier of the previous k-step
        }
        for (int s = 0; s < tail; ++s) {
            const int col = 16 * kb_full + 4 * s + q;
            float b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = (cell[nt] >= 0) ? a.Zcos[(size_t)cell[nt] * a.dp + col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < mtn) {
                    const float ya = a.Y[(size_t)(16 * mt + c16) * a.ldy + col];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(ya, b[nt], acc[mt][nt]);
                }
            }
        }

        // ---- per tile: softmax, penalty, write-back, running sums ---------------------
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (grp[nt] < 0) continue;  // wave-uniform (also: no tile in this step)
            if (grp[nt] != cur_g) {
                flush(cur_g);
                cur_g = grp[nt];
            }
            const bool live = cell[nt] >= 0;
            float e[MT][4];
            float e1 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < mtn) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dist = 2.f * (1.f - acc[mt][nt][r]);  // :380 / :447
                        const float arg = -dist / sg[mt][r];              // :383 / :466
                        acc[mt][nt][r] = arg;
                        e[mt][r] = expf(arg);                             // :384 / :467
                        e1 += e[mt][r];
                    }
                }
            }
            e1 = wave_sum_q(e1);  // column sum of :385 / :468
            float u1 = 1.f;
            if (PENALTY) {
                const float* rp = a.rp + (size_t)cur_g * a.K16;
                float us = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (mt < mtn) {
                        const f32x4 pw = ld4(rp + 16 * mt + 4 * q);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            e[mt][r] = (e[mt][r] / e1) * pw[r];  // :468 then :500
                            us += e[mt][r];
                        }
                    }
                }
                us = wave_sum_q(us);
                u1 = fmaxf(us, 1e-8f);  // :501-502
            }
            const float log_e1 = logf(e1);
            const float log_u1 = PENALTY ? logf(u1) : 0.f;
            float km = 0.f, ent = 0.f;
            float* rrow = a.R + (size_t)(live ? cell[nt] : 0) * a.Kp;
            const float* lrp = PENALTY ? (a.lrp + (size_t)cur_g * a.K16) : nullptr;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < mtn) {
                    f32x4 lp = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (PENALTY) lp = ld4(lrp + 16 * mt + 4 * q);
                    f32x4 rv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float Rv = PENALTY ? (e[mt][r] / u1) : (e[mt][r] / e1);  // :503 / :385
                        if (!live) Rv = 0.f;
                        rv[r] = Rv;
                        sacc[mt][r] += Rv;
                        if (Rv > 0.f) {
                            const float arg = acc[mt][nt][r];
                            const float logR = arg - log_e1 + lp[r] - log_u1;
                            km += Rv * (-arg * sg[mt][r]);   // R * dist           (:399)
                            ent += sg[mt][r] * (Rv * logR);  // sigma * R log R    (:402)
                        }
                    }
                    const int col = 16 * mt + 4 * q;
                    if (live && col < a.Kp) st4(rrow + col, rv);
                }
            }
            km_acc += (double)km;
            ent_acc += (double)ent;
        }
    }
    flush(cur_g);
    km_acc = wave_sum_all(km_acc);
    ent_acc = wave_sum_all(ent_acc);
    if (lane == 0) {
        atomicAdd(&a.obj[0], km_acc);
        atomicAdd(&a.obj[1], ent_acc);
    }
}


// ------------------------------------------------------------------------------------------
// k_assign_lds: the assignment kernel for shapes whose centroid table fits the LDS
// (K <= 112, d <= 64: every BASELINE config except C5).  Same arithmetic as k_assign; the
// structure is what makes it fast:
//   * 512-thread workgroups; Y, sigma and the penalty tables rp / log rp are staged once per
//     workgroup into LDS, A fragments come from ds_read_b128;
//   * each wave owns a 16-cell tile at a time: it gathers the tile's Z_cos rows (16 B per lane,
//     four lanes cover 64 contiguous bytes of a row) into a private LDS tile, so the k-loop is
//     a short runtime loop without per-lane operand arrays -- ~100 VGPRs, 4-5 waves per SIMD,
//     and one wave's gather hides under the other waves' MFMA / VALU work;
//   * block sums: 16-lane DPP reduction -> fp64 ds_add -> one global fp64 atomic per table
//     entry and workgroup; objective partials: one fp64 atomic per workgroup into 64 slots.
// MT is the exact number of 16-cluster tiles (no guards inside the unrolled loops).
// ------------------------------------------------------------------------------------------
// exp(x) for x <= 0 as 2^(x log2 e): the product is split into a rounded head and an fma tail so
// that the relative error stays ~1 ulp over the whole range of arguments (|x| up to ~100 would
// otherwise lose 2^-24 * |x| log2 e).  v_exp_f32 is the hardware exp2.
__device__ __forceinline__ float fast_exp(float x) {
    const float L2E_HI = 1.44269502162933349609375f;        // float(log2 e)
    const float L2E_LO = 1.925963033500011e-08f;             // log2 e - L2E_HI
    x = fmaxf(x, -120.f);                                    // exp underflows to 0 long before; keeps -inf (padded clusters) finite
    const float th = x * L2E_HI;
    const float tl = fmaf(x, L2E_HI, -th) + x * L2E_LO;      // exact product tail + low part
    const float p = __builtin_amdgcn_exp2f(th);
    return fmaf(p, tl * 0.693147182464599609375f, p);        // 2^(th+tl) ~= 2^th (1 + tl ln 2)
}

#define ASSIGN_WAVES 8
#ifndef HMX_ABL
#define HMX_ABL 0   /* timing experiments only: 1 no block sums, 2 no R store, 4 cheap math, 8 no GEMM, 16 no tile loop, 32 no gather */
#endif

template <int MT, bool PENALTY>
__global__ __launch_bounds__(64 * ASSIGN_WAVES, 4) void k_assign_lds(AssignArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LDY = a.ldy_lds;  // row stride of Ys and of the Z tiles (floats, multiple of 4)
    constexpr int K16 = 16 * MT;
    float* Ys = reinterpret_cast<float*>(smem);                         // K16 x LDY
    float* sig = Ys + (size_t)K16 * LDY;                                // K16 sigma
    float* nis = sig + K16;                                             // K16 -1/sigma (-inf for pads)
    float* rpT = nis + K16;                                             // G x K16
    float* lrpT = rpT + (PENALTY ? (size_t)a.G * K16 : 0);              // G x K16
    double* Sd = reinterpret_cast<double*>(lrpT + (PENALTY ? (size_t)a.G * K16 : 0));  // G x K16
    float* Zt_all = reinterpret_cast<float*>(Sd + (size_t)a.G * K16);   // waves x 16 x LDY
    double* objw = reinterpret_cast<double*>(Zt_all + (size_t)ASSIGN_WAVES * 16 * LDY);  // waves x 2

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int c4n = a.dp >> 2;
    const int kb_full = a.dp >> 4;
    const int tail = c4n - 4 * kb_full;
    float* Zt = Zt_all + (size_t)wv * 16 * LDY;

    // one tile per wave: the grid covers (an upper bound of) the block's tiles
    const int tile_begin = a.blk_start ? a.blk_start[a.blk] : a.tile_begin;
    const int tile_end = a.blk_start ? a.blk_start[a.blk + 1] : a.tile_end;
    const int t = tile_begin + blockIdx.x * ASSIGN_WAVES + wv;
    const bool has_tile = t < tile_end;                                 // wave-uniform

    // the tile's cell ids and Z_cos rows are requested first: their latency overlaps the table fill
    const int cell = has_tile ? a.cells[(size_t)t * 16 + c16] : -1;
    const bool live = cell >= 0;
    f32x4 zrow[4];   // d <= 64: at most 4 pieces of 16 B per lane
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c4 = min(q + 4 * j, c4n - 1);
        zrow[j] = ld4(a.Zcos + (size_t)(live ? cell : 0) * a.dp + 4 * c4);
    }
    const int grp = has_tile ? __builtin_amdgcn_readfirstlane(a.tile_grp[t]) : 0;

    // ---- stage the tables -------------------------------------------------------------------
    for (int i = tid; i < K16 * c4n; i += 64 * ASSIGN_WAVES) {
        const int row = i / c4n, c4 = i - row * c4n;
        st4(Ys + (size_t)row * LDY + 4 * c4, ld4(a.Y + (size_t)row * a.ldy + 4 * c4));
    }
    for (int i = tid; i < K16; i += 64 * ASSIGN_WAVES) {
        const float sgm = (i < a.K) ? a.sigma[i] : 0.f;
        sig[i] = sgm;
        nis[i] = (i < a.K) ? -1.0f / sgm : -INFINITY;   // pads: exp(2 * -inf) = 0
    }
    {
        const int gk = a.G * K16;
        if (PENALTY && a.tables_in_lds)
            for (int i = tid; i < gk; i += 64 * ASSIGN_WAVES) {
                rpT[i] = a.rp[i];
                lrpT[i] = a.lrp[i];
            }
        if (a.tables_in_lds)
            for (int i = tid; i < gk; i += 64 * ASSIGN_WAVES) Sd[i] = 0.0;
    }
    // the wave's Z tile (dead lanes hold zeros)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c4 = q + 4 * j;
        if (c4 < c4n) st4(Zt + (size_t)c16 * LDY + 4 * c4, live ? zrow[j] : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    __syncthreads();

    double km_d = 0.0, ent_d = 0.0;
    if (has_tile && !(HMX_ABL & 16)) {
        // ---- distance GEMM: C[cluster][cell] = Y . Zcos^T over the PC dimension ------------
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < ((HMX_ABL & 8) ? 0 : kb_full); ++kb) {
            const f32x4 zb = ld4(Zt + (size_t)c16 * LDY + 16 * kb + 4 * q);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 ya = ld4(Ys + (size_t)(16 * mt + c16) * LDY + 16 * kb + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mt] = MFMA16(ya[i], zb[i], acc[mt]);
            }
        }
        for (int s = 0; s < ((HMX_ABL & 8) ? 0 : tail); ++s) {
            const int col = 16 * kb_full + 4 * s + q;
            const float zb = Zt[(size_t)c16 * LDY + col];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA16(Ys[(size_t)(16 * mt + c16) * LDY + col], zb, acc[mt]);
        }

        // ---- softmax over clusters (a lane holds clusters 16mt + 4q + r of cell c16) --------
        // Divisions of the reference (:383, :385, :468, :503) are evaluated as products with
        // correctly rounded reciprocals (<= 1 ulp apart; tests/test_parity_gpu.py pins the effect).
        // Only the exponent arguments stay in registers; exp() is re-evaluated per pass (5 VALU
        // ops) instead of holding a second K-sized array per lane: 4 waves per SIMD.
        float e1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 ni = ld4(nis + 16 * mt + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dist = 2.f * (1.f - acc[mt][r]);   // :380 / :447
                const float arg = dist * ni[r];                // :383 / :466   -dist / sigma
                acc[mt][r] = arg;
                e1 += (HMX_ABL & 4) ? arg * 0.001f + 1.0f : fast_exp(arg);   // :384 / :467
            }
        }
        e1 = wave_sum_q(e1);                                   // column sum of :385 / :468
        const float inv_e1 = 1.0f / e1;
        const float* rp = nullptr;
        float u1 = 1.f;
        if (PENALTY) {
            rp = a.tables_in_lds ? (rpT + (size_t)grp * K16) : (a.rp + (size_t)grp * K16);
            float us = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 pw = ld4(rp + 16 * mt + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ex = (HMX_ABL & 4) ? acc[mt][r] * 0.001f + 1.0f : fast_exp(acc[mt][r]);
                    us += (ex * inv_e1) * pw[r];               // :468 then :500
                }
            }
            us = wave_sum_q(us);
            u1 = fmaxf(us, 1e-8f);                             // :501-502
        }
        const float log_den = logf(e1) + (PENALTY ? logf(u1) : 0.f);
        const float inv_u1 = PENALTY ? 1.0f / u1 : 1.0f;
        float km = 0.f, ent = 0.f;
        float* rrow = a.R + (size_t)(live ? cell : 0) * a.Kp;
        const float* lrp = PENALTY ? (a.tables_in_lds ? (lrpT + (size_t)grp * K16) : (a.lrp + (size_t)grp * K16)) : nullptr;
        double* sdst = a.tables_in_lds ? (Sd + (size_t)grp * K16) : (a.S_out + (size_t)grp * K16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 sg = ld4(sig + 16 * mt + 4 * q);
            f32x4 lp = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 pw = (f32x4){1.f, 1.f, 1.f, 1.f};
            if (PENALTY) {
                lp = ld4(lrp + 16 * mt + 4 * q);
                pw = ld4(rp + 16 * mt + 4 * q);
            }
            f32x4 rv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float arg = acc[mt][r];
                const float ex = (HMX_ABL & 4) ? arg * 0.001f + 1.0f : fast_exp(arg);
                float Rv = PENALTY ? ((ex * inv_e1) * pw[r]) * inv_u1 : ex * inv_e1;   // :503 / :385
                Rv = live ? Rv : 0.f;
                rv[r] = Rv;
                const float logR = arg - log_den + lp[r];
                const bool pos = Rv > 0.f;
                km += pos ? Rv * (-arg * sg[r]) : 0.f;         // R * dist         (:399)
                ent += pos ? sg[r] * (Rv * logR) : 0.f;        // sigma R log R    (:402)
            }
            const int col = 16 * mt + 4 * q;
            if (live && col < a.Kp && !(HMX_ABL & 2)) st4(rrow + col, rv);
            // block sums of the new R (:506-507): 16-cell DPP reduction, one fp64 add per (group, cluster)
            if (!(HMX_ABL & 1)) {
                f32x4 ss;
#pragma unroll
                for (int r = 0; r < 4; ++r) ss[r] = row16_sum(rv[r]);
                if (c16 == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(sdst + col + r, (double)ss[r]);
                }
            }
        }
        km_d = wave_sum_all((double)km);
        ent_d = wave_sum_all((double)ent);
    }
    if (lane == 0) {
        objw[2 * wv] = km_d;
        objw[2 * wv + 1] = ent_d;
    }
    __syncthreads();
    if (tid < 2) {
        double v = 0.0;
        for (int w = 0; w < ASSIGN_WAVES; ++w) v += objw[2 * w + tid];
        if (v != 0.0) atomicAdd(&a.obj[2 * (blockIdx.x & (HMX_OBJ_SLOTS - 1)) + tid], v);
    }
    if (a.tables_in_lds) {
        const int gk = a.G * K16;
        for (int i = tid; i < gk; i += 64 * ASSIGN_WAVES) {
            const double v = Sd[i];
            if (v != 0.0) atomicAdd(&a.S_out[i], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_round: one whole update_R sweep (harmony.py:464-513, all blocks) in ONE persistent launch.
//
// The blocks of a sweep are strictly sequential -- block b needs O and E with the new sums of
// block b-1 put back (harmony.py:491-507) -- but only through a K x B table; everything else a
// cell needs (its Z_cos row, the distance GEMM against Y, exp and the first normalisation,
// harmony.py:447, 466-468) does not depend on that table.  So:
//   * one workgroup of 8 waves per CU stays resident for the whole sweep; Y, sigma live in LDS;
//   * block b's tiles are dealt out two per wave -- round-robin over all workgroups (classic map), or every workgroup from
//     the run of its OWN batch group (group-affine map, round 6: see the comment at the top of the kernel body); the
//     table-independent half of a tile ("pre": distance MFMAs; the exponent arguments stay in
//     4*MT registers per tile) runs BEFORE the wave waits for block b-1 to complete, i.e. it
//     overlaps the grid-wide hand-off;
//   * the tiles' Z_cos rows travel global -> LDS directly (landing zones, no destination
//     registers), requested a block ahead from inside the "pre" phase (tile_step): anything in
//     flight in front of the poll or of the hand-off's loads would be waited for by them, and
//     rows held in registers across "post" put the kernel into scratch;
//   * a wave's own coordinates (lane, c16, q, tid) are refreshed through an empty asm at the top
//     of every block, so nothing derived from them is hoisted out of the sweep and spilled;
//   * hand-off (classic map; the group-affine map exchanges self-validating fixed-point words instead: no returning
//     atomics, no counter): every workgroup adds its block sums to one of HMX_ROUND_SLOTS fp64 tables with
//     agent-scope atomics, drains them (s_waitcnt vmcnt(0)), and one lane bumps an arrival
//     counter; consumers poll the counter with relaxed agent-scope loads and read the tables
//     with agent-scope atomic loads -- 8-byte agent atomics on both sides, so no cache
//     write-back / invalidate is needed (cdna_hip_programming.md, Guideline 16);
//   * every workgroup then rebuilds the block's diversity table itself (k_block_table's
//     arithmetic: O, E = T Pr_b, clamp, pow, log) in LDS and finishes its tiles ("post":
//     penalty, renormalisation, R row store, new block sums, objective terms).
// Waits are bounded: a workgroup that is not resident would otherwise hang the grid; on timeout
// a.error is set and the host reports it.
// ------------------------------------------------------------------------------------------
#ifndef HMX_RABL
#define HMX_RABL 0   /* timing experiments only: 1 no block sums, 2 no R store, 4 no exp in round_post */
#endif
#ifdef HMX_ROUND_PROF   /* timing experiments only: per-workgroup phase stamps (s_memtime) */
#define RSTAMP(slot) do { if (tid == 0 && a.prof) a.prof[((size_t)wg * a.nblk + b) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define TSTAMP(slot) do { if (tid == 0 && a.prof && prof_b >= 0) a.prof[((size_t)wg * a.nblk + prof_b) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RSTAMP(slot) do { } while (0)
#define TSTAMP(slot) do { } while (0)
#endif
// s_waitcnt vmcnt(0) as the builtin (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15), so that the compiler's own wait
// insertion knows nothing is outstanding afterwards; as inline assembly it is invisible to it and every later
// wait it computes assumes the older operations are still in flight
#define WAIT_VMEM_ALL() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)
#define ROUND_WAVES 8   /* 2 waves per SIMD: 256 registers per lane, two tiles in flight per wave */
#define ROUND_TPW 2     /* tiles a wave carries across the hand-off */
#define ROUND_THREADS (64 * ROUND_WAVES)

// KS = row length of Z_cos in 4-float k-steps (dp = 4 KS).  Lane (c16, q) takes the 16-byte pieces
// q, q+4, .. of its cell's row -- piece j is exactly the B operand of the four k-steps of column
// block j -- plus one float per tail k-step.  The matching A operands come from the centroid
// table in LDS.
template <int KS>
struct RoundZ {
    static constexpr int NF = KS / 4, NT = KS % 4;
    f32x4 zp[NF > 0 ? NF : 1];
    float zt[NT > 0 ? NT : 1];
};
template <int MT>
struct RoundTile {
    f32x4 arg[MT];   // -dist / sigma (:447, :466); round_post_pass1 turns it into exp(arg) * ratio^theta in place
    int cell, grp;
};

// issue the loads of a tile's Z_cos rows.  Dead lanes (cell < 0: list padding, or no tile at
// all) read cell 0's row: their results are finite and are multiplied by zero later.
template <int KS>
__device__ __forceinline__ void round_issue_z(const float* __restrict__ Zcos, int cell, int q, RoundZ<KS>& Z) {
    constexpr int NF = KS / 4, NT = KS % 4;
    const float* zr = Zcos + (size_t)(cell >= 0 ? cell : 0) * (4 * KS);
#pragma unroll
    for (int j = 0; j < NF; ++j) Z.zp[j] = ld4(zr + 16 * j + 4 * q);
#pragma unroll
    for (int s = 0; s < NT; ++s) Z.zt[s] = zr[16 * NF + 4 * s + q];
}

// table-independent half of a tile: distance GEMM against the LDS-resident centroids -> exponent arguments.
// LOG2 (k_round): the centroid rows in LDS are pre-scaled by c_k = 2 log2(e) / sigma_k and `nis` holds -c_k, so the
// accumulators START at -c_k and the products land directly on  -dist / sigma * log2(e) = c_k (y.z - 1)  -- the argument of
// the hardware exp2 (:447, :466) -- and the three VALU operations per entry of the conversion are gone.  The rounding of
// the scaled dot product (magnitudes up to c_k ~ 29: 2e-6 absolute in the argument) is what 2 (1 - y.z) / sigma carried
// already (the rounding of y.z, 1e-7, times 2 / sigma).
#ifndef HMX_ROUND_EXP2
#define HMX_ROUND_EXP2 1
#endif
template <int MT, int KS, bool LOG2 = false>
__device__ __forceinline__ void round_compute(const float* Ys, const float* nis, int LDY, int c16, int q,
                                              const RoundZ<KS>& Z, RoundTile<MT>& T) {
    constexpr int NF = KS / 4, NT = KS % 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) T.arg[mt] = LOG2 ? ld4(nis + 16 * mt + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NF; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 ya = ld4(Ys + (size_t)(16 * mt + c16) * LDY + 16 * j + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) T.arg[mt] = MFMA16(ya[i], Z.zp[j][i], T.arg[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the centroid fragments of one column block live at a time
    }
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int col = 16 * NF + 4 * s + q;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) T.arg[mt] = MFMA16(Ys[(size_t)(16 * mt + c16) * LDY + col], Z.zt[s], T.arg[mt]);
    }
    if (!LOG2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 ni = ld4(nis + 16 * mt + 4 * q);
            const f32x4 one = (f32x4){1.f, 1.f, 1.f, 1.f};
            T.arg[mt] = (2.f * (one - T.arg[mt])) * ni;          // dist = 2 (1 - Y.Z) (:447), arg = -dist / sigma (:466)
        }
    }
}

// ---- the same distance GEMM on the bf16 matrix pipe, fp32 operands as three bf16 terms (hmx_device.h) -------------
// A k-step is 32 row floats = 8 pieces of 16 bytes; lane (c16, q) supplies pieces 8 s + 2 q and 8 s + 2 q + 1 of step s
// (pieces past the row are zero on both sides).  The centroid table sits in LDS as three bf16 planes (h, m, l), rows
// of 16 NS + 4 registers (the odd multiple of 4 spreads a 16-lane group's 16-byte reads over all 64 banks); the cells' rows
// are split in registers when their fragments are read -- 9 vector instructions per pair of values, next to six
// 16-cycle MFMAs per cluster tile and k-step where the f32-input form took eight 32-cycle ones.
#ifndef HMX_ROUND_BF3
#define HMX_ROUND_BF3 1
#endif
constexpr int bf3_steps(int KS) { return (KS + 7) / 8; }
constexpr int bf3_ldb(int KS) { return 16 * bf3_steps(KS) + 4; }   // row stride of a plane, in 32-bit words
// Row requests of the bf16-pipe instances: one LDS-DMA request brings WHOLE rows -- 64 / KS of them, lane l the piece l % KS
// of row l / KS, both fixed per lane -- so that the only per-request work is the row id (one ds_bpermute with an immediate
// offset) and one 64-bit multiply-add; the dense form of the f32-input instances (request i brings pieces 64 i .. of the
// tile's 16 KS) costs ~25 instructions per request.  Consecutive requests land back to back, so the zone is the tile
// row-major either way (rows of 52 words: the 16 rows' 16-byte fragment reads of a lane group cover all 64 banks).
constexpr int zone_rows_per_request(int KS) { return 64 / KS; }
template <int KS>
struct RoundZ3 {
    static constexpr int NS = bf3_steps(KS);
    u32x4 h[NS], m[NS], l[NS];
};
// raw pieces (in the order s, half) -> split fragments
template <int KS>
__device__ __forceinline__ void round_split_rows(const f32x4 (&raw)[2 * bf3_steps(KS)], RoundZ3<KS>& Z) {
    constexpr int NS = bf3_steps(KS);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 v = raw[2 * s + hh];
            unsigned h0, m0, l0, h1, m1, l1;
            bf16_split3((f32x2){v[0], v[1]}, h0, m0, l0);
            bf16_split3((f32x2){v[2], v[3]}, h1, m1, l1);
            Z.h[s][2 * hh] = h0; Z.h[s][2 * hh + 1] = h1;
            Z.m[s][2 * hh] = m0; Z.m[s][2 * hh + 1] = m1;
            Z.l[s][2 * hh] = l0; Z.l[s][2 * hh + 1] = l1;
        }
}
template <int KS>
__device__ __forceinline__ void round_split_step(const f32x4 (&raw)[2 * bf3_steps(KS)], int s, RoundZ3<KS>& Z) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const f32x4 v = raw[2 * s + hh];
        unsigned h0, m0, l0, h1, m1, l1;
        bf16_split3((f32x2){v[0], v[1]}, h0, m0, l0);
        bf16_split3((f32x2){v[2], v[3]}, h1, m1, l1);
        Z.h[s][2 * hh] = h0; Z.h[s][2 * hh + 1] = h1;
        Z.m[s][2 * hh] = m0; Z.m[s][2 * hh + 1] = m1;
        Z.l[s][2 * hh] = l0; Z.l[s][2 * hh + 1] = l1;
    }
}
// the lane's raw pieces of a row at `zr` (LDS landing zone or global memory); reads past the row are clamped and zeroed
template <int KS>
__device__ __forceinline__ void round_raw_pieces(const float* zr, int q, f32x4 (&raw)[2 * bf3_steps(KS)]) {
    constexpr int NS = bf3_steps(KS);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int p = 8 * s + 2 * q + hh;
            if (8 * s + 7 < KS) {
                raw[2 * s + hh] = ld4(zr + 4 * p);
            } else {
                const f32x4 v = ld4(zr + 4 * min(p, KS - 1));
                raw[2 * s + hh] = p < KS ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
}
// Yb: the three planes, K16 x bf3_ldb(KS) words each; nis: the accumulators' start (-c_k), see round_compute<.., true>
template <int MT, int KS>
__device__ __forceinline__ void round_compute_bf3(const unsigned* Yb, const float* nis, int c16, int q,
                                                  const RoundZ3<KS>& Z, RoundTile<MT>& T) {
    constexpr int NS = bf3_steps(KS), LDB = bf3_ldb(KS), PL = 16 * MT * LDB;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) T.arg[mt] = ld4(nis + 16 * mt + 4 * q);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // cluster tiles two at a time (three for the odd one out): consecutive MFMAs never share an accumulator
#pragma unroll
        for (int g = 0; g < (MT >= 2 ? MT / 2 : 1); ++g) {
            const int mt0 = 2 * g;
            const int n = (g == (MT >= 2 ? MT / 2 : 1) - 1) ? MT - mt0 : 2;
            u32x4 yh[3], ym[3], yl[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < n) {
                    const unsigned* r = Yb + (size_t)(16 * (mt0 + i) + c16) * LDB + 16 * s + 4 * q;
                    yh[i] = ld4u(r);
                    ym[i] = ld4u(r + PL);
                    yl[i] = ld4u(r + 2 * PL);
                }
            // smallest terms first
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < n) T.arg[mt0 + i] = MFMA_BF16(yl[i], Z.h[s], T.arg[mt0 + i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < n) T.arg[mt0 + i] = MFMA_BF16(yh[i], Z.l[s], T.arg[mt0 + i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < n) T.arg[mt0 + i] = MFMA_BF16(ym[i], Z.m[s], T.arg[mt0 + i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < n) T.arg[mt0 + i] = MFMA_BF16(ym[i], Z.h[s], T.arg[mt0 + i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < n) T.arg[mt0 + i] = MFMA_BF16(yh[i], Z.m[s], T.arg[mt0 + i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < n) T.arg[mt0 + i] = MFMA_BF16(yh[i], Z.h[s], T.arg[mt0 + i]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one k-step's centroid fragments live at a time
    }
}

// two tiles against one read of the centroid fragments (half the LDS traffic per MFMA, and every pair of consecutive
// MFMAs has different accumulators whatever the number of cluster tiles); `mid()` runs between the k-steps 0 and 1
template <int MT, int KS, typename F, typename F2>
__device__ __forceinline__ void round_compute_bf3_pair(const unsigned* Yb, const float* nis, int c16, int q,
                                                       const f32x4 (&raw0)[2 * bf3_steps(KS)], const f32x4 (&raw1)[2 * bf3_steps(KS)],
                                                       RoundTile<MT>& T0, RoundTile<MT>& T1, F&& mid, F2&& ready) {
    constexpr int NS = bf3_steps(KS), LDB = bf3_ldb(KS), PL = 16 * MT * LDB;
    RoundZ3<KS> Z0, Z1;
    round_split_step<KS>(raw0, 0, Z0);
    round_split_step<KS>(raw1, 0, Z1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) T1.arg[mt] = T0.arg[mt] = ld4(nis + 16 * mt + 4 * q);
    __builtin_amdgcn_sched_barrier(0);
    ready();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // (the split of the next k-step's pieces between this k-step's MFMAs -- sched_group_barrier, the pieces pinned into the
        // region by empty volatile statements -- measured no different: 302 vs 303 us; the partner wave fills the gaps)
        if (s + 1 < NS) {
            round_split_step<KS>(raw0, s + 1, Z0);
            round_split_step<KS>(raw1, s + 1, Z1);
        }
        // centroid fragments one cluster tile ahead of their MFMAs (read in the order of use: l, h, m)
        u32x4 ya[2][3];
        auto read_y = [&](int mt, u32x4 (&y)[3]) {
            const unsigned* r = Yb + (size_t)(16 * mt + c16) * LDB + 16 * s + 4 * q;
            y[0] = ld4u(r + 2 * PL);
            y[1] = ld4u(r);
            y[2] = ld4u(r + PL);
        };
        read_y(0, ya[0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt + 1 < MT) read_y(mt + 1, ya[(mt + 1) & 1]);
            const u32x4 yl = ya[mt & 1][0], yh = ya[mt & 1][1], ym = ya[mt & 1][2];
            T0.arg[mt] = MFMA_BF16(yl, Z0.h[s], T0.arg[mt]);   // smallest terms first
            T1.arg[mt] = MFMA_BF16(yl, Z1.h[s], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(yh, Z0.l[s], T0.arg[mt]);
            T1.arg[mt] = MFMA_BF16(yh, Z1.l[s], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(ym, Z0.m[s], T0.arg[mt]);
            T1.arg[mt] = MFMA_BF16(ym, Z1.m[s], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(ym, Z0.h[s], T0.arg[mt]);
            T1.arg[mt] = MFMA_BF16(ym, Z1.h[s], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(yh, Z0.m[s], T0.arg[mt]);
            T1.arg[mt] = MFMA_BF16(yh, Z1.m[s], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(yh, Z0.h[s], T0.arg[mt]);
            T1.arg[mt] = MFMA_BF16(yh, Z1.h[s], T1.arg[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) {
            mid();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// a tile's Z_cos fragments from its landing zone in LDS (16 rows x 4 KS floats, row-major)
template <int KS>
__device__ __forceinline__ void round_rows_from_lds(const float* zt, int c16, int q, RoundZ<KS>& Z) {
    constexpr int NF = KS / 4, NT = KS % 4;
    const float* zr = zt + c16 * (4 * KS);
#pragma unroll
    for (int j = 0; j < NF; ++j) Z.zp[j] = ld4(zr + 16 * j + 4 * q);
#pragma unroll
    for (int s = 0; s < NT; ++s) Z.zt[s] = zr[16 * NF + 4 * s + q];
}

// table-dependent half: exp, penalty, renormalisation, R row, block sums, objective terms.
// With ex = exp(arg), e1 = sum ex (:467-468), t = ex * ratio^theta (:500) and U = sum t:
//   R = (t / e1) / max(U / e1, 1e-8) = t / max(U, 1e-8 e1)                         (:501-503)
//   sum R dist          = -scl sum t sigma arg,                  scl = 1 / max(U, 1e-8 e1)   (:399, dist = -arg sigma)
//   sum sigma R log R   =  scl [sum t sigma arg + sum t sigma log(ratio^theta) - log(max(U, 1e-8 e1)) sum t sigma]   (:402)
// so ONE exp per entry and one pass over the exponent arguments yields everything; t replaces
// arg in place.  The arithmetic is written on pairs so that it maps to v_pk_* instructions.
// Padded clusters carry arg = -120 (exp underflows to an exact 0) and sigma 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LOG2: the arguments are in log2 units (round_compute<.., true>): one v_exp_f32 per entry, the sum of t sigma arg is scaled
// by ln 2 once per tile.  A2TAB: the term  sum t sigma log(ratio^theta)  is NOT accumulated here -- summed over the cells of a
// block it equals  sum_(g,k) log(ratio^theta)[g][k] sigma_k S[g][k]  with S the block sums of the new R that the caller
// keeps anyway (k_round adds it when it publishes them): one fma per entry and the log table's reads are gone.
#ifndef HMX_ROUND_A2TAB
#define HMX_ROUND_A2TAB 1
#endif
#ifndef HMX_ROUND_PK
#define HMX_ROUND_PK 1
#endif
template <int MT, bool PENALTY = true, bool LOG2 = false, bool A2TAB = false, bool PK = (HMX_ROUND_PK != 0)>
__device__ __forceinline__ void round_post_pass1(const float* sig, const float* rpT, const float* lrpT, int q,
                                                 RoundTile<MT>& T, float& scl, double& km_acc, double& ent_acc) {
    constexpr int K16 = 16 * MT;
    const float* rp = PENALTY ? rpT + (size_t)T.grp * K16 : nullptr;
    const float* lrp = PENALTY ? lrpT + (size_t)T.grp * K16 : nullptr;
    float e1 = 0.f, us = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (PK && LOG2 && PENALTY) {
        // the same arithmetic on pairs of entries: v_pk_mul / v_pk_add / v_pk_fma do two fp32 operations per lane and
        // issue slot, so an entry costs one v_exp_f32 + three packed operations instead of one + seven
        f32x2 e1v = {0.f, 0.f}, usv = {0.f, 0.f}, a1v = {0.f, 0.f}, a2v = {0.f, 0.f}, a3v = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 pw = ld4(rp + 16 * mt + 4 * q);
            const f32x4 lp = !A2TAB ? ld4(lrp + 16 * mt + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 sg = ld4(sig + 16 * mt + 4 * q);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 arg = h ? T.arg[mt].zw : T.arg[mt].xy;
                f32x2 ex;
                ex.x = __builtin_amdgcn_exp2f(arg.x);                // :467
                ex.y = __builtin_amdgcn_exp2f(arg.y);
                e1v += ex;
                const f32x2 t = ex * (h ? pw.zw : pw.xy);            // :500
                const f32x2 ts = t * (h ? sg.zw : sg.xy);
                usv += t;
                a1v = __builtin_elementwise_fma(ts, arg, a1v);
                if (!A2TAB) a2v = __builtin_elementwise_fma(ts, h ? lp.zw : lp.xy, a2v);
                a3v += ts;
                if (h) T.arg[mt].zw = t; else T.arg[mt].xy = t;
            }
        }
        e1 = e1v.x + e1v.y; us = usv.x + usv.y; a1 = a1v.x + a1v.y; a2 = a2v.x + a2v.y; a3 = a3v.x + a3v.y;
    } else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 pw = PENALTY ? ld4(rp + 16 * mt + 4 * q) : (f32x4){1.f, 1.f, 1.f, 1.f};   // no penalty: init_cluster (:383-385)
        const f32x4 lp = (PENALTY && !A2TAB) ? ld4(lrp + 16 * mt + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 sg = ld4(sig + 16 * mt + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float arg = T.arg[mt][r];
#if HMX_RABL & 4
            const float ex = arg * 0.001f + 1.0f;
#else
            const float ex = LOG2 ? __builtin_amdgcn_exp2f(arg) : fast_exp_finite(arg);           // :467
#endif
            e1 += ex;
            const float t = ex * pw[r];                      // :500 (the 1/e1 of :468 cancels, see above)
            const float ts = t * sg[r];
            us += t;
            a1 = fmaf(ts, arg, a1);
            if (!A2TAB) a2 = fmaf(ts, lp[r], a2);
            a3 += ts;
            T.arg[mt][r] = t;
        }
    }
    e1 = wave_sum_q(e1);                                   // column sum of :468
    us = wave_sum_q(us);
    if (LOG2) a1 *= 0.693147182464599609375f;
    const float den = fmaxf(us, 1e-8f * e1);               // e1 * max(sum R_new, 1e-8)   (:501-502)
    scl = (T.cell >= 0) ? 1.0f / den : 0.f;                // dead lanes (list padding) contribute exact zeros
    km_acc -= (double)(scl * a1);
    ent_acc += (double)(scl * (a1 + a2 - logf(den) * a3));
}

// second pass for the two tiles a wave carries: R rows out, sums over the 16 cells of a tile into
// the block's (group, cluster) table.  Tiles of one group (the usual case: a wave's tiles are
// neighbours in the block's group-sorted list) share one cross-lane reduction.
#ifndef HMX_ROUND_RETURNING
/* 1 (default): the slot atomics of k_round (and the peer-box writes of its gateway workgroup) RETURN their old value, so the
   wave's vmcnt(0) before the arrival means "performed", not "accepted".  The persistent wide sweep of round 3 showed what the difference
   can be: a flag raised behind non-returning atomics or stores was seen by other XCDs before some of the data.  k_round queues
   at most two atomics per thread and never showed it in any parity run, but "never observed" is not an ordering guarantee.
   Measured cost at C3: 3.67 -> 3.75 ms of sweeps per Harmony iteration (+2 %). */
#define HMX_ROUND_RETURNING 1
#endif
// Sums of NV = 4 MT per-lane values over the 16 lanes of each DPP row (the 16 cells of a tile; a row = one q), delivered
// SCATTERED: lane c16 ends up with the total of value bitrev4(c16) of each group of 16 values.  Four butterfly stages; in
// stage s a lane keeps one value of every pair and hands the other to its partner -- row_mirror / row_half_mirror /
// quad_perm [3,2,1,0] / quad_perm [1,0,3,2], each keeping the bits the earlier stages decided on -- so 16 values cost
// 8 + 4 + 2 + 1 pairs x (2 selects + 1 DPP add) = 45 operations instead of 16 x 5 for sixteen full row sums, and the
// result is one value per lane: ONE conversion and ONE fp64 LDS add per group of 16 clusters and wave instead of 16 x 4.
#ifndef HMX_ROUND_RS
#define HMX_ROUND_RS 1
#endif
#ifndef HMX_ROUND_R_NT
#define HMX_ROUND_R_NT 0
#endif
#ifndef HMX_ROUND_PUBWAVE
#define HMX_ROUND_PUBWAVE 1
#endif
template <int N>
__device__ __forceinline__ float rs16(const float (&v)[N], int c16) {   // N <= 16 live values, the rest count as zero
    static_assert(N >= 1 && N <= 16, "group of 16");
    float a[8], b4[4], c2[2];
    const bool k3 = (c16 & 8) != 0, k2 = (c16 & 4) != 0, k1 = (c16 & 2) != 0, k0 = (c16 & 1) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float A = 2 * j < N ? v[2 * j] : 0.f, B = 2 * j + 1 < N ? v[2 * j + 1] : 0.f;
        if (2 * j >= N) { a[j] = 0.f; continue; }
        const float keep = k3 ? B : A, send = k3 ? A : B;
        a[j] = keep + DPP_F(send, 0x140);     // row_mirror: lane 15 - l
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * j >= N) { b4[j] = 0.f; continue; }
        const float keep = k2 ? a[2 * j + 1] : a[2 * j], send = k2 ? a[2 * j] : a[2 * j + 1];
        b4[j] = keep + DPP_F(send, 0x141);    // row_half_mirror: lane l ^ 7
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (8 * j >= N) { c2[j] = 0.f; continue; }
        const float keep = k1 ? b4[2 * j + 1] : b4[2 * j], send = k1 ? b4[2 * j] : b4[2 * j + 1];
        c2[j] = keep + DPP_F(send, 0x1B);     // quad_perm [3,2,1,0]: lane l ^ 3
    }
    const float keep = k0 ? c2[1] : c2[0], send = k0 ? c2[0] : c2[1];
    return keep + DPP_F(send, 0xB1);          // quad_perm [1,0,3,2]: lane l ^ 1
}
// one group of up to four cluster tiles (16 values per lane): scatter-reduce over the tile's cells, one fp64 LDS add per lane
template <int NM>
__device__ __forceinline__ void block_sums_rs(const f32x4 (&sm)[NM], double* sd, int c16, int q) {
    const int i = ((c16 & 1) << 3) | ((c16 & 2) << 1) | ((c16 & 4) >> 1) | ((c16 & 8) >> 3);   // the value this lane ends up with
    float v[4 * NM];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * m + r] = sm[m][r];
    const float tot = rs16<4 * NM>(v, c16);
    if (i < 4 * NM) atomicAdd(sd + 16 * (i >> 2) + 4 * q + (i & 3), (double)tot);   // value i = cluster tile i / 4 of the group, register i % 4
}

// the R rows of one tile (:503)
template <int MT>
__device__ __forceinline__ void round_store_rows(float* R, int Kp, int q, const RoundTile<MT>& T, float scl) {
    if (T.cell < 0) return;
    float* row = R + (size_t)T.cell * Kp;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int col = 16 * mt + 4 * q;
        if (col < Kp) st4(row + col, T.arg[mt] * scl);
    }
}
// STORE0 = false: the first tile's rows are out already (k_round stores them in front of the second tile's exp pass: a longer,
// flatter burst of row stores -- 0-2 % on every configuration, profiles/r06_ab_k_round_early_store.txt)
template <int MT, bool STORE0 = true>
__device__ __forceinline__ void round_post_pass2(float* R, int Kp, double* Sd, int c16, int q, const RoundTile<MT>& T0,
                                                 float scl0, bool has1, const RoundTile<MT>& T1, float scl1) {
    constexpr int K16 = 16 * MT;
    const bool live0 = T0.cell >= 0, live1 = has1 && T1.cell >= 0;
    float* row0 = R + (size_t)(live0 ? T0.cell : 0) * Kp;
    float* row1 = R + (size_t)(live1 ? T1.cell : 0) * Kp;
    const bool joint = has1 && T1.grp == T0.grp;           // wave-uniform
    double* sd0 = Sd + (size_t)T0.grp * K16;
    double* sd1 = Sd + (size_t)T1.grp * K16;
    if (!has1) scl1 = 0.f;
#if HMX_ROUND_RS
    // four cluster tiles at a time: R rows out, then the group's 16 values per lane summed over the cells (:506-507)
    auto group = [&](auto nm_c, int g) {
        constexpr int NM = decltype(nm_c)::value;
        f32x4 rv0[NM], rv1[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int mt = 4 * g + m, col = 16 * mt + 4 * q;
            rv0[m] = T0.arg[mt] * scl0;                    // :503
            rv1[m] = T1.arg[mt] * scl1;
#if !(HMX_RABL & 2)
#if HMX_ROUND_R_NT   /* (experiment of round 6: the R rows are written once per round and read once by the next pass -- streaming stores) */
            if (live0 && col < Kp) __builtin_nontemporal_store(rv0[m], reinterpret_cast<f32x4*>(row0 + col));
            if (live1 && col < Kp) __builtin_nontemporal_store(rv1[m], reinterpret_cast<f32x4*>(row1 + col));
#else
            if (STORE0 && live0 && col < Kp) st4(row0 + col, rv0[m]);
            if (live1 && col < Kp) st4(row1 + col, rv1[m]);
#endif
#endif
        }
#if !(HMX_RABL & 1)
        if (joint || !has1) {                              // one table row for both tiles (scl1 == 0 without a second tile)
#pragma unroll
            for (int m = 0; m < NM; ++m) rv0[m] += rv1[m];
            block_sums_rs<NM>(rv0, sd0 + 64 * g, c16, q);
        } else {
            block_sums_rs<NM>(rv0, sd0 + 64 * g, c16, q);
            block_sums_rs<NM>(rv1, sd1 + 64 * g, c16, q);
        }
#endif
    };
#pragma unroll
    for (int g = 0; g < MT / 4; ++g) group(std::integral_constant<int, 4>{}, g);
    if constexpr (MT % 4 != 0) group(std::integral_constant<int, (MT % 4 ? MT % 4 : 1)>{}, MT / 4);
#else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int col = 16 * mt + 4 * q;
        const f32x4 rv0 = T0.arg[mt] * scl0;               // :503
        const f32x4 rv1 = T1.arg[mt] * scl1;
#if !(HMX_RABL & 2)
        if (live0 && col < Kp) st4(row0 + col, rv0);
        if (live1 && col < Kp) st4(row1 + col, rv1);
#endif
#if !(HMX_RABL & 1)
        if (joint) {
            const f32x4 sm = rv0 + rv1;
            f32x4 ss;
#pragma unroll
            for (int r = 0; r < 4; ++r) ss[r] = row16_sum(sm[r]);      // (:506-507)
            if (c16 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(sd0 + col + r, (double)ss[r]);
            }
        } else {
            f32x4 s0, s1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = row16_sum(rv0[r]);
                s1[r] = row16_sum(rv1[r]);
            }
            if (c16 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(sd0 + col + r, (double)s0[r]);
                    if (has1) atomicAdd(sd1 + col + r, (double)s1[r]);
                }
            }
        }
#endif
    }
#endif
}

// ------------------------------------------------------------------------------------------
// k_assign_wide: the assignment of one block (or of all cells: init_cluster) for shapes beyond the
// LDS-resident kernels -- K up to 208, d up to 208 (BASELINE config 5: 200 x 200): the MFMA-bound
// regime (2 d K flop per cell against 4 (d + K) bytes).  The centroid table (160 KB) does not fit
// the LDS next to anything else, so the PC dimension is walked in 16-column k-steps: a step's
// centroid columns (K16 x 16 floats, 13 KB) are staged once per workgroup, double buffered, the
// next step's columns and Z_cos pieces travelling in registers while the current step multiplies
// (52 MFMAs per wave and step).  Eight waves, one tile each per pass, 8 tiles per staged step;
// the finishing passes are k_round's (one exp per entry, objective sums in the same pass).
// Rows of Z_cos are padded to a multiple of 16 floats for these shapes (no tail steps).
// ------------------------------------------------------------------------------------------
#define WIDE_WAVES 8
template <int MT, bool PENALTY>
__global__ __launch_bounds__(64 * WIDE_WAVES, 2) void k_assign_wide(AssignArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K16 = 16 * MT;
    float* Ysh = reinterpret_cast<float*>(smem);                         // 2 x K16 x 20
    float* nis = Ysh + 2 * K16 * 20;                                     // K16: -1/sigma (-60 for pads)
    float* sig = nis + K16;                                              // K16
    double* Sd = reinterpret_cast<double*>(sig + K16);                   // G x K16 block sums (when they fit)
    double* objw = Sd + (a.tables_in_lds ? (size_t)a.G * K16 : 0);       // waves x 2
    int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar
    int lane = tid & 63;
    int c16 = lane & 15, q = lane >> 4;                         // (refreshed per pass, see the loop)
    const int nkb = a.dp >> 4;
    const int GK = a.G * K16;
    const int tile_begin = a.blk_start ? a.blk_start[a.blk] : a.tile_begin;
    const int tile_end = a.blk_start ? a.blk_start[a.blk + 1] : a.tile_end;
    const int ntiles = tile_end - tile_begin;

    const bool hard = !PENALTY && a.hn != nullptr;              // Lloyd iteration of the device k-means: nearest centre, one-hot row
    for (int i = tid; i < K16; i += 64 * WIDE_WAVES) {
        const float sgm = (i < a.K) ? a.sigma[i] : 0.f;
        sig[i] = hard ? a.hn[i] : sgm;                   // (hard: the half squared norms take sigma's place in LDS)
        nis[i] = (i < a.K) ? -1.0f / sgm : -60.f;       // pads: Y row 0 -> dist 2 -> arg -120 -> exp == 0
    }
    if (a.tables_in_lds)
        for (int i = tid; i < GK; i += 64 * WIDE_WAVES) Sd[i] = 0.0;
    __syncthreads();

    constexpr int YPT = (K16 * 4 + 64 * WIDE_WAVES - 1) / (64 * WIDE_WAVES);   // 16-byte pieces of a step per thread
    double km_acc = 0.0, ent_acc = 0.0;
    int stage = 0;
    for (int base = blockIdx.x * WIDE_WAVES; base < ntiles; base += gridDim.x * WIDE_WAVES) {   // workgroup-uniform trip count
        // the lane's coordinates pass through an empty asm once per pass: what is derived from them (staging and fragment
        // addresses) is recomputed instead of being kept -- partly in scratch -- across the passes
        asm volatile("" : "+v"(tid), "+v"(lane), "+v"(c16), "+v"(q));
        const int t = tile_begin + base + wv;
        const bool has = base + wv < ntiles;
        RoundTile<MT> T;
        T.cell = has ? a.cells[(size_t)t * 16 + c16] : -1;
        T.grp = has ? a.tile_grp[t] : 0;
        const float* zr = a.Zcos + (size_t)(T.cell >= 0 ? T.cell : 0) * a.dp;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) T.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 ynext[YPT], bnext;
        auto fetch_step = [&](int kb) {
#pragma unroll
            for (int p2 = 0; p2 < YPT; ++p2) {
                const int i = tid + 64 * WIDE_WAVES * p2;
                ynext[p2] = (i < K16 * 4) ? ld4(a.Y + (size_t)(i >> 2) * a.ldy + 16 * kb + 4 * (i & 3)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            bnext = ld4(zr + 16 * kb + 4 * q);
        };
        fetch_step(0);
        for (int kb = 0; kb < nkb; ++kb) {
            const f32x4 b = bnext;
#pragma unroll
            for (int p2 = 0; p2 < YPT; ++p2) {
                const int i = tid + 64 * WIDE_WAVES * p2;
                if (i < K16 * 4) st4(Ysh + (size_t)stage * K16 * 20 + (i >> 2) * 20 + 4 * (i & 3), ynext[p2]);
            }
            if (kb + 1 < nkb) fetch_step(kb + 1);
            __syncthreads();
            const float* Yst = Ysh + (size_t)stage * K16 * 20;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 ya = ld4(Yst + (16 * mt + c16) * 20 + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) T.arg[mt] = MFMA16(ya[i], b[i], T.arg[mt]);
            }
            stage ^= 1;
        }
        if (hard) {
            // sklearn's Lloyd step (harmony.py:370-372): the nearest centre maximises z.c - |c|^2 / 2; ties go to the smaller index
            float best = -INFINITY;
            int bk = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 h4 = ld4(sig + 16 * mt + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = T.arg[mt][r] - h4[r];
                    if (sc > best) { best = sc; bk = 16 * mt + 4 * q + r; }
                }
            }
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float ob = __shfl_xor(best, m, 64);
                const int ok = __shfl_xor(bk, m, 64);
                if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; }
            }
            if (has && T.cell >= 0) {
                float* row = a.R + (size_t)T.cell * a.Kp;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int col = 16 * mt + 4 * q;
                    f32x4 oh;
#pragma unroll
                    for (int r = 0; r < 4; ++r) oh[r] = (col + r == bk) ? 1.f : 0.f;
                    if (col < a.Kp) st4(row + col, oh);
                }
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 ni = ld4(nis + 16 * mt + 4 * q);
            const f32x4 one = (f32x4){1.f, 1.f, 1.f, 1.f};
            T.arg[mt] = (2.f * (one - T.arg[mt])) * ni;      // dist = 2 (1 - Y.Z) (:447), arg = -dist / sigma (:466)
        }
        if (has) {
            float scl;
            round_post_pass1<MT, PENALTY>(sig, a.rp, a.lrp, q, T, scl, km_acc, ent_acc);
            if (a.tables_in_lds) {
                round_post_pass2<MT>(a.R, a.Kp, Sd, c16, q, T, scl, false, T, 0.f);
            } else {   // the (group, cluster) table does not fit the LDS: fp64 atomics straight to memory
                const bool live = T.cell >= 0;
                float* row = a.R + (size_t)(live ? T.cell : 0) * a.Kp;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int col = 16 * mt + 4 * q;
                    const f32x4 rv = T.arg[mt] * scl;
                    if (live && col < a.Kp) st4(row + col, rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = row16_sum(rv[r]);
                        if (c16 == 0 && sv != 0.f) atomicAdd(&a.S_out[(size_t)T.grp * K16 + col + r], (double)sv);
                    }
                }
            }
        }
    }
    km_acc = wave_sum_all(km_acc);
    ent_acc = wave_sum_all(ent_acc);
    if (lane == 0) {
        objw[2 * wv] = km_acc;
        objw[2 * wv + 1] = ent_acc;
    }
    __syncthreads();
    if (tid < 2) {
        double v = 0.0;
        for (int w = 0; w < WIDE_WAVES; ++w) v += objw[2 * w + tid];
        if (v != 0.0) atomicAdd(&a.obj[2 * (blockIdx.x & (HMX_OBJ_SLOTS - 1)) + tid], v);
    }
    if (a.tables_in_lds)
        for (int i = tid; i < GK; i += 64 * WIDE_WAVES) {
            const double v = Sd[i];
            if (v != 0.0) atomicAdd(&a.S_out[i], v);
        }
}

// ------------------------------------------------------------------------------------------
// k_assign_wide3: the wide block assignment on the bf16 matrix pipe (hmx_device.h: every fp32 operand as the exact sum of
// three bf16 terms, six products, fp32 accumulation) with the centroids PRE-SPLIT.  The wide shapes are bound by the f32-input
// MFMA (52 k-steps of 32 cycles per cluster tile and cell tile at d = 208; here 7 x 6 of 16).  A first cut of round 5 kept an
// fp32 centroid ring and split the centroid values in registers, per workgroup and k-step -- 36 vector instructions for 12
// MFMAs: it ran at the VECTOR rate, 59 us per block launch at the configs[4] shard against 66 with the f32-input MFMA
// (profiles/r05_ab_wide_bf16_pipe.txt; NOTES.md).  But Y changes once per ROUND: k_y_planes (one small launch per round) writes it as the A fragments of v_mfma_f32_16x16x32_bf16, three bf16
// planes, Yf[step s][plane h, m, l][cluster tile mt][lane][8 bf16] (row i = cluster 16 mt + c16, k slot j of lane (c16, q) =
// PC 32 s + 8 q + j, zeros past the row).  A (plane, tile) fragment is 1 KB contiguous: one LDS-DMA request brings it, one
// conflict-free 16-byte read per lane hands it to the matrix pipe, no vector instruction touches it.  What is left to split
// in registers are the tiles' own Z_cos values, once per step (72 instructions for 156 MFMAs).
// One workgroup of EIGHT waves per CU (a step of all three planes is 3 MT KB: the two-slot ring takes 78 KB at K16 = 208),
// two tiles per wave, sixteen per workgroup; table rows and block sums for the groups of those sixteen tiles (a block's list is
// sorted by group); finishing passes as in k_round.  One vmcnt(0) + one barrier per step; the requests of step s+1 go out behind the
// first cluster tile of step s.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_y_planes(const float* __restrict__ Y, int ldy, int mt_n, unsigned* __restrict__ Yf) {
    const int lane = threadIdx.x, c16 = lane & 15, q = lane >> 4;
    const int s = blockIdx.x / mt_n, mt = blockIdx.x - s * mt_n;
    const float* row = Y + (size_t)(16 * mt + c16) * ldy;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * s + 8 * q + j;
        x[j] = (k < ldy) ? row[k] : 0.f;
    }
    u32x4 pl[3];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned h, m, l;
        bf16_split3((f32x2){x[2 * p], x[2 * p + 1]}, h, m, l);
        pl[0][p] = h; pl[1][p] = m; pl[2][p] = l;
    }
#pragma unroll
    for (int pn = 0; pn < 3; ++pn) *reinterpret_cast<u32x4*>(Yf + (((size_t)s * 3 + pn) * mt_n + mt) * 256 + 4 * lane) = pl[pn];
}

#define WIDE3_WAVES 8
#define WIDE3_SLOTS (2 * WIDE3_WAVES)
#define WIDE3_TSUB 4  /* the fused table's cluster masses: partial sums over the groups g = th (mod 4) */
typedef double f64x2 __attribute__((ext_vector_type(2)));
#ifdef HMX_WIDE3_PROF   /* timing experiments only: s_memtime stamps per wave (workgroups x waves x 16) */
__device__ unsigned long long g_w3prof[512 * WIDE3_WAVES * 32];
#define W3STAMP(k) do { if (lane == 0 && blockIdx.x < 512) g_w3prof[((size_t)blockIdx.x * WIDE3_WAVES + wv) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W3ACC(k, v) do { if (lane == 0 && blockIdx.x < 512) g_w3prof[((size_t)blockIdx.x * WIDE3_WAVES + wv) * 32 + (k)] += (v); } while (0)
#define W3ZERO(k) do { if (lane == 0 && blockIdx.x < 512) g_w3prof[((size_t)blockIdx.x * WIDE3_WAVES + wv) * 32 + (k)] = 0; } while (0)
#else
#define W3STAMP(k) do { } while (0)
#define W3ACC(k, v) do { } while (0)
#define W3ZERO(k) do { } while (0)
#endif
template <int MT>
__global__ __launch_bounds__(64 * WIDE3_WAVES, 1) void k_assign_wide3(AssignArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K16 = 16 * MT;
    constexpr int SLOT = 3 * MT * 256;                                   // dwords of a ring slot (one step, three planes)
    unsigned* ring = reinterpret_cast<unsigned*>(smem);                  // 2 x SLOT
    float* sig = reinterpret_cast<float*>(ring + 2 * SLOT);              // K16
    float* nis = sig + K16;                                              // K16: -2 log2(e) / sigma (-200 for pads)
    float* rpL = nis + K16;                                              // slots x K16
    float* lrpL = rpL + WIDE3_SLOTS * K16;
    double* Sd = reinterpret_cast<double*>(lrpL + WIDE3_SLOTS * K16);    // slots x K16 block sums
    double* objw = Sd + WIDE3_SLOTS * K16;                               // waves x 2
    int* tg = reinterpret_cast<int*>(objw + 2 * WIDE3_WAVES);            // group of the workgroup's tile j (-1: no such tile)
    int* ts = tg + WIDE3_SLOTS;                                          // its slot: tiles of one group share table rows and sums
    int* sg = ts + WIDE3_SLOTS;                                          // group of a slot
    double* Tp = reinterpret_cast<double*>(sg + WIDE3_SLOTS);            // WIDE3_TSUB x K16 partial cluster masses (fused table only)
    float* spr = reinterpret_cast<float*>(Tp + WIDE3_TSUB * K16);        // Pr_b of a slot's group
    float* sth = spr + WIDE3_SLOTS;                                      // theta of a slot's group
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int ns = (a.dp + 31) >> 5;                                     // k-steps of 32 PCs
    W3STAMP(0); W3ZERO(9);
#ifdef HMX_WIDE3_PROF
    if (lane == 0 && blockIdx.x < 512) g_w3prof[((size_t)blockIdx.x * WIDE3_WAVES + wv) * 32 + 16] = wall_clock64();
#endif
    // The prologue is a chain of memory round trips (2-3 k cycles each at the start of a launch, nothing of the block is in a
    // cache): what does not depend on the block -- the first centroid fragments, sigma, the O / S tables of the fused table --
    // is requested FIRST, then the block's cells and groups, then the Z rows; the table is folded out of registers once the
    // slots are known (profiles/r06_ab_wide3_prologue.txt: 22 k of 92 k cycles per wave with one round trip after the other).
    auto u64 = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    };
    const unsigned long long ysrc = u64((unsigned long long)a.Yf);
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring);
    const unsigned voff = 16u * (unsigned)lane;
    auto request = [&](int s) {                                          // this wave's fragments of step s: pieces wv, wv + 8, ... of 3 MT
        const unsigned long long src0 = ysrc + (unsigned long long)s * (SLOT * 4);
        const unsigned zone0 = ring0 + (unsigned)(s & 1) * (SLOT * 4);
#pragma unroll
        for (int j = 0; j < (3 * MT + WIDE3_WAVES - 1) / WIDE3_WAVES; ++j) {
            const int p = wv + WIDE3_WAVES * j;                          // wave-uniform
            if (p < 3 * MT) {
                const unsigned long long src = src0 + 1024ull * p;
                const unsigned zone = zone0 + 1024u * p;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(zone) : "memory", "m0");
            }
        }
    };
    request(0);
    // the fused table's inputs: thread (th, pair of clusters tk) takes groups th, th + 4, ... -- eight per trip, 24 sixteen-byte loads
    constexpr int KP = K16 / 2;
    const bool tsum = a.fuse_table && tid < WIDE3_TSUB * KP;
    const int th = tid / KP, tk = 2 * (tid - th * KP);
    f64x2 op[8], sa[8], ss[8];
    auto table_load = [&](int g0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t i = (size_t)min(g0 + WIDE3_TSUB * j, a.G - 1) * K16 + tk;
            op[j] = *reinterpret_cast<const f64x2*>(a.O_prev + i);
            sa[j] = a.S_add ? *reinterpret_cast<const f64x2*>(a.S_add + i) : (f64x2){0.0, 0.0};
            ss[j] = *reinterpret_cast<const f64x2*>(a.S_sub + i);
        }
    };
    if (tsum) table_load(th);
    const float sgm = (tid < a.K) ? a.sigma[tid] : 0.f;                  // (K16 <= 208 < threads of the workgroup)
    __builtin_amdgcn_sched_barrier(0);
    W3STAMP(10);
    const int tile_begin = a.blk_start ? a.blk_start[a.blk] : a.tile_begin;
    const int tile_end = a.blk_start ? a.blk_start[a.blk + 1] : a.tile_end;
    const int ntiles = tile_end - tile_begin;
    const int base = blockIdx.x * WIDE3_SLOTS;
    if (base >= ntiles) {                                                // (the grid is sized for an upper bound of the block)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the fragment requests write LDS: not past the end of the workgroup
        return;
    }

    W3STAMP(11);
    RoundTile<MT> T0, T1;
    const int j0 = base + 2 * wv;
    const bool has0 = j0 < ntiles, has1 = j0 + 1 < ntiles;              // wave-uniform
    T0.cell = has0 ? a.cells[(size_t)(tile_begin + j0) * 16 + c16] : -1;
    T1.cell = has1 ? a.cells[(size_t)(tile_begin + j0 + 1) * 16 + c16] : -1;
    const int tgv = (tid < WIDE3_SLOTS && base + tid < ntiles) ? a.tile_grp[tile_begin + base + tid] : -1;
    __builtin_amdgcn_sched_barrier(0);
    const float* zr0 = a.Zcos + (size_t)(T0.cell >= 0 ? T0.cell : 0) * a.dp + 8 * q;
    const float* zr1 = a.Zcos + (size_t)(T1.cell >= 0 ? T1.cell : 0) * a.dp + 8 * q;
    f32x4 z[4];                                                          // raw Z_cos values of the coming step (dead once the step has split them)
    auto load_z = [&](int s) {                                           // ordinary loads, pinned where they are written; PCs past the row are zeros
        __builtin_amdgcn_sched_barrier(0);
        const bool in_row = 32 * s + 8 * q < a.dp;                       // (dp is a multiple of 16: the lane's eight columns are inside or outside together)
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        z[0] = in_row ? ld4(zr0 + 32 * s) : zero;
        z[1] = in_row ? ld4(zr0 + 32 * s + 4) : zero;
        z[2] = in_row ? ld4(zr1 + 32 * s) : zero;
        z[3] = in_row ? ld4(zr1 + 32 * s + 4) : zero;
        __builtin_amdgcn_sched_barrier(0);
    };
    load_z(0);
    W3STAMP(12);
    const float prv = (a.fuse_table && tgv >= 0) ? a.Pr_b[tgv] : 0.f;    // (one batch variable: group g is batch g)
    const float thv = (a.fuse_table && tgv >= 0) ? a.theta[tgv] : 0.f;

    // ---- set-up: sigma, the groups of the workgroup's tiles, their table rows, zeroed sums ----
    if (tid < K16) {
        sig[tid] = sgm;
        nis[tid] = (tid < a.K) ? -(2.885390081777926814f / sgm) : -200.f;   // -c_k = -2 log2(e) / sigma_k; pads: Y row 0 -> 2^-200 == 0
    }
    if (tid < WIDE3_SLOTS) tg[tid] = tgv;
    for (int i = tid; i < WIDE3_SLOTS * K16; i += 64 * WIDE3_WAVES) Sd[i] = 0.0;
    W3STAMP(13);
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0) only: the requests above stay in flight
    __builtin_amdgcn_s_barrier();
    W3STAMP(14);
    if (tid < WIDE3_SLOTS) {
        int slot = 0;
        for (int u = 1; u <= tid; ++u) slot += (tg[u] != tg[u - 1] && tg[u] >= 0) ? 1 : 0;
        ts[tid] = slot;
        if (tg[tid] >= 0 && (tid == 0 || tg[tid] != tg[tid - 1])) { sg[slot] = tg[tid]; spr[slot] = prv; sth[slot] = thv; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    const int nslots = ts[WIDE3_SLOTS - 1] + 1;
    W3STAMP(1);
    if (a.fuse_table) {
        // The block's diversity table (k_block_table's arithmetic, harmony.py:491-499; one batch variable: group g is batch g) built
        // HERE instead of by a launch of its own in front of every block (200 launches of ~9.5 us per Harmony iteration at
        // configs[4]): O of every group without this block's old sums and with the previous block's new ones -- complete since
        // the previous launch ended -- summed to the cluster masses by WIDE3_TSUB x K16 / 2 threads out of the registers filled at
        // the top; the O values of the workgroup's own groups are parked in rpL and turned into ratio ** theta and its log below.
        // Workgroup 0 also writes the O chain.
        if (tsum) {
            f64x2 t = {0.0, 0.0};
            for (int g0 = th; g0 < a.G; g0 += 8 * WIDE3_TSUB) {
                if (g0 != th) table_load(g0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int g = g0 + WIDE3_TSUB * j;
                    if (g < a.G) {
                        const f64x2 o = op[j] + sa[j] - ss[j];
                        t += o;
                        if (blockIdx.x == 0 && a.O_out) *reinterpret_cast<f64x2*>(a.O_out + (size_t)g * K16 + tk) = o;
                        for (int sl = 0; sl < nslots; ++sl)
                            if (sg[sl] == g) { rpL[sl * K16 + tk] = (float)o[0]; rpL[sl * K16 + tk + 1] = (float)o[1]; }
                    }
                }
            }
            Tp[th * K16 + tk] = t[0];
            Tp[th * K16 + tk + 1] = t[1];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        W3STAMP(2);
        for (int i = tid; i < nslots * K16; i += 64 * WIDE3_WAVES) {
            const int sl = i / K16, k = i - sl * K16;
            const float O = rpL[i];
            double T = Tp[k];
#pragma unroll
            for (int u = 1; u < WIDE3_TSUB; ++u) T += Tp[u * K16 + k];   // (all WIDE3_TSUB partials exist: 4 x K16 / 2 <= 416 threads)
            const float E = (float)T * spr[sl];
            const float oe = fmaxf(O + E, 1e-8f);                       // :495-496
            const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);      // :497-498
            const float rp = pow_unit(ratio, sth[sl]);                  // :499 (v_log_f32 / v_exp_f32 with split products, as k_round's table)
            rpL[i] = rp;
            lrpL[i] = __builtin_amdgcn_logf(rp) * 0.693147182464599609375f;
        }
    } else
    for (int i = tid; i < nslots * K16; i += 64 * WIDE3_WAVES) {
        const int sl = i / K16, k = i - sl * K16;
        const size_t src = (size_t)sg[sl] * K16 + k;
        rpL[i] = a.rp[src];
        lrpL[i] = a.lrp[src];
    }
    T0.grp = ts[2 * wv];
    T1.grp = has1 ? ts[2 * wv + 1] : T0.grp;              // (a missing tile computes on cell 0's row and counts for nothing)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        T0.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        T1.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    W3STAMP(3);

    // ---- the k-steps ---------------------------------------------------------------------------------------------------
    auto split4 = [&](const f32x4& lo, const f32x4& hi, u32x4 (&pl)[3]) {   // 8 values -> three bf16 planes (k slot j <-> value j)
        unsigned h, m, l;
        bf16_split3((f32x2){lo[0], lo[1]}, h, m, l); pl[0][0] = h; pl[1][0] = m; pl[2][0] = l;
        bf16_split3((f32x2){lo[2], lo[3]}, h, m, l); pl[0][1] = h; pl[1][1] = m; pl[2][1] = l;
        bf16_split3((f32x2){hi[0], hi[1]}, h, m, l); pl[0][2] = h; pl[1][2] = m; pl[2][2] = l;
        bf16_split3((f32x2){hi[2], hi[3]}, h, m, l); pl[0][3] = h; pl[1][3] = m; pl[2][3] = l;
    };
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#ifdef HMX_WIDE3_PROF
        const unsigned long long w3t0 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's fragments and Z values of step s (requested a whole step ago)
#ifdef HMX_WIDE3_PROF
        const unsigned long long w3t1 = __builtin_amdgcn_s_memtime();
#endif
        wg_barrier_lds();                                            // everybody's fragments of step s are in; nobody reads step s-1 any more
#ifdef HMX_WIDE3_PROF
        W3ACC(9, (__builtin_amdgcn_s_memtime() - w3t1) << 32 | (w3t1 - w3t0));
#endif
        const unsigned* slot = ring + (size_t)(s & 1) * SLOT + 4 * lane;
        u32x4 zp0[3], zp1[3];                                        // the two tiles' B planes of this step
        split4(z[0], z[1], zp0);
        split4(z[2], z[3], zp1);
        u32x4 yp[2][3];                                              // A planes of the current / the next cluster tile
        auto fetch = [&](int mt, u32x4 (&pl)[3]) {
#pragma unroll
            for (int pn = 0; pn < 3; ++pn) pl[pn] = ld4u(slot + (pn * MT + mt) * 256);
        };
        auto products = [&](int mt, const u32x4 (&pl)[3]) {          // smallest terms first; the two tiles alternate
            T0.arg[mt] = MFMA_BF16(pl[2], zp0[0], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[2], zp1[0], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(pl[0], zp0[2], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[0], zp1[2], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(pl[1], zp0[1], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[1], zp1[1], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(pl[1], zp0[0], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[1], zp1[0], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(pl[0], zp0[1], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[0], zp1[1], T1.arg[mt]);
            T0.arg[mt] = MFMA_BF16(pl[0], zp0[0], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[0], zp1[0], T1.arg[mt]);
        };
        fetch(0, yp[0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt + 1 < MT) fetch(mt + 1, yp[(mt + 1) & 1]);
            products(mt, yp[mt & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (mt == 0 && s + 1 < ns) {                             // behind the first cluster tile: slot (s+1) & 1 is free, z[] is split
                request(s + 1);
                load_z(s + 1);
            }
        }
    }

    // ---- finish: exp, penalty, renormalisation, R rows, block sums, objective terms (k_round's passes) ---------------
    W3STAMP(4);
    double km_acc = 0.0, ent_acc = 0.0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 ni = ld4(nis + 16 * mt + 4 * q);       // dist = 2 (1 - Y.Z) (:447), arg = -dist / sigma (:466), here in log2 units:
        T0.arg[mt] = __builtin_elementwise_fma(T0.arg[mt], -ni, ni);   // c_k (y.z - 1), the argument of the hardware exp2 (one fma per entry)
        T1.arg[mt] = __builtin_elementwise_fma(T1.arg[mt], -ni, ni);
    }
    if (has0) {
        float scl0, scl1 = 0.f;
        round_post_pass1<MT, true, true, false, (HMX_ROUND_PK != 0 && MT <= 8)>(sig, rpL, lrpL, q, T0, scl0, km_acc, ent_acc);
        if (has1) round_post_pass1<MT, true, true, false, (HMX_ROUND_PK != 0 && MT <= 8)>(sig, rpL, lrpL, q, T1, scl1, km_acc, ent_acc);
        W3STAMP(5);
        round_post_pass2<MT>(a.R, a.Kp, Sd, c16, q, T0, scl0, has1, T1, scl1);
    } else {
        W3STAMP(5);
    }
    W3STAMP(6);
    km_acc = wave_sum_all(km_acc);
    ent_acc = wave_sum_all(ent_acc);
    if (lane == 0) {
        objw[2 * wv] = km_acc;
        objw[2 * wv + 1] = ent_acc;
    }
    __syncthreads();
    if (tid < 2) {
        double v = 0.0;
        for (int w = 0; w < WIDE3_WAVES; ++w) v += objw[2 * w + tid];
        if (v != 0.0) atomicAdd(&a.obj[2 * (blockIdx.x & (HMX_OBJ_SLOTS - 1)) + tid], v);
    }
    for (int i = tid; i < nslots * K16; i += 64 * WIDE3_WAVES) {
        const double v = Sd[i];
        const int sl = i / K16;
        if (v != 0.0) atomicAdd(&a.S_out[(size_t)sg[sl] * K16 + (i - sl * K16)], v);
    }
    W3STAMP(7);
#ifdef HMX_WIDE3_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (stamp 8: the wave's stores and atomics have been acknowledged)
    W3STAMP(8);
    if (lane == 0 && blockIdx.x < 512) g_w3prof[((size_t)blockIdx.x * WIDE3_WAVES + wv) * 32 + 17] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------
// k_sweep_wide3: ALL blocks of a wide update_R sweep (harmony.py:476-507) in ONE persistent launch -- k_assign_wide3's chunk
// of sixteen tiles as the unit of work, chunk c of every block on workgroup c mod grid, one workgroup per CU, all resident.
// What a launch per block pays twenty times per round -- 9 us between the last wave of a launch and the first of the next,
// the chain of cold round trips kernel arguments -> block bounds -> cells -> Z rows, the O / S tables read by every
// workgroup (profiles/r06_ab_wide3_prologue.txt) -- is paid once or hidden behind the block before:
//   * O of all groups lives in REGISTERS of the workgroup (thread (th, pair of clusters): groups th, th + 4, ...; G <= 32),
//     updated per block with the block's removal sums (plain loads, requested early) and the previous block's new sums;
//   * the new sums travel as k_round's SELF-VALIDATING fixed-point words (group-affine map): a chunk adds
//     (1 << 55) + sum * 2^32  to the word of every (group it holds, cluster) of the block's table with one non-returning 64-bit
//     add; a reader that finds the count field equal to the number of chunks holding the group (from the (block, group) run
//     offsets of the lists) holds the complete sum.  No counter, no flag, no returning atomic; the successful poll IS the data;
//   * cells, groups, first centroid fragments and first Z rows of a workgroup's next chunk are requested behind the
//     finishing pass of the current one.
// The last block's sums go out as fp64 adds (the closing table launch reads them), workgroup 0 writes the O chain as before.
// A wait that exceeds the spin limit counts the workgroup in *fail and ends the launch; the host replays the round with one
// launch per block.  Table and sums per (block, group, cluster): ONE word -- at most 511 chunks per block and group, sums
// below 2^23 (the host checks the block size).
// ------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(64 * WIDE3_WAVES, 1) void k_sweep_wide3(AssignArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K16 = 16 * MT;
    constexpr int SLOT = 3 * MT * 256;                                   // dwords of a ring slot (one step, three planes)
    constexpr int KP = K16 / 2;
    constexpr unsigned long long FXM = (1ull << 55) - 1;
    unsigned* ring = reinterpret_cast<unsigned*>(smem);                  // 2 x SLOT
    float* sig = reinterpret_cast<float*>(ring + 2 * SLOT);              // K16
    float* nis = sig + K16;                                              // K16: -2 log2(e) / sigma (-200 for pads)
    float* rpL = nis + K16;                                              // slots x K16
    float* lrpL = rpL + WIDE3_SLOTS * K16;
    double* Sd = reinterpret_cast<double*>(lrpL + WIDE3_SLOTS * K16);    // slots x K16 block sums
    double* objw = Sd + WIDE3_SLOTS * K16;                               // waves x 2
    int* tsA = reinterpret_cast<int*>(objw + 2 * WIDE3_WAVES);           // 2 x slots: slot of the chunk's tile j (tiles of one group share table rows and sums)
    int* sgA = tsA + 2 * WIDE3_SLOTS;                                    // 2 x slots: group of a slot (both by chunk parity: the sums of a chunk go out
                                                                         //            while the first wave already deals with the next chunk)
    double* Tp = reinterpret_cast<double*>(sgA + 2 * WIDE3_SLOTS);       // WIDE3_TSUB x K16 partial cluster masses
    float* prG = reinterpret_cast<float*>(Tp + WIDE3_TSUB * K16);        // 32: Pr_b of a group (one batch variable: group g is batch g)
    float* thG = prG + 32;                                               // 32: theta of a group
    int* bst = reinterpret_cast<int*>(thG + 32);                         // nblk + 1 block starts (tiles)
    int* wfail = bst + a.nblk + 1;                                       // a wave's wait gave up
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int ns = (a.dp + 31) >> 5;                                     // k-steps of 32 PCs
    const int nwg = gridDim.x, wg = blockIdx.x;
#ifdef HMX_WIDE3_PROF
    for (int i_ = 0; i_ < 16; ++i_) W3ZERO(i_);
    unsigned long long pt_ = __builtin_amdgcn_s_memtime();
#define SWSEG(k_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); W3ACC(k_, n_ - pt_); pt_ = n_; } while (0)
#else
#define SWSEG(k_) do { } while (0)
#endif
    const size_t GK = (size_t)a.G * K16;
    auto u64 = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    };
    const unsigned long long ysrc = u64((unsigned long long)a.Yf);
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring);
    const unsigned voff = 16u * (unsigned)lane;
    auto request = [&](int s) {                                          // this wave's fragments of step s: pieces wv, wv + 8, ... of 3 MT
        const unsigned long long src0 = ysrc + (unsigned long long)s * (SLOT * 4);
        const unsigned zone0 = ring0 + (unsigned)(s & 1) * (SLOT * 4);
#pragma unroll
        for (int j = 0; j < (3 * MT + WIDE3_WAVES - 1) / WIDE3_WAVES; ++j) {
            const int p = wv + WIDE3_WAVES * j;                          // wave-uniform
            if (p < 3 * MT) {
                const unsigned long long src = src0 + 1024ull * p;
                const unsigned zone = zone0 + 1024u * p;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(zone) : "memory", "m0");
            }
        }
    };
    request(0);                                                          // (every chunk starts with the same fragments)
    // O of all groups is kept per workgroup in a private global table (a.O_priv: grid x G x K16 doubles, touched only by the
    // threads that own its entries): thread (th, clusters tk, tk + 1) owns groups th, th + 4, ..., th + 28
    const bool tsum = tid < WIDE3_TSUB * KP;
    const int th = tid / KP, tk = 2 * (tid - th * KP);
    double* Opriv = a.O_priv + (size_t)wg * GK;                          // (a second table of the same shape follows at + grid x G x K16)
    const int goff = th * K16 + tk;                                      // (group th, cluster tk) in a G x K16 table
    if (tid < K16) {
        const float sgm = (tid < a.K) ? a.sigma[tid] : 0.f;
        sig[tid] = sgm;
        nis[tid] = (tid < a.K) ? -(2.885390081777926814f / sgm) : -200.f;   // -c_k = -2 log2(e) / sigma_k; pads: Y row 0 -> 2^-200 == 0
    }
    for (int i = tid; i <= a.nblk; i += 64 * WIDE3_WAVES) bst[i] = a.blk_start[i];
    for (int i = tid; i < WIDE3_SLOTS * K16; i += 64 * WIDE3_WAVES) Sd[i] = 0.0;
    if (tid == 0) *wfail = 0;
    if (tid < 32) { prG[tid] = tid < a.G ? a.Pr_b[tid] : 0.f; thG[tid] = tid < a.G ? a.theta[tid] : 0.f; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();

    // what is requested ahead for a chunk: its cells and (in every wave's lanes 0 .. 15) the groups of its sixteen tiles
    int pf_b = -1, pf_c = -1, pf_cell0 = -1, pf_cell1 = -1, pf_tg = -1;
    auto prefetch = [&](int b, int c) {
        const int t0 = bst[b], nt = bst[b + 1] - t0, j0 = 16 * c + 2 * wv;
        pf_cell0 = j0 < nt ? a.cells[(size_t)(t0 + j0) * 16 + c16] : -1;
        pf_cell1 = j0 + 1 < nt ? a.cells[(size_t)(t0 + j0 + 1) * 16 + c16] : -1;
        pf_tg = (lane < WIDE3_SLOTS && 16 * c + lane < nt) ? a.tile_grp[t0 + 16 * c + lane] : -1;
        pf_b = b; pf_c = c;
    };
    {
        const int nt0 = bst[1] - bst[0];
        if (16 * wg < nt0) prefetch(0, wg);
    }
    double km_acc = 0.0, ent_acc = 0.0;
    bool ring_requested = true;                                          // request(0) of the coming chunk is out
    int parity = 0;

#pragma unroll 1
    for (int b = 0; b < a.nblk; ++b) {
        const int tile_begin = bst[b], ntiles = bst[b + 1] - tile_begin;
        const int nch = (ntiles + WIDE3_SLOTS - 1) / WIDE3_SLOTS;
        bool first_chunk = true;
        int c = wg;
#pragma unroll 1
        do {                                                             // (at least once per block: the table is kept by every workgroup)
            const bool work = c < nch;
            SWSEG(0);                                                    // 0: between chunks (the sums' adds, loop overhead)
            // ---- slots of the chunk's groups: every wave works them out for itself (no barrier; sg goes to LDS for indexed reads) ----
            int* sg = sgA + parity * WIDE3_SLOTS;
            parity ^= 1;
            int myslot = 0, nslots = 0;
            if (work) {
                if (!(pf_b == b && pf_c == c)) prefetch(b, c);           // (cold: the first chunk of a workgroup that sat out a block)
                if (!ring_requested) request(0);
                const int tgv = pf_tg;
                const int prev = __shfl_up(tgv, 1);
                const bool first = lane < WIDE3_SLOTS && tgv >= 0 && (lane == 0 || tgv != prev);
                const bool later = lane < WIDE3_SLOTS && lane > 0 && tgv >= 0 && tgv != prev;   // (the serial count of k_assign_wide3: changes of group behind tile 0)
                const unsigned long long changes = __ballot(later);
                myslot = __popcll(changes & ((2ull << lane) - 1ull));
                nslots = __shfl(myslot, WIDE3_SLOTS - 1) + 1;
                if (first) sg[myslot] = tgv;                             // (every wave writes the same values)
            }
            SWSEG(1);                                                    // 1: the chunk's groups in
            const int base = c * WIDE3_SLOTS;
            RoundTile<MT> T0, T1;
            const int j0 = base + 2 * wv;
            const bool has0 = j0 < ntiles, has1 = j0 + 1 < ntiles;      // wave-uniform
            T0.cell = pf_cell0; T1.cell = pf_cell1;
            const float* zr0 = a.Zcos + (size_t)(T0.cell >= 0 ? T0.cell : 0) * a.dp + 8 * q;
            const float* zr1 = a.Zcos + (size_t)(T1.cell >= 0 ? T1.cell : 0) * a.dp + 8 * q;
            f32x4 z[4];
            auto load_z = [&](int s) {                                   // ordinary loads, pinned where they are written; PCs past the row are zeros
                __builtin_amdgcn_sched_barrier(0);
                const bool in_row = 32 * s + 8 * q < a.dp;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                z[0] = in_row ? ld4(zr0 + 32 * s) : zero;
                z[1] = in_row ? ld4(zr0 + 32 * s + 4) : zero;
                z[2] = in_row ? ld4(zr1 + 32 * s) : zero;
                z[3] = in_row ? ld4(zr1 + 32 * s + 4) : zero;
                __builtin_amdgcn_sched_barrier(0);
            };
            if (work) {
                // the distance GEMM needs nothing of the previous block: it runs AHEAD of the hand-off, whose wait then finds the sums in
                load_z(0);
                T0.grp = __shfl(myslot, 2 * wv);
                T1.grp = has1 ? __shfl(myslot, 2 * wv + 1) : T0.grp;
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    T0.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    T1.arg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }

                // ---- the k-steps (k_assign_wide3's) ----------------------------------------------------------------------------
                auto split4 = [&](const f32x4& lo, const f32x4& hi, u32x4 (&pl)[3]) {
                    unsigned h, m, l;
                    bf16_split3((f32x2){lo[0], lo[1]}, h, m, l); pl[0][0] = h; pl[1][0] = m; pl[2][0] = l;
                    bf16_split3((f32x2){lo[2], lo[3]}, h, m, l); pl[0][1] = h; pl[1][1] = m; pl[2][1] = l;
                    bf16_split3((f32x2){hi[0], hi[1]}, h, m, l); pl[0][2] = h; pl[1][2] = m; pl[2][2] = l;
                    bf16_split3((f32x2){hi[2], hi[3]}, h, m, l); pl[0][3] = h; pl[1][3] = m; pl[2][3] = l;
                };
    #pragma unroll 1
                for (int s = 0; s < ns; ++s) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    wg_barrier_lds();
                    const unsigned* slot = ring + (size_t)(s & 1) * SLOT + 4 * lane;
                    u32x4 zp0[3], zp1[3];
                    split4(z[0], z[1], zp0);
                    split4(z[2], z[3], zp1);
                    u32x4 yp[2][3];
                    auto fetch = [&](int mt, u32x4 (&pl)[3]) {
    #pragma unroll
                        for (int pn = 0; pn < 3; ++pn) pl[pn] = ld4u(slot + (pn * MT + mt) * 256);
                    };
                    auto products = [&](int mt, const u32x4 (&pl)[3]) {
                        T0.arg[mt] = MFMA_BF16(pl[2], zp0[0], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[2], zp1[0], T1.arg[mt]);
                        T0.arg[mt] = MFMA_BF16(pl[0], zp0[2], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[0], zp1[2], T1.arg[mt]);
                        T0.arg[mt] = MFMA_BF16(pl[1], zp0[1], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[1], zp1[1], T1.arg[mt]);
                        T0.arg[mt] = MFMA_BF16(pl[1], zp0[0], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[1], zp1[0], T1.arg[mt]);
                        T0.arg[mt] = MFMA_BF16(pl[0], zp0[1], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[0], zp1[1], T1.arg[mt]);
                        T0.arg[mt] = MFMA_BF16(pl[0], zp0[0], T0.arg[mt]);  T1.arg[mt] = MFMA_BF16(pl[0], zp1[0], T1.arg[mt]);
                    };
                    fetch(0, yp[0]);
    #pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (mt + 1 < MT) fetch(mt + 1, yp[(mt + 1) & 1]);
                        products(mt, yp[mt & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (mt == 0 && s + 1 < ns) {
                            request(s + 1);
                            load_z(s + 1);
                        }
                    }
                }

            }
            SWSEG(7);                                                    // 7: the k-steps (first wait behind the R rows' stores)
            // ---- the block's O: without its old sums, with the previous block's new ones (:491-492, :506-507); the chunk's rows parked
            bool failed = false;
            if (tsum) {
                // (addresses: a wave-uniform base per group trip + ONE per-thread offset; groups past G are predicated off, not clamped.
                //  Two passes of four groups each: the whole table at once does not fit the registers beside the loop's state -- nor do the
                //  words of both passes requested ahead (tried: 48 spilled registers at K16 = 208).
                //  ONE round trip per pass: the private table holds O with the removal sums of the block ALREADY taken off -- done
                //  a block ahead, with the next block's sums loaded beside the words of the hand-off.)
                f64x2 t = {0.0, 0.0};
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                    const size_t hoff = (size_t)(4 * half) * (WIDE3_TSUB * K16) + goff;
                    const int g0 = th + WIDE3_TSUB * 4 * half;           // groups g0, g0 + 4, g0 + 8, g0 + 12
                    f64x2 o[4];
                    // (a later chunk of the block: O of the block as the first chunk left it, in the second private table)
                    const double* osrc = (first_chunk ? (b == 0 ? a.O_prev : Opriv) : Opriv + (size_t)nwg * GK) + hoff;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = (g0 + WIDE3_TSUB * j < a.G) ? *reinterpret_cast<const f64x2*>(osrc + (size_t)j * (WIDE3_TSUB * K16)) : (f64x2){0.0, 0.0};
                    if (first_chunk) {
                        f64x2 ss[4];                                     // the NEXT block's old sums (block 0: its own first)
                        const bool pre = b + 1 < a.nblk;
                        if (b == 0) {
                            const double* ssub = a.S_sub + hoff;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                ss[j] = (g0 + WIDE3_TSUB * j < a.G) ? *reinterpret_cast<const f64x2*>(ssub + (size_t)j * (WIDE3_TSUB * K16)) : (f64x2){0.0, 0.0};
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] -= ss[j];  // without the block's old sums (:491-492)
                        }
                        {
                            const double* ssub = a.S_sub + (size_t)(pre ? b + 1 : b) * GK + hoff;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                ss[j] = (pre && g0 + WIDE3_TSUB * j < a.G) ? *reinterpret_cast<const f64x2*>(ssub + (size_t)j * (WIDE3_TSUB * K16)) : (f64x2){0.0, 0.0};
                        }
                        if (b > 0) {
                            const int t0 = bst[b - 1];
                            const int* runs = a.run_tiles + (size_t)(b - 1) * a.G + g0;
                            int rs[4], re[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool v = g0 + WIDE3_TSUB * j < a.G;
                                rs[j] = v ? runs[WIDE3_TSUB * j] : 0;
                                re[j] = v ? runs[WIDE3_TSUB * j + 1] : 0;
                            }
                            const unsigned long long* W = reinterpret_cast<const unsigned long long*>(a.S_out + (size_t)(b - 1) * GK) + hoff;
                            unsigned long long w[4][2];
                            int need[4];
                            unsigned spins = 0;
                            if (a.spin_limit == 0) failed = true;        // test knob: give up without looking
                            while (true) {
                                bool ok = true;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {            // eight independent loads in flight (beside the ones above on the first trip)
                                    w[j][0] = w[j][1] = 0ull;
                                    if (g0 + WIDE3_TSUB * j < a.G) {
                                        const unsigned long long* wp = W + (size_t)j * (WIDE3_TSUB * K16);
                                        w[j][0] = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        w[j][1] = __hip_atomic_load(wp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    }
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) {            // chunks of block b-1 holding tiles of group g
                                    const int r0 = rs[j] - t0, r1 = re[j] - t0;
                                    need[j] = r1 > r0 ? (r1 - 1) / WIDE3_SLOTS - r0 / WIDE3_SLOTS + 1 : 0;
                                    ok = ok && (int)(w[j][0] >> 55) == need[j] && (int)(w[j][1] >> 55) == need[j];
                                }
                                if (__all(ok) || failed) break;
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins > a.spin_limit || (spins % 64 == 0 && __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.fail), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) { failed = true; break; }
                            }
                            W3ACC(10, spins);
#pragma unroll
                            for (int j = 0; j < 4; ++j)                  // with the previous block's new sums (:506-507)
                                o[j] += (f64x2){(double)(long long)(w[j][0] & FXM) * 0x1p-32, (double)(long long)(w[j][1] & FXM) * 0x1p-32};
                        }
                        double* odst = Opriv + hoff;
                        double* hdst = a.O_out + (size_t)b * GK + hoff;
                        const bool more = c + nwg < nch;                 // (uniform) later chunks of this block will want O of the block itself
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (g0 + WIDE3_TSUB * j < a.G) {
                                t += o[j];
                                *reinterpret_cast<f64x2*>(odst + (size_t)j * (WIDE3_TSUB * K16)) = o[j] - ss[j];   // (the next block's old sums taken off ahead)
                                if (more) *reinterpret_cast<f64x2*>(odst + (size_t)nwg * GK + (size_t)j * (WIDE3_TSUB * K16)) = o[j];
                                if (wg == 0) *reinterpret_cast<f64x2*>(hdst + (size_t)j * (WIDE3_TSUB * K16)) = o[j];
                            }
                        }
                    }
                    if (work) {                                          // O of the chunk's groups, parked where its table rows go
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int g = g0 + WIDE3_TSUB * j;
                            if (g < a.G)
                                for (int sl = 0; sl < nslots; ++sl)
                                    if (sg[sl] == g) { rpL[sl * K16 + tk] = (float)o[j][0]; rpL[sl * K16 + tk + 1] = (float)o[j][1]; }
                        }
                    }
                }
                SWSEG(2);                                                // 2: loads, polls, O update, parked rows
                if (first_chunk) {
                    Tp[th * K16 + tk] = t[0];
                    Tp[th * K16 + tk + 1] = t[1];
                    if (failed && lane == 0) *wfail = 1;
                }
            }
            SWSEG(3);                                                    // 3: O update, partial masses, parked rows
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                                // Tp, parked rows, wfail
            SWSEG(4);                                                    // 4: barrier behind the table (waves without table duty wait here for the poll)
            if (*wfail) {
                if (tid == 0) atomicAdd(a.fail, 1.0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (fragment requests write LDS: not past the end of the workgroup)
                return;
            }
            first_chunk = false;
            if (!work) break;
            for (int i = tid; i < nslots * K16; i += 64 * WIDE3_WAVES) {
                const int sl = i / K16, k = i - sl * K16;
                const float O = rpL[i];
                double T = Tp[k];
#pragma unroll
                for (int u = 1; u < WIDE3_TSUB; ++u) T += Tp[u * K16 + k];
                const float E = (float)T * prG[sg[sl]];
                const float oe = fmaxf(O + E, 1e-8f);                   // :495-496
                const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);  // :497-498
                const float rp = pow_unit(ratio, thG[sg[sl]]);          // :499
                rpL[i] = rp;
                lrpL[i] = __builtin_amdgcn_logf(rp) * 0.693147182464599609375f;
            }
            SWSEG(6);                                                    // 6: table rows
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                                // the chunk's table rows
            SWSEG(5);                                                    // 5: barrier behind them
            // ---- finish: exp, penalty, renormalisation, R rows, block sums, objective terms ----------------------------------
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 ni = ld4(nis + 16 * mt + 4 * q);
                T0.arg[mt] = __builtin_elementwise_fma(T0.arg[mt], -ni, ni);
                T1.arg[mt] = __builtin_elementwise_fma(T1.arg[mt], -ni, ni);
            }
            if (has0) {
                float scl0, scl1 = 0.f;
                round_post_pass1<MT, true, true, false, (HMX_ROUND_PK != 0 && MT <= 6)>(sig, rpL, lrpL, q, T0, scl0, km_acc, ent_acc);
                if (has1) round_post_pass1<MT, true, true, false, (HMX_ROUND_PK != 0 && MT <= 6)>(sig, rpL, lrpL, q, T1, scl1, km_acc, ent_acc);
                round_post_pass2<MT>(a.R, a.Kp, Sd, c16, q, T0, scl0, has1, T1, scl1);
            }
            SWSEG(8);                                                    // 8: finishing passes
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                                // Sd complete; the ring is free
            SWSEG(9);                                                    // 9: barrier behind them
            // the workgroup's next chunk: its fragments, cells and groups are requested before the sums go out
            {
                int nb = b, nc = c + nwg;
                if (nc >= nch) { nb = b + 1; nc = wg; }
                ring_requested = false;
                if (nb < a.nblk && 16 * nc < bst[nb + 1] - bst[nb]) {
                    request(0);
                    ring_requested = true;
                    prefetch(nb, nc);
                }
            }
            if (b + 1 < a.nblk) {
                unsigned long long* W = reinterpret_cast<unsigned long long*>(a.S_out + (size_t)b * GK);
                for (int i = tid; i < nslots * K16; i += 64 * WIDE3_WAVES) {
                    const double v = Sd[i];
                    Sd[i] = 0.0;
                    const int sl = i / K16;
                    const unsigned long long word = (1ull << 55) + (unsigned long long)__double2ll_rn(v * 0x1p32);
                    __hip_atomic_fetch_add(W + (size_t)sg[sl] * K16 + (i - sl * K16), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                double* So = a.S_out + (size_t)b * GK;
                for (int i = tid; i < nslots * K16; i += 64 * WIDE3_WAVES) {
                    const double v = Sd[i];
                    Sd[i] = 0.0;
                    const int sl = i / K16;
                    if (v != 0.0) atomicAdd(&So[(size_t)sg[sl] * K16 + (i - sl * K16)], v);
                }
            }
            SWSEG(11);                                                   // 11: next chunk's requests + the sums' adds issued
            // (sg of this chunk is read here while a faster wave may fill the other parity's; Sd entries are zeroed by their reader)
            c += nwg;
        } while (c < nch);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    km_acc = wave_sum_all(km_acc);
    ent_acc = wave_sum_all(ent_acc);
    if (lane == 0) {
        objw[2 * wv] = km_acc;
        objw[2 * wv + 1] = ent_acc;
    }
    __syncthreads();
    if (tid < 2) {
        double v = 0.0;
        for (int w = 0; w < WIDE3_WAVES; ++w) v += objw[2 * w + tid];
        if (v != 0.0) atomicAdd(&a.obj[2 * (blockIdx.x & (HMX_OBJ_SLOTS - 1)) + tid], v);
    }
}

#ifdef HMX_WIDE3_PROF
#include <vector>
static void sweep3_prof_dump(int wgs, int nblk, hipStream_t s) {
    static int calls = 0;
    ++calls;
    if (calls != 25 && calls != 26) return;
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h((size_t)512 * WIDE3_WAVES * 32);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_w3prof), h.size() * 8);
    const char* names[12] = {"between chunks", "groups in (store drain)", "loads + poll + O + park", "Tp", "barrier behind table", "barrier behind rows", "table rows", "k-steps",
                             "finishing passes", "barrier behind passes", "(polls)", "requests+adds"};
    const int n = std::min(wgs, 512);
    fprintf(stderr, "[k_sweep_wide3 prof] call %d, %d wgs, %d blocks: per wave and block, mean over waves (max over waves):\n", calls, wgs, nblk);
    double tot = 0;
    for (int k = 0; k < 12; ++k) {
        double sum = 0, mx = 0;
        for (int w = 0; w < n * WIDE3_WAVES; ++w) { const double v = (double)h[(size_t)w * 32 + k] / nblk; sum += v; mx = std::max(mx, v); }
        fprintf(stderr, "[k_sweep_wide3 prof]   %-24s %9.0f (%9.0f)\n", names[k], sum / (n * WIDE3_WAVES), mx);
        if (k != 10) tot += sum / (n * WIDE3_WAVES);
    }
    fprintf(stderr, "[k_sweep_wide3 prof]   %-24s %9.0f\n", "sum", tot);
}
static void wide3_prof_dump(int wgs, hipStream_t s) {
    static int calls = 0;
    ++calls;
    if (calls != 450 && calls != 451 && calls != 1250) return;
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h((size_t)512 * WIDE3_WAVES * 32);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_w3prof), h.size() * 8);
    const int n = std::min(wgs, 512);
    // (the counters of the eight XCDs are not aligned with each other: spans and skews per XCD = workgroup index mod 8; workgroups
    // past the block's tiles leave old stamps: a wave counts when its stamps lie within 1M ticks of its XCD's latest)
    double seg[9] = {0}, segmax[9] = {0}, wv_ = 0, wb = 0, start = 0, startmax = 0, nw = 0, span = 0, fine[6] = {0};
    for (int x = 0; x < 8; ++x) {
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int g = x; g < n; g += 8) for (int v = 0; v < WIDE3_WAVES; ++v) t1 = std::max(t1, h[((size_t)g * WIDE3_WAVES + v) * 32 + 8]);
        auto live = [&](const unsigned long long* r) { return r[8] + 1000000ull > t1 && r[0] < r[8] && r[0] + 1000000ull > t1; };
        for (int g = x; g < n; g += 8) for (int v = 0; v < WIDE3_WAVES; ++v) { const unsigned long long* r = &h[((size_t)g * WIDE3_WAVES + v) * 32]; if (live(r)) t0 = std::min(t0, r[0]); }
        span += (double)(t1 - t0) / 8;
        for (int g = x; g < n; g += 8) for (int v = 0; v < WIDE3_WAVES; ++v) {
            const unsigned long long* r = &h[((size_t)g * WIDE3_WAVES + v) * 32];
            if (!live(r)) continue;
            nw += 1;
            for (int k = 1; k <= 8; ++k) { const double d = (double)(r[k] - r[k - 1]); seg[k] += d; segmax[k] = std::max(segmax[k], d); }
            wv_ += (double)(r[9] & 0xffffffffull); wb += (double)(r[9] >> 32);
            fine[0] += (double)(r[10] - r[0]); fine[1] += (double)(r[11] - r[10]); fine[2] += (double)(r[12] - r[11]); fine[3] += (double)(r[13] - r[12]);
            fine[4] += (double)(r[14] - r[13]); fine[5] += (double)(r[1] - r[14]);
            start += (double)(r[0] - t0); startmax = std::max(startmax, (double)(r[0] - t0));
        }
    }
    const unsigned long long t0 = 0, t1 = (unsigned long long)span;
    {   // the constant-rate counter (100 MHz, one for the chip): the kernel's active span and the core clock it implies
        unsigned long long w1 = 0, w0 = ~0ull; double life_c = 0, life_w = 0, cnt = 0;
        for (int w = 0; w < n * WIDE3_WAVES; ++w) w1 = std::max(w1, h[(size_t)w * 32 + 17]);
        for (int w = 0; w < n * WIDE3_WAVES; ++w) {
            const unsigned long long* r = &h[(size_t)w * 32];
            if (r[17] + 100000ull < w1 || r[16] > r[17]) continue;
            w0 = std::min(w0, r[16]); life_c += (double)(r[8] - r[0]); life_w += (double)(r[17] - r[16]); cnt += 1;
        }
        fprintf(stderr, "[k_assign_wide3 prof]   constant-rate counter: first wave start to last wave end %.2f us; wave lifetime %.2f us mean = %.0f core ticks (%.0f ticks per us); %.0f waves\n",
                (double)(w1 - w0) / 100.0, life_w / cnt / 100.0, life_c / cnt, life_c / life_w * 100.0, cnt);
    }
    fprintf(stderr, "[k_assign_wide3 prof] call %d, %d wgs, span %.0f ticks (%.1f us at 100 MHz); per wave mean (max): start after first %.0f (%.0f); "
            "setup %.0f (%.0f), table sums %.0f (%.0f), table rows %.0f (%.0f), k-steps %.0f (%.0f) [vmcnt wait %.0f, barrier wait %.0f], pass1 %.0f (%.0f), "
            "pass2 %.0f (%.0f), tail+atomics %.0f (%.0f), drain %.0f (%.0f); %.0f live waves\n", calls, wgs, (double)(t1 - t0), (double)(t1 - t0) / 100.0, start / nw, startmax,
            seg[1] / nw, segmax[1], seg[2] / nw, segmax[2], seg[3] / nw, segmax[3], seg[4] / nw, segmax[4], wv_ / nw, wb / nw, seg[5] / nw, segmax[5],
            seg[6] / nw, segmax[6], seg[7] / nw, segmax[7], seg[8] / nw, segmax[8], nw);
    fprintf(stderr, "[k_assign_wide3 prof]   setup in detail: first requests issued %.0f, block bounds (scalar loads) %.0f, cells in + Z requested %.0f, "
            "sigma/groups in + LDS filled %.0f, barrier %.0f, slots + barrier %.0f\n", fine[0] / nw, fine[1] / nw, fine[2] / nw, fine[3] / nw, fine[4] / nw, fine[5] / nw);
}
#else
static inline void wide3_prof_dump(int, hipStream_t) {}
static inline void sweep3_prof_dump(int, int, hipStream_t) {}
#endif

template <int MT, int KS, bool BF3T>
__global__ __launch_bounds__(ROUND_THREADS) void k_round(RoundArgs a) {
    if (a.frozen && *a.frozen) return;   // an earlier sweep of this cluster() call timed out: R, O, the objective block stay as it left them
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K16 = 16 * MT;
    constexpr int NF = KS / 4, NT = KS % 4;
    // Two ways of dealing a block's tiles out.  CLASSIC: tile pair p goes to workgroup p % nwg -- a workgroup's tiles are
    // spread over the whole group-sorted block, so it keeps the diversity table of ALL groups (G x K16 entries), publishes
    // sums for all of them with returning fp64 adds and arrives at a counter.  GROUP-AFFINE (a.ga; one batch variable, one
    // engine): workgroup w owns ONE group g(w) (host table wg_map, workgroup counts in proportion to the groups' sizes) and
    // takes its tiles from that group's run inside every block (run_start), so the per-block hand-off shrinks from G x K16
    // entries to K16 of its own group plus the K16 cluster masses T_k = sum_g O[g][k] (:491), which travel as row G of the
    // slot tables.  Its entries are SELF-VALIDATING: a workgroup adds  (1 << 55) + sum * 2^32  to every entry of its two rows
    // with one 64-bit integer add, fire and forget; a reader that finds the count fields of an entry's four slots adding up
    // to the number of contributors holds the complete sum -- no returning atomics, no vmcnt(0) in front of an arrival, no
    // arrival counter, no poll of a line that carries read-modify-writes of 200 workgroups, and the poll that succeeds has
    // brought the data along: one memory round trip where the classic hand-off has four (measured: profiles/r06_ab_k_round_ga_*).
    const bool ga = a.ga != 0;
    const int Gl = ga ? 1 : a.G;                       // rows of the workgroup's LDS tables
    const int GK = Gl * K16;                           // entries of the LDS tables
    const int GKg = a.G * K16;                         // entries of the global tables O_start, S_old, O_out
    const int GKs = (a.G + 1) * K16;                   // stride of a slot table / of a rank's share of a peer box (row G: cluster mass, group-affine map only)
    const int LDY = a.ldy_lds;
    constexpr bool LOG2 = HMX_ROUND_EXP2 != 0, A2TAB = HMX_ROUND_A2TAB != 0;
    constexpr bool BF3 = BF3T && LOG2;                                   // distance GEMM on the bf16 pipe (round_compute_bf3)
    float* Ys0 = reinterpret_cast<float*>(smem);                         // K16 x LDY, or the three bf16 planes of the centroids
    unsigned* Yb0 = reinterpret_cast<unsigned*>(smem);
    // landing zones of the waves' Z_cos rows (global -> LDS directly): waves x ROUND_TPW tiles x 16 rows x dp floats,
    // 16-byte aligned (LDY is a multiple of 4); a plain offset from the LDS base: a pointer rebuilt from an integer
    // would turn every read of the zones into a flat load, which waits on the memory counter as well
    float* zbuf = Ys0 + (BF3 ? (size_t)3 * K16 * bf3_ldb(KS) : (size_t)K16 * LDY);
    constexpr int ZTILE = 16 * (4 * KS);
    float* sig0 = zbuf + (size_t)ROUND_WAVES * ROUND_TPW * ZTILE;         // K16
    float* nis0 = sig0 + K16;                                            // K16  -1/sigma (-1e30 for pads)
    float* sig = sig0;
    float* nis = nis0;
    float* rpT = nis + K16;                                              // Gl x K16
    float* lrpT = rpT + GK;                                              // Gl x K16
    float* rpc = lrpT + GK;                                              // B x K16 powered ratios (several batch variables only)
    double* Ocur = reinterpret_cast<double*>(rpc + (a.V == 1 ? 0 : (size_t)K16 * a.B));   // Gl x K16 (all offsets so far are even)
    double* Sd = Ocur + GK;                                              // Gl x K16 this block's new sums
    double* Tm = Sd + GK;                                                // K16 cluster mass
    double* objw = Tm + K16;                                             // waves x 2
    float* prb = reinterpret_cast<float*>(objw + 2 * ROUND_WAVES);       // B
    float* tht = prb + a.B;                                              // B
    int* gcol = reinterpret_cast<int*>(tht + a.B);                       // G x V
    int* bgrp = gcol + a.G * a.V;                                        // B: the group holding batch b (V == 1)
    int* bsS = bgrp + a.B;                                               // nblk + 3: first tile of the workgroup's run in block b (two sentinels)
    int* bsN = bsS + a.nblk + 3;                                         // nblk + 3: tiles of that run (classic: the whole block)
    int* wgfail = bsN + a.nblk + 3;                                      // group-affine map: the hand-off wave's wait gave up

    int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the wave's landing zones and roles are wave-uniform
    int lane = tid & 63;
    int c16 = lane & 15, q = lane >> 4;   // (refreshed per block, see the sweep loop)
    const bool multi = a.n_ranks > 1;
    const int wg = blockIdx.x, nwg = multi ? gridDim.x - 1 : gridDim.x;   // compute workgroups
    unsigned long long* my_flags = multi ? reinterpret_cast<unsigned long long*>(a.my_box + box_flags(a.n_ranks, GKs)) : nullptr;

    if (multi && wg == nwg) {
        // ---- gateway workgroup: once every local workgroup has added its sums of block b, write the
        //      rank's total into every rank's box (xGMI peer writes), then raise this rank's flag there.
        bool gfail = false;
        for (int b = 0; b < a.nblk; ++b) {
            if (wv == 0 && !gfail) {
                const unsigned want = (unsigned)(b + 1) * (unsigned)nwg;
                unsigned spins = 0;
                while (ld_agent(a.counter) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) { gfail = true; break; }
                }
            }
            __syncthreads();
            for (int i = tid; i < GKg; i += ROUND_THREADS) {
                const double* sn = a.S_new + (size_t)b * HMX_ROUND_SLOTS * GKs + i;
                double v = 0.0;
#pragma unroll
                for (int s = 0; s < HMX_ROUND_SLOTS; ++s) v += ld_agent(sn + (size_t)s * GKs);
#if HMX_ROUND_RETURNING
                for (int r = 0; r < a.n_ranks; ++r) xchg_sys(a.peer_box[r] + box_data(a.n_ranks, GKs, b & 1, a.rank) + i, v);
#else
                for (int r = 0; r < a.n_ranks; ++r) st_sys(a.peer_box[r] + box_data(a.n_ranks, GKs, b & 1, a.rank) + i, v);
#endif
            }
            WAIT_VMEM_ALL();
            __syncthreads();
            if (tid < a.n_ranks)
                st_sys(reinterpret_cast<unsigned long long*>(a.peer_box[tid] + box_flags(a.n_ranks, GKs)) + (size_t)(b & 1) * a.n_ranks + a.rank,
                       a.epoch + (unsigned long long)b + 1ull);
        }
        if (gfail && tid == 0) {
            atomicExch(a.error, 1u); if (a.frozen) atomicExch(a.frozen, 1u);
            atomicAdd(&a.obj[0], __builtin_nan(""));
            atomicAdd(&a.obj[2 * HMX_OBJ_SLOTS + 1], 1.0);   // every rank sees the count of failures in the all-reduce of the objective block
        }
        return;
    }

    for (int i = tid; i < a.B; i += ROUND_THREADS) {
        prb[i] = a.Pr_b[i];
        tht[i] = a.theta[i];
    }
    for (int i = tid; i < a.G * a.V; i += ROUND_THREADS) {
        gcol[i] = a.group_cols[i];
        if (a.V == 1) bgrp[a.group_cols[i]] = i;
    }
    // the workgroup's map: group-affine (its group, its rank among the group's workgroups, their number) or classic
    const int g_own = ga ? a.wg_map[3 * wg] : 0;
    const int wl = ga ? a.wg_map[3 * wg + 1] : wg;
    const int ng = ga ? a.wg_map[3 * wg + 2] : nwg;
    const int n_tiles_all = a.blk_start[a.nblk];
    for (int i = tid; i < a.nblk + 3; i += ROUND_THREADS) {
        if (i >= a.nblk) { bsS[i] = n_tiles_all; bsN[i] = 0; }
        else if (ga) { bsS[i] = a.run_start[i * a.G + g_own]; bsN[i] = a.run_start[i * a.G + g_own + 1] - bsS[i]; }
        else { bsS[i] = a.blk_start[i]; bsN[i] = a.blk_start[i + 1] - bsS[i]; }
    }
    if (tid == 0) *wgfail = 0;
    constexpr float TWO_LOG2E = 2.885390081777926814f;
    for (int i = tid; i < K16; i += ROUND_THREADS) {
        const float sgm = (i < a.K) ? a.sigma[i] : 0.f;
        sig[i] = sgm;
        if (LOG2) nis[i] = (i < a.K) ? -(TWO_LOG2E / sgm) : -200.f;   // the accumulators start here; pads: Y row 0 -> 2^-200 == 0
        else nis[i] = (i < a.K) ? -1.0f / sgm : -60.f;               // pads: Y row 0 -> dist 2 -> arg -120 -> exp == 0
    }
    for (int i = tid; i < GK; i += ROUND_THREADS) {
        Ocur[i] = a.O_start[(ga ? g_own * K16 : 0) + i];
        Sd[i] = 0.0;
    }
    if (ga)
        for (int k = tid; k < K16; k += ROUND_THREADS) {
            double t = 0.0;
            for (int g = 0; g < a.G; ++g) t += a.O_start[(size_t)g * K16 + k];
            Tm[k] = t;
        }
    if (BF3) {
        constexpr int NP = 8 * bf3_steps(KS), LDB = bf3_ldb(KS);     // pieces per row incl. the zero ones past the row
        for (int i = tid; i < K16 * NP; i += ROUND_THREADS) {
            const int row = i / NP, p = i - row * NP;
            f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < KS && row < a.K) y = ld4(a.Y + (size_t)row * a.ldy + 4 * p) * (TWO_LOG2E / a.sigma[row]);
            unsigned h0, m0, l0, h1, m1, l1;
            bf16_split3((f32x2){y[0], y[1]}, h0, m0, l0);
            bf16_split3((f32x2){y[2], y[3]}, h1, m1, l1);
            unsigned* dst = Yb0 + (size_t)row * LDB + 2 * p;
            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(dst + (size_t)K16 * LDB) = (u32x2){m0, m1};
            *reinterpret_cast<u32x2*>(dst + (size_t)2 * K16 * LDB) = (u32x2){l0, l1};
        }
    } else
    for (int i = tid; i < K16 * KS; i += ROUND_THREADS) {
        const int row = i / KS, c4 = i - row * KS;
        f32x4 y = ld4(a.Y + (size_t)row * a.ldy + 4 * c4);
        if (LOG2) y *= (row < a.K) ? TWO_LOG2E / a.sigma[row] : 0.f;   // rows scaled by c_k: the products are exp2 arguments
        st4(Ys0 + (size_t)row * LDY + 4 * c4, y);
    }
    __syncthreads();
    (void)NF; (void)NT;
    // The grid is sized for ~14 of a workgroup's 16 tile slots, so the last wave of a workgroup never owns a tile: it becomes
    // the PUBLISHER -- after the finishing pass it alone adds the workgroup's block sums to the slot table and arrives.  Its
    // vmcnt(0) covers only its own (returning) adds, whereas a wave that has just written 14 R rows sees its adds
    // acknowledged behind those stores (memory operations retire in order): the arrival no longer waits for the block's row
    // stores, and the other waves go straight on to the next distance GEMM.  Only when that wave owns no tile in ANY block
    // and the table is SMALL (at most four adds per lane): measured on one box (profiles/r04_ab_k_round_publisher_wave.txt),
    // configs[1] (G K16 = 128) 129.5-131.8 -> 126.7 us per sweep, but C3 (896 entries: 14 adds per lane from ONE wave, behind
    // the other waves' row stores in the CU's memory pipeline) 322 -> 390 us -- there everybody publishes as before.
    bool pubwave = false;
    if (HMX_ROUND_PUBWAVE && GK <= 256 && !ga) {
        int max_ntl = 0;
        for (int bb = 0; bb < a.nblk; ++bb) max_ntl = max(max_ntl, bsN[bb]);
        pubwave = ROUND_TPW * (blockIdx.x + (multi ? (int)gridDim.x - 1 : (int)gridDim.x) * (ROUND_WAVES - 1)) >= max_ntl;
    }
    const bool service = pubwave && wv == ROUND_WAVES - 1;        // wave-uniform
    // group-affine: the last wave runs the hand-off (table at the top of a block, publish behind its finishing pass); the host
    // plans the map for 14 tiles per workgroup and block where the grid allows it, so that this wave seldom carries tiles
    const bool hand = ga && wv == ROUND_WAVES - 1;

    RoundTile<MT> T[ROUND_TPW];
    float* zb[ROUND_TPW];
#pragma unroll
    for (int u = 0; u < ROUND_TPW; ++u) zb[u] = zbuf + (size_t)(wv * ROUND_TPW + u) * ZTILE;
    int cell1[ROUND_TPW], grp1[ROUND_TPW];   // ids of block b+1's tiles (landed)
    int cell2[ROUND_TPW] = {0, 0}, grp2[ROUND_TPW] = {0, 0};   // ids of block b+2's tiles (travelling: raw loads, untouched until they are shifted in)
    bool valid2[ROUND_TPW] = {false, false};
    double km_acc = 0.0, ent_acc = 0.0;
    bool failed = false;
    unsigned ws_n = 0, ws_sum = 0, ws_max = 0;   // this workgroup's grid-wide waits (wave 0; group-affine: the hand-off wave)
    // classic: tiles 2p, 2p+1 of a block go to workgroup p % nwg, wave (p / nwg) % WAVES (pass p / nwg / WAVES);
    // group-affine: the same deal among the ng workgroups of the group over its run
    const int j_first = ROUND_TPW * (wl + ng * wv);
    const int j_slot = ROUND_TPW * ng * ROUND_WAVES;
    auto load_ids = [&](int blk, int u, int& cell, int& grp) {   // blk may run past the last block: bsS / bsN have sentinels
        const int j = j_first + u;
        const bool valid = j < bsN[blk];
        cell = valid ? a.cells[(size_t)(bsS[blk] + j) * 16 + c16] : -1;
        grp = (valid && !ga) ? a.tile_grp[bsS[blk] + j] : 0;    // (group-affine: the LDS tables hold the workgroup's own group only)
    };

    // A tile's 16 Z_cos rows travel global -> LDS directly (no destination registers: held in VGPRs across the finishing
    // pass they push it into scratch, and every scratch reload waits for ALL outstanding memory operations of the wave).
    // The landing zone is linear in 16-byte pieces: lane l of instruction i brings piece (64 i + l) % KS of row
    // (64 i + l) / KS, so the zone is the tile row-major.  Dead rows (list padding) read cell 0's row.
    auto issue_rows = [&](int cell_c16, float* dst) {
        if (BF3) {
            // whole rows per request: lane l brings piece l % KS of row RPD it + l / KS
            constexpr int RPD = zone_rows_per_request(KS), NIT = 16 / RPD;
            static_assert(16 % RPD == 0, "whole rows per request");
            const int r_in = lane / KS, piece = lane - r_in * KS;
            const float* src[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cell = __builtin_amdgcn_ds_bpermute(4 * (RPD * it + r_in), cell_c16);   // the rows' ids sit in lanes 0..15
                src[it] = a.Zcos + (size_t)(cell >= 0 ? cell : 0) * (4 * KS) + 4 * piece;
            }
            if (lane < RPD * KS) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const unsigned zone = __builtin_amdgcn_readfirstlane(
                        (unsigned)(size_t)(__attribute__((address_space(3))) void*)(dst + RPD * (4 * KS) * it));
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                                 :: "v"(src[it]), "s"(zone) : "memory", "m0");
                }
            }
            return;
        }
        constexpr int NCH = 16 * KS, NIT = (NCH + 63) / 64;
        const float* src[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {   // all source addresses first: one wait for the id exchange, not one per request
            const int idx = 64 * it + lane;
            const int row = idx / KS, piece = idx - row * KS;
            const int cell = __shfl(cell_c16, row & 15, 64);   // the rows' ids sit in lanes 0..15
            src[it] = a.Zcos + (size_t)(cell >= 0 ? cell : 0) * (4 * KS) + 4 * piece;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // Issued as inline assembly on purpose: after the builtin the compiler waits for vmcnt(0) before the next
            // LDS read of ANY address (it cannot tell the landing zone from the centroid table), i.e. for the rows.
            // The waits it computes for its own loads stay correct: memory operations return in order.
            const unsigned zone = __builtin_amdgcn_readfirstlane(
                (unsigned)(size_t)(__attribute__((address_space(3))) void*)(dst + 256 * it));
            if (64 * (it + 1) <= NCH || 64 * it + lane < NCH)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                             :: "v"(src[it]), "s"(zone) : "memory", "m0");
        }
    };
    const float* Ys = Ys0;
    const unsigned* Yb = Yb0;
    // ---- prologue: block 0 computed, block 1's ids landed ---------------------------------------
#pragma unroll
    for (int u = 0; u < ROUND_TPW; ++u) {
        load_ids(0, u, T[u].cell, T[u].grp);
        load_ids(1, u, cell1[u], grp1[u]);
    }
    // tile step shared by the prologue and the sweep: fragments of the landed rows into registers, the landing zones
    // handed to the rows of the block after (ids in cell1; ids of the block after that requested too), then the distance
    // GEMM.  The requests are issued here because this is where they are free -- the SIMD's matrix pipe is the bound of
    // this phase, the partner wave's MFMAs cover them -- and far from the hand-off: a wave's memory operations return
    // in order, so a request in flight in front of the poll or of the hand-off's loads would be waited for by them.
    // By the next poll these rows have had a whole distance GEMM to land; they are used a block later.
    int prof_b = -1;   // (profiling build: the block whose record the tile step's stamps go to)
    (void)prof_b;
    auto tile_step = [&](int blk_ids) {
        // a wave without a tile in the block it is about to multiply (the last wave of most workgroups, every block) only
        // keeps its id pipeline going: no rows, no GEMM on dummy operands -- it is the wave that arrives for the workgroup
        const bool live = j_first < bsN[blk_ids - 2];   // wave-uniform
        RoundZ<KS> Zf[ROUND_TPW];
        f32x4 raw[ROUND_TPW][2 * bf3_steps(KS)];
        if (live) {
#pragma unroll
        for (int u = 0; u < ROUND_TPW; ++u) {
            if (BF3) round_raw_pieces<KS>(zb[u] + c16 * (4 * KS), q, raw[u]);
            else round_rows_from_lds<KS>(zb[u], c16, q, Zf[u]);
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zones are read before they are handed on
        auto request = [&]() {
#pragma unroll
        for (int u = 0; u < ROUND_TPW; ++u) {
            if (j_first + u < bsN[blk_ids - 1]) issue_rows(cell1[u], zb[u]);   // (wave-uniform: the tile exists)
            // ids of the block after: clamped loads, the "no such tile" case applied when they are shifted in -- a
            // predicated load or a select here would be waited for at once, and with it every row request above
            const int j = j_first + u;
            const int t0 = bsS[blk_ids];
            valid2[u] = j < bsN[blk_ids];
            const int tl = max(min(t0 + j, n_tiles_all - 1), 0);   // (a shard may hold no tile at all)
            cell2[u] = a.cells[(size_t)tl * 16 + c16];
            grp2[u] = a.tile_grp[tl];
        }
        };
        // the two waves of a SIMD stagger the request code: one issues it while the other's first tile keeps the pipe busy
        static_assert(ROUND_TPW == 2, "tile_step is written for two tiles per wave");
        // Where a wave issues its row requests.  Classic map (req_mode 0): half the waves in front of their GEMM, half between
        // its k-steps -- never near the hand-off, whose poll and loads would wait behind them.  Under the group-affine map only
        // the last wave polls, and the GEMM starts while the whole chip's row stores of the block are still draining (nobody
        // waits for returning adds any more), so the best place depends on the shape (same-box A/B, profiles/r06_ab_k_round_requests.txt):
        // 1: BEHIND the GEMM -- workgroups whose eight waves all carry tiles (configs[3] shard: 401-412 -> 326-332 us);
        // 2: every wave between the k-steps -- large grids with the last wave free (C3: 247-255 -> 233-250 us); 0: small grids.
        const bool late = a.req_mode == 1 && ga && !hand;
        const bool second = (a.req_mode == 2 && ga) || wv >= ROUND_WAVES / 2;
        TSTAMP(11);
        if (!second && !late) request();
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(12);
        if (!live) {
            if (second || late) request();
            return;
        }
        if (BF3) {
            // both tiles against one read of the centroid fragments; the `second` waves' requests ride between the k-steps
            round_compute_bf3_pair<MT, KS>(Yb, nis, c16, q, raw[0], raw[1], T[0], T[1], [&]() { TSTAMP(14); if (second && !late) request(); }, [&]() { TSTAMP(13); });
            if (late) request();
            return;
        }
        round_compute<MT, KS, LOG2>(Ys, nis, LDY, c16, q, Zf[0], T[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (second && !late) request();
        __builtin_amdgcn_sched_barrier(0);
        round_compute<MT, KS, LOG2>(Ys, nis, LDY, c16, q, Zf[1], T[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (late) request();
    };
    // ---- grid-wide wait: every workgroup has added its sums of block `bdone` (one wave; its lanes are all active) --------
    auto wait_block = [&](int bdone) {
        if (failed) return;          // (a wait that gave up is not repeated block after block: the launch is lost)
        unsigned spins = 0;
        if (a.spin_limit == 0) failed = true;   // test knob: give up without looking
        if (failed) {
        } else if (!multi) {
            const unsigned want = (unsigned)(bdone + 1) * (unsigned)nwg;
            while (ld_agent(a.counter) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) { failed = true; break; }
            }
        } else {   // every rank's total of the block has landed in this rank's box
            const unsigned long long want = a.epoch + (unsigned long long)bdone + 1ull;
            const unsigned long long* fl = my_flags + (size_t)(bdone & 1) * a.n_ranks;
            while (true) {
                const bool ok = lane >= a.n_ranks || ld_sys(fl + lane) >= want;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) { failed = true; break; }
            }
        }
        ws_n += 1; ws_sum += spins; ws_max = max(ws_max, spins);
#ifdef HMX_ROUND_PROF
        if (lane == 0 && a.prof) a.prof[((size_t)wg * a.nblk + bdone + 1) * 32 + 10] = spins;
#endif
    };
    // ---- group-affine hand-off, run by the workgroup's last wave; a lane owns the table entries k = lane and lane + 64 -------
    // The sums all workgroups added for block bp, per own entry: own group's row so[n], cluster-mass row st[n].  Polls until
    // every entry's count fields (bits 55.. of the four slot words) add up to the number of contributors -- ng workgroups of
    // the own group, all nwg for the cluster masses; the words that pass the test ARE the data (2^-32 fixed point).
#ifndef HMX_GA_SLOTS
#define HMX_GA_SLOTS 2   /* slot tables the group-affine map spreads a block's adds over AT MOST (of the HMX_ROUND_SLOTS allocated; a.ga_slots of them are
                            used): every slot is two more loads per entry of the poll; 8 / 4 / 2 / 1 measured (profiles/r06_ab_k_round_slots_poll.txt):
                            contention on the words only costs on large grids (one slot: C3 294 vs 242 us, configs[1] 110.5 vs 112 us) */
#endif
    static_assert(HMX_GA_SLOTS <= HMX_ROUND_SLOTS, "the slot tables are allocated for HMX_ROUND_SLOTS");
    constexpr unsigned long long FX_MASK = (1ull << 55) - 1ull;
    auto ga_fetch = [&](int bp, double (&so)[2], double (&st)[2]) {
        so[0] = so[1] = st[0] = st[1] = 0.0;
        if (bp < 0) return;
        unsigned long long wo[2][HMX_GA_SLOTS], wt[2][HMX_GA_SLOTS];
        unsigned spins = 0;
        if (a.spin_limit == 0) failed = true;   // test knob: give up without looking
        while (true) {
            bool ok = true;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int k = min(lane + 64 * n, K16 - 1);                   // clamped, not predicated
                const unsigned long long* sn = reinterpret_cast<const unsigned long long*>(a.S_new) + (size_t)bp * HMX_ROUND_SLOTS * GKs + k;
#pragma unroll
                for (int s = 0; s < HMX_GA_SLOTS; ++s) {                  // independent loads, all in flight together
                    wo[n][s] = wt[n][s] = 0ull;
                    if (s < a.ga_slots) {                                    // (uniform: small grids use one slot table)
                        wo[n][s] = __hip_atomic_load(sn + (size_t)s * GKs + (size_t)g_own * K16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wt[n][s] = __hip_atomic_load(sn + (size_t)s * GKs + (size_t)a.G * K16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                unsigned co = 0, ct = 0;
#pragma unroll
                for (int s = 0; s < HMX_GA_SLOTS; ++s) { co += (unsigned)(wo[n][s] >> 55); ct += (unsigned)(wt[n][s] >> 55); }
                ok = ok && co == (unsigned)ng && ct == (unsigned)nwg;
            }
            if (__all(ok) || failed) break;
#ifndef HMX_GA_POLL_SLEEP
#define HMX_GA_POLL_SLEEP 2
#endif
            __builtin_amdgcn_s_sleep(HMX_GA_POLL_SLEEP);
            if (++spins > a.spin_limit) { failed = true; break; }   // (a wait that gave up is not repeated block after block: the launch is lost)
        }
        ws_n += 1; ws_sum += spins; ws_max = max(ws_max, spins);
#ifdef HMX_ROUND_PROF
        if (lane == 0 && a.prof) a.prof[((size_t)wg * a.nblk + min(bp + 1, a.nblk - 1)) * 32 + 10] = spins;
#endif
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int s = 0; s < HMX_GA_SLOTS; ++s) {
                so[n] += (double)(long long)(wo[n][s] & FX_MASK) * 0x1p-32;
                st[n] += (double)(long long)(wt[n][s] & FX_MASK) * 0x1p-32;
            }
    };
    // O of the own group and the cluster mass T without block b's old sums and with block b-1's new ones (:491-492,
    // 506-507), then ratio ** theta and its log for block b (:495-499): one round trip of loads, one power chain.
    auto ga_table = [&](int b) {
        double ro[2], rt[2], ao[2], at[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int k = min(lane + 64 * n, K16 - 1);
            const double* sold = a.S_old + (size_t)b * GKg + k;         // constant during the launch: plain loads
            ro[n] = sold[(size_t)g_own * K16];
            double t = 0.0;
            for (int g0 = 0; g0 < a.G; g0 += 8) {                       // eight loads in flight at a time
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = sold[(size_t)min(g0 + j, a.G - 1) * K16];
#pragma unroll
                for (int j = 0; j < 8; ++j) t += (g0 + j < a.G) ? v[j] : 0.0;
            }
            rt[n] = t;
        }
        ga_fetch(b - 1, ao, at);
        const int bo = gcol[g_own];                                      // the group's batch (one batch variable)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int k = lane + 64 * n;
            if (k < K16) {
                const double o = Ocur[k] - ro[n] + ao[n], t = Tm[k] - rt[n] + at[n];
                Ocur[k] = o;
                Tm[k] = t;
                const float O = (float)o;
                const float E = (float)t * prb[bo];                         // :491 (E kept as mass T)
                const float oe = fmaxf(O + E, 1e-8f);                       // :495-496
                const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);      // :497-498
                const float rp = pow_unit(ratio, tht[bo]);                  // :499
                rpT[k] = rp;
                lrpT[k] = __builtin_amdgcn_logf(rp) * 0.693147182464599609375f;   // v_log_f32 (log2, 1 ulp) * ln 2
            }
        }
    };
    // the workgroup's sums of block b into its group's row and into the cluster-mass row of one slot table: count and sum in
    // one word, not returning, nothing to wait for (see ga_fetch).  EVERY entry is added to, zero sums too: the count is the arrival.
    auto ga_publish = [&](int b) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.S_new) + ((size_t)b * HMX_ROUND_SLOTS + (wg % a.ga_slots)) * GKs;
        double a2s = 0.0;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int k = lane + 64 * n;
            if (k < K16) {
                const double v = Sd[k];
                Sd[k] = 0.0;
                if (A2TAB) a2s += v * (double)(lrpT[k] * sig[k]);       // (:402), see round_post_pass1
                const unsigned long long w = (1ull << 55) + (unsigned long long)__double2ll_rn(v * 4294967296.0);
                __hip_atomic_fetch_add(dst + (size_t)g_own * K16 + k, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(dst + (size_t)a.G * K16 + k, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (A2TAB) ent_acc += a2s;
    };

    if (!service) {
#pragma unroll
        for (int u = 0; u < ROUND_TPW; ++u) issue_rows(T[u].cell, zb[u]);
        WAIT_VMEM_ALL();   // landed (nothing else orders an LDS read behind an LDS-DMA)
        tile_step(2);
    }

    for (int b = 0; b < a.nblk; ++b) {
        const int tb = bsS[b], ntl = bsN[b];
        {   // the tables do not change, but re-reading them every block is cheaper than the ~150
            // registers the compiler would spend keeping their fragments live across the sweep
            int zero = 0;
            asm volatile("" : "+v"(zero));
            Ys = Ys0 + zero;
            Yb = Yb0 + zero;
            sig = sig0 + zero;
            nis = nis0 + zero;
            // likewise the lane's own coordinates: everything derived from them (table and row addresses, list
            // positions) is recomputed per block instead of being kept -- in scratch -- across the sweep
            asm volatile("" : "+v"(tid), "+v"(lane), "+v"(c16), "+v"(q));
        }
        RSTAMP(0);
        RSTAMP(8);
        if (ga) {
            RSTAMP(9); RSTAMP(1); RSTAMP(6); RSTAMP(7);
            if (hand) ga_table(b);   // polls for block b-1's sums, folds them, builds block b's table; the other waves wait at the barrier
        }
        if (!ga) {
        // ---- wait until every workgroup has added its sums of block b-1 ---------------------
        if (b > 0 && wv == 0) wait_block(b - 1);
        RSTAMP(9);
        wg_barrier_lds();
        RSTAMP(1);
        // ---- O without this block, with the previous block's new sums (:491-492, 506-507) ---
        for (int base = tid; base < GK; base += 2 * ROUND_THREADS) {   // two entries per thread and trip, all loads in flight together
            double add[2][8], so[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i = min(base + n * ROUND_THREADS, GK - 1);    // clamped, not predicated
#pragma unroll
                for (int s = 0; s < 8; ++s) add[n][s] = 0.0;
                so[n] = a.S_old[(size_t)b * GKg + i];
                if (b > 0 && !(HMX_RABL & 8)) {
                    if (!multi) {
                        const double* sn = a.S_new + (size_t)(b - 1) * HMX_ROUND_SLOTS * GKs + i;
#pragma unroll
                        for (int s = 0; s < HMX_ROUND_SLOTS; ++s) add[n][s] = ld_agent(sn + (size_t)s * GKs);   // independent loads
                    } else {
                        const double* bx = a.my_box + box_data(a.n_ranks, GKs, (b - 1) & 1, 0) + i;
#pragma unroll
                        for (int s = 0; s < 8; ++s) add[n][s] = ld_sys(bx + (size_t)min(s, a.n_ranks - 1) * GKs);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i = base + n * ROUND_THREADS;
                if (i < GK) {
                    double o = Ocur[i] - so[n];
#pragma unroll
                    for (int s = 0; s < 8; ++s) o += (!multi || s < a.n_ranks) ? add[n][s] : 0.0;
                    Ocur[i] = o;
                    Sd[i] = 0.0;
                }
            }
        }
        wg_barrier_lds();
        RSTAMP(6);
        // ---- cluster mass, then ratio ** theta per (batch, cluster)  (:491, 495-499) ----------
        if (a.V == 1) {
            // one batch variable (group g is batch g): mass, ratio, power and log in one pass over the table entries
            for (int i = tid; i < GK; i += ROUND_THREADS) {
                const int g = i / K16, k = i - g * K16;
                double t = 0.0;
                for (int gg = 0; gg < a.G; ++gg) t += Ocur[(size_t)gg * K16 + k];
                const float O = (float)Ocur[i];
                const float E = (float)t * prb[g];                          // :491 (E kept as mass T)
                const float oe = fmaxf(O + E, 1e-8f);                       // :495-496
                const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);      // :497-498
                const float rp = pow_unit(ratio, tht[g]);                   // :499
                rpT[i] = rp;
                lrpT[i] = __builtin_amdgcn_logf(rp) * 0.693147182464599609375f;   // v_log_f32 (log2, 1 ulp) * ln 2
            }
            RSTAMP(7);
        } else {
        for (int k = tid; k < K16; k += ROUND_THREADS) {
            double t = 0.0;
            for (int g = 0; g < a.G; ++g) t += Ocur[(size_t)g * K16 + k];
            Tm[k] = t;
        }
        wg_barrier_lds();
        for (int i = tid; i < K16 * a.B; i += ROUND_THREADS) {
            const int bb = i / K16, k = i - bb * K16;
            double Ob = 0.0;
            for (int g = 0; g < a.G; ++g) {
                bool has = false;
                for (int v = 0; v < a.V; ++v) has |= gcol[g * a.V + v] == bb;
                if (has) Ob += Ocur[(size_t)g * K16 + k];
            }
            const float O = (float)Ob;
            const float E = (float)Tm[k] * prb[bb];                     // :491 (E kept as mass T)
            const float oe = fmaxf(O + E, 1e-8f);                       // :495-496
            const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);      // :497-498
            rpc[i] = pow_unit(ratio, tht[bb]);                          // :499
        }
        wg_barrier_lds();
        RSTAMP(7);
        for (int i = tid; i < GK; i += ROUND_THREADS) {
            const int g = i / K16, k = i - g * K16;
            float s = 0.f;
            for (int v = 0; v < a.V; ++v) s += rpc[(size_t)gcol[g * a.V + v] * K16 + k];
            rpT[i] = s;                                                 // (ratio_pow @ Phi) for the cells of group g
            lrpT[i] = __builtin_amdgcn_logf(s) * 0.693147182464599609375f;   // v_log_f32 (log2, 1 ulp) * ln 2
        }
        }
        }   // (!ga)
        wg_barrier_lds();
        RSTAMP(2);
        // ---- finish this block's tiles --------------------------------------------------------
        if (j_first < ntl) {
            float scl0, scl1 = 0.f;
            const bool has1 = j_first + 1 < ntl;
            round_post_pass1<MT, true, LOG2, A2TAB>(sig, rpT, lrpT, q, T[0], scl0, km_acc, ent_acc);
            __builtin_amdgcn_sched_barrier(0);
            round_store_rows<MT>(a.R, a.Kp, q, T[0], scl0);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) round_post_pass1<MT, true, LOG2, A2TAB>(sig, rpT, lrpT, q, T[1], scl1, km_acc, ent_acc);
            __builtin_amdgcn_sched_barrier(0);
            round_post_pass2<MT, false>(a.R, a.Kp, Sd, c16, q, T[0], scl0, has1, T[1], scl1);
        }
        for (int j = j_first + j_slot; j < ntl; j += j_slot) {   // blocks larger than the grid carries
#pragma unroll 1
            for (int u = 0; u < ROUND_TPW; ++u) {
                if (j + u >= ntl) break;
                RoundTile<MT> X;
                X.cell = a.cells[(size_t)(tb + j + u) * 16 + c16];
                X.grp = ga ? 0 : a.tile_grp[tb + j + u];
                if (BF3) {
                    f32x4 xraw[2 * bf3_steps(KS)];
                    RoundZ3<KS> XZ3;
                    round_raw_pieces<KS>(a.Zcos + (size_t)(X.cell >= 0 ? X.cell : 0) * (4 * KS), q, xraw);
                    round_split_rows<KS>(xraw, XZ3);
                    round_compute_bf3<MT, KS>(Yb, nis, c16, q, XZ3, X);
                } else {
                    RoundZ<KS> XZ;
                    round_issue_z<KS>(a.Zcos, X.cell, q, XZ);
                    round_compute<MT, KS, LOG2>(Ys, nis, LDY, c16, q, XZ, X);
                }
                float sclx;
                round_post_pass1<MT, true, LOG2, A2TAB>(sig, rpT, lrpT, q, X, sclx, km_acc, ent_acc);
                round_post_pass2<MT>(a.R, a.Kp, Sd, c16, q, X, sclx, false, X, 0.f);
            }
        }
        wg_barrier_lds();
        RSTAMP(3);
        // ---- publish the block's new sums, then arrive ---------------------------------------
        if (ga) {
            if (hand) ga_publish(b);   // behind the workgroup's barrier: the block sums in LDS are complete
            WAIT_VMEM_ALL();           // this wave's row stores (and the next operands landed); nobody waits for the adds
        } else if (pubwave) {
            if (service) {
                double* dst = a.S_new + ((size_t)b * HMX_ROUND_SLOTS + (wg % HMX_ROUND_SLOTS)) * GKs;
                double a2s = 0.0;
                // all adds in flight together (each one's old value is only looked at behind the last: one round trip, not
                // four); returning, so that behind the wave's vmcnt(0) they are PERFORMED
                double olds[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = lane + 64 * t;
                    olds[t] = 0.0;
                    if (i < GK) {
                        const double v = Sd[i];
                        if (A2TAB) a2s += v * (double)(lrpT[i] * sig[i % K16]);   // (:402), see round_post_pass1
                        if (v != 0.0) olds[t] = atomicAdd(dst + i, v);
                    }
                }
                if (A2TAB) ent_acc += a2s;
                WAIT_VMEM_ALL();
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(olds[t]));
                if (lane == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                WAIT_VMEM_ALL();   // this wave's row stores (and the next operands landed); nobody waits for it
            }
        } else {
        {
            double* dst = a.S_new + ((size_t)b * HMX_ROUND_SLOTS + (wg % HMX_ROUND_SLOTS)) * GKs;
            double a2s = 0.0;
            for (int i = tid; i < GK; i += ROUND_THREADS) {
                double v = Sd[i];
                if (A2TAB) a2s += v * (double)(lrpT[i] * sig[i % K16]);   // sum over this block's cells of R sigma log(ratio^theta) (:402), see round_post_pass1
#if HMX_ROUND_RETURNING
                if (v != 0.0) { const double old = atomicAdd(dst + i, v); asm volatile("" ::"v"(old)); }
#else
                if (v != 0.0) atomicAdd(dst + i, v);
#endif
            }
            if (A2TAB) ent_acc += a2s;
        }
        WAIT_VMEM_ALL();   // the sums are performed (and the next operands landed)
        wg_barrier_lds();
        // (Wave 0 arrives.  Its next vector-memory wait -- the first use of the next tiles' ids, a loop-carried load the
        // compiler drains the counter for -- therefore covers this add's round trip, ~2 k cycles in front of its distance
        // GEMM; letting the tile-less last wave arrive instead was SLOWER on one box, 303 vs 296 us per sweep at C3:
        // profiles/r04_ab_k_round_bf16_pipe.txt)
        if (tid == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        RSTAMP(4);
        // ---- table-independent half of the next block's tiles: overlaps the hand-off ----------
#pragma unroll
        for (int u = 0; u < ROUND_TPW; ++u) {
            T[u].cell = cell1[u];
            T[u].grp = grp1[u];
            cell1[u] = valid2[u] ? cell2[u] : -1;
            grp1[u] = (valid2[u] && !ga) ? grp2[u] : 0;
        }
        prof_b = b;
        if (b + 1 < a.nblk && !service) tile_step(b + 3);   // rows landed: vmcnt(0) above
        RSTAMP(5);
    }

    if (lane == 0 && wv == (ga ? ROUND_WAVES - 1 : 0) && a.wait_stats) {   // (the wave that waits)
        atomicAdd(a.wait_stats, (unsigned long long)ws_n);
        atomicAdd(a.wait_stats + 1, (unsigned long long)ws_sum);
        atomicMax(a.wait_stats + 2, (unsigned long long)ws_max);
    }
    if (hand && failed && lane == 0) *wgfail = 1;
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
        for (int w = 0; w < ROUND_WAVES; ++w) v += objw[2 * w + tid];
        if (v != 0.0) atomicAdd(&a.obj[2 * (wg & (HMX_OBJ_SLOTS - 1)) + tid], v);
    }
    if (tid == 0 && (failed || (ga && *wgfail))) {
        atomicExch(a.error, 1u); if (a.frozen) atomicExch(a.frozen, 1u);
        atomicAdd(&a.obj[0], __builtin_nan(""));
        atomicAdd(&a.obj[2 * HMX_OBJ_SLOTS + 1], 1.0);   // the host (every rank's, after the all-reduce) replays the round block by block
    }
    auto give_up = [&]() {
        if (lane == 0) { atomicExch(a.error, 1u); if (a.frozen) atomicExch(a.frozen, 1u); atomicAdd(&a.obj[0], __builtin_nan("")); atomicAdd(&a.obj[2 * HMX_OBJ_SLOTS + 1], 1.0); }
    };
    auto wait_last = [&]() {   // one wave: every workgroup (every rank) has added its sums of the last block
        unsigned spins = 0;
        if (!multi) {
            const unsigned want = (unsigned)a.nblk * (unsigned)nwg;
            while (ld_agent(a.counter) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) { give_up(); break; }
            }
        } else {
            const unsigned long long want = a.epoch + (unsigned long long)a.nblk;
            const unsigned long long* fl = my_flags + (size_t)((a.nblk - 1) & 1) * a.n_ranks;
            while (true) {
                const bool ok = lane >= a.n_ranks || ld_sys(fl + lane) >= want;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) { give_up(); break; }
            }
        }
    };
    if (ga) {
        // ---- group-affine: the first workgroup of every group closes its group's row of O, its share of the
        //      cross-entropy term (:405-411) and -- group 0's -- the cluster mass
        if (wl != 0 || !hand) return;
        const int bo = gcol[g_own];
        double part = 0.0;
        double ao[2], at[2];
        const bool failed_before = failed;
        ga_fetch(a.nblk - 1, ao, at);
        if (failed && !failed_before) give_up();
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int k = min(lane + 64 * n, K16 - 1);
            const double o = Ocur[k] + ao[n], t = Tm[k] + at[n];
            if (lane + 64 * n < K16) {
                a.O_out[(size_t)g_own * K16 + k] = o;
                if (g_own == 0) a.T_out[k] = t;
                const float O = (float)o;
                const float Oc = fmaxf(O, 1e-8f);                               // :407
                const float Ec = fmaxf((float)t * prb[bo], 1e-8f);              // :408
                const float tl = tht[bo] * logf((Oc + Ec) / Ec);                // :409-410
                part += (double)(sig[k] * O * tl);
            }
        }
        part = wave_sum_all(part);
        // sharded: the term comes from job-wide tables, identical on every rank, and the objective block is summed over
        // the ranks as a whole (it carries the failure count too): rank 0 contributes it
        if (lane == 0 && (!multi || a.rank == 0)) atomicAdd(&a.obj[2 * HMX_OBJ_SLOTS], part);
        return;
    }
    if (wg != 0) return;

    // ---- workgroup 0 closes the sweep: O, cluster mass, cross-entropy term (:405-411) -----------
    if (wv == 0) wait_last();
    __syncthreads();
    for (int i = tid; i < GK; i += ROUND_THREADS) {
        double o = Ocur[i];
        if (!multi) {
            const double* sn = a.S_new + (size_t)(a.nblk - 1) * HMX_ROUND_SLOTS * GKs + i;
#pragma unroll
            for (int s = 0; s < HMX_ROUND_SLOTS; ++s) o += ld_agent(sn + (size_t)s * GKs);
        } else {
            const double* bx = a.my_box + box_data(a.n_ranks, GKs, (a.nblk - 1) & 1, 0) + i;
            for (int s = 0; s < a.n_ranks; ++s) o += ld_sys(bx + (size_t)s * GKs);
        }
        Ocur[i] = o;
        a.O_out[i] = o;
    }
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < K16 * a.B; i += ROUND_THREADS) {
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
    if (lane == 0) objw[wv] = part;
    __syncthreads();
    if (tid == 0) {
        double v = 0.0;
        for (int w = 0; w < ROUND_WAVES; ++w) v += objw[w];
        // sharded: the term comes from job-wide tables, identical on every rank, and the objective block is summed over
        // the ranks as a whole (it carries the failure count too): rank 0 contributes it
        if (!multi || a.rank == 0) atomicAdd(&a.obj[2 * HMX_OBJ_SLOTS], v);
    }
}

// ------------------------------------------------------------------------------------------
// R^T . Z over a list of cells:  out[cluster][pc] += sum_cells R[cell][cluster] * Z[cell][pc]
//   - centroid numerator  Z_cos R^T        (harmony.py:443), whole round list, one output
//   - block removal sums  R_blk Phi_blk^T  (harmony.py:491-492) as per-(block, group) column sums
//   - ridge right-hand sides Phi_Rk Z_orig^T (harmony.py:556-563), one output per task
// A = R^T: lane holds R[cell 4ks+q][16mt + c16]; B = Z: lane holds Z[cell 4ks+q][16nt + c16].
// Each wave owns a contiguous tile range and a (MTW x NTW) block of output tiles selected
// by blockIdx.y; its accumulators go to a per-wave slab in fragment order (reduced later).
// ------------------------------------------------------------------------------------------
template <int MTW, int NTW>
__global__ __launch_bounds__(256, 2) void k_rtz(RtzArgs a) {
    const int lane = threadIdx.x & 63;
    const int c16 = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int sub = blockIdx.y;
    const int nsub_n = (a.ntd + NTW - 1) / NTW;
    const int mt0 = (sub / nsub_n) * MTW, nt0 = (sub % nsub_n) * NTW;
    const bool first_col_block = (sub % nsub_n) == 0;

    int t0, t1, task_grp = -1;
    if (a.task_tile0) {  // one task per wave (ridge statistics)
        if (wave >= a.ntasks) return;
        t0 = a.task_tile0[wave];
        t1 = a.task_tile1[wave];
        task_grp = a.task_grp[wave];
    } else {
        const int n_tiles = a.blk_start ? a.blk_start[a.nblk] : a.n_tiles;
        const int per = (n_tiles + nwaves - 1) / nwaves;
        t0 = min(wave * per, n_tiles);
        t1 = min(t0 + per, n_tiles);
    }

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double csum[MTW];   // four values per tile in fp32, tiles in fp64 (see k_rtz2)
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) csum[mt] = 0.0;
    int cur_g = -1, cur_b = 0;

    auto flush = [&]() {
        if (cur_g < 0 || !first_col_block || a.S_out == nullptr) return;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            double s = csum[mt];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const int k = 16 * (mt0 + mt) + c16;
            if (q == 0 && k < a.K && s != 0.0)
                atomicAdd(&a.S_out[((size_t)cur_b * a.G + cur_g) * a.K16 + k], s);
            csum[mt] = 0.0;
        }
    };

    for (int t = t0; t < t1; ++t) {
        const int g = a.tile_grp[t];
        int b = cur_b;
        if (a.blk_start) {
            while (t >= a.blk_start[b + 1]) ++b;  // lists are block-major, so b only grows
        }
        if (g != cur_g || b != cur_b) {
            flush();
            cur_g = g;
            cur_b = b;
        }
        float tsum[MTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) tsum[mt] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cell = a.cells[(size_t)t * 16 + 4 * ks + q];
            float av[MTW], bv[NTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int col = 16 * (mt0 + mt) + c16;
                av[mt] = (cell >= 0 && col < a.Kp) ? a.R[(size_t)cell * a.Kp + col] : 0.f;
                tsum[mt] += av[mt];
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int col = 16 * (nt0 + nt) + c16;
                bv[nt] = (cell >= 0 && col < a.dp) ? a.Z[(size_t)cell * a.dp + col] : 0.f;
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA16(av[mt], bv[nt], acc[mt][nt]);
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) csum[mt] += (double)tsum[mt];
    }
    flush();

    // slab in fragment order: [wave][sub][mt][nt][r][lane]
    float* slab = a.slab + ((size_t)wave * gridDim.y + sub) * (MTW * NTW * 256);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[((mt * NTW + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    (void)task_grp;
}

// ------------------------------------------------------------------------------------------
// k_rtz2: the same product for shapes whose rows fit the LDS comfortably (K <= 112, d <= 64),
// built for HBM streaming instead of latency:
//   * a workgroup of 4 waves owns ALL output tiles (wave w: PC column block w % NTD, cluster tiles
//     w / NTD, + 4/NTD, ..) and walks a contiguous range of tiles (or one task);
//   * a tile's 16 R rows and 16 Z rows are fetched with 16-byte loads (row-contiguous, so every
//     cache line is used whole), staged through a double-buffered LDS tile whose row strides
//     (= 16 mod 32 floats) make the fragment reads conflict-free, one workgroup barrier per tile;
//   * the loads of tile t+2 and the cell ids of tile t+3 are in flight while tile t is multiplied;
//   * two workgroups per CU (512 in all) so that one's MFMA phase covers the other's memory phase.
// Per-workgroup accumulators go to a slab in fragment order [tile][r][lane] (reduced by k_rtz2_reduce).
// ------------------------------------------------------------------------------------------
#ifdef RTZ_PROF   // cycle stamps per loop segment of k_rtz2 (timing experiments only)
__device__ unsigned long long g_rtz_prof[8];
#define RTZ_STAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pf_[k] += now_ - pt_; pt_ = now_; }
#else
#define RTZ_STAMP(k)
#endif
template <int MT, int NTD, bool ONES>
__global__ __launch_bounds__(256, 2) void k_rtz2(RtzArgs a) {
    constexpr int SPLIT = 4 / NTD;                  // waves sharing one PC column block
    constexpr int MTW = (MT + SPLIT - 1) / SPLIT;   // cluster tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int LDR = a.ldr, LDZ = a.ldz;
    const int tile_floats = 16 * (LDR + LDZ);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int nt = wv % NTD, ms = SPLIT == 1 ? 0 : wv / NTD;   // SPLIT == 1: a compile-time 0 keeps `mt < MT` out of the loop
    const int wg = blockIdx.x;

    int t0, t1;
    if (a.task_tile0) {
        if (wg >= a.ntasks) return;
        t0 = a.task_tile0[wg];
        t1 = a.task_tile1[wg];
    } else {
        const int n_tiles = a.blk_start ? a.blk_start[a.nblk] : a.n_tiles;
        const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
        t0 = min(wg * per, n_tiles);
        t1 = min(t0 + per, n_tiles);
    }

    // this thread's (up to 3) 16-byte pieces of a tile: fixed (array, row, column), only the cell changes
    const int kp4 = a.Kp >> 2, dp4 = a.dp >> 2;
    const int nR = 16 * kp4, nZ = 16 * dp4;
    // Pieces a thread does not have (the last slots of a narrow tile) and padding cells are handled without
    // branches: they load a valid address, are masked to zero and land in a padding column of the R tile
    // that nobody reads, so the loop body stays one straight block the scheduler can interleave.
    int it_row[3], it_dst[3], it_stride[3];
    const float* it_base[3];
    bool it_ok[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int i = tid + 256 * s;
        if (i < nR) {
            it_row[s] = i / kp4;
            const int c = 4 * (i - it_row[s] * kp4);
            it_base[s] = a.R + c; it_stride[s] = a.Kp; it_ok[s] = true;
            it_dst[s] = it_row[s] * LDR + c;
        } else if (i < nR + nZ) {
            const int j = i - nR;
            it_row[s] = j / dp4;
            const int c = 4 * (j - it_row[s] * dp4);
            it_base[s] = a.Z + c; it_stride[s] = a.dp; it_ok[s] = true;
            it_dst[s] = 16 * LDR + it_row[s] * LDZ + c;
        } else {
            it_row[s] = 0;
            it_base[s] = a.R; it_stride[s] = 0; it_ok[s] = false;
            it_dst[s] = a.K16;                       // row 0, first padding column (LDR >= K16 + 16)
        }
    }
    for (int i = tid; i < 2 * tile_floats; i += 256) lds[i] = 0.f;   // the padding columns stay zero
    if (ONES) {
        // ... except the last one, which holds 1 for every cell: the MFMAs then deliver the column sums of R
        // (the removal sums) in PC column 16*NTD-1 for free, instead of 28 additions per tile and wave
        static_assert(!ONES || NTD == 4, "the ones column needs a wave that owns a whole column block");
        __syncthreads();
        if (tid < 32) lds[(size_t)(tid >> 4) * tile_floats + 16 * LDR + (tid & 15) * LDZ + 16 * NTD - 1] = 1.f;
    }
    // per-tile bookkeeping (group, block) comes from LDS copies: a dependent global load per tile
    // would cost a full memory round trip on the critical path
    int* grp_l = reinterpret_cast<int*>(lds + 2 * tile_floats);      // window of 256 tiles: key = block * G + group
    int* bs_l = grp_l + 256;                                          // nblk + 1 tile offsets (room for 256: HMX_MAX_BLOCKS = 250)
    double* sd = reinterpret_cast<double*>(bs_l + 256);                // ONES: K16 running column sums (fp64), owned by the last wave
    if (ONES) for (int i = tid; i < a.K16; i += 256) sd[i] = 0.0;
    const int task_g = a.task_tile0 ? a.task_grp[wg] : -1;
    if (a.blk_start)
        for (int i = tid; i <= a.nblk; i += 256) bs_l[i] = a.blk_start[i];

    f32x4 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // column sums of R (the removal sums): four values per tile in fp32, tiles in fp64 -- a long fp32
    // accumulation would drop the many tiny entries of R and bias O by ~5e-8 per round
    double csum[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) csum[i] = 0.0;
    int cur_key = -1;                              // block * G + group of the running column sums
    const bool sums = a.S_out != nullptr && (ONES ? nt == NTD - 1 : nt == 0);
    auto flush = [&](bool final_flush) {
        if (ONES) {
            // the sums sit in the accumulators of the last column block, PC column 16*NTD-1 (lanes c16 == 15):
            // short fp32 runs (16 tiles) are moved into fp64 LDS sums -- a long fp32 accumulation drops the
            // many tiny entries of R and biases O -- and the LDS sums go out when the (block, group) changes
            if (nt != NTD - 1) return;
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = acc[i][r];
                    if (c16 == 15) {
                        if (sv != 0.f) atomicAdd(&sd[16 * i + 4 * q + r], (double)sv);
                        acc[i][r] = 0.f;
                    }
                }
            if (!final_flush) return;
            for (int k = lane; k < a.K16; k += 64) {        // this wave's own LDS operations: in order, no barrier
                const double v = sd[k];
                if (v != 0.0) {
                    if (sums && cur_key >= 0 && k < a.K) atomicAdd(&a.S_out[(size_t)cur_key * a.K16 + k], v);
                    sd[k] = 0.0;
                }
            }
            return;
        }
        if (cur_key < 0 || !sums) return;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mt = ms + i * SPLIT;
            double sv = csum[i];
            sv += __shfl_xor(sv, 16, 64);
            sv += __shfl_xor(sv, 32, 64);
            const int k = 16 * mt + c16;
            if (q == 0 && mt < MT && k < a.K && sv != 0.0) atomicAdd(&a.S_out[(size_t)cur_key * a.K16 + k], sv);
            csum[i] = 0.0;
        }
    };
    // the loaded id is not touched here either: `live` says whether it counts, fetch() combines the two
    auto ids_of = [&](int t, int (&id)[3], bool& live) {
        const int tc = min(t, t1 - 1);                 // always a valid tile; tiles past the end are masked
#pragma unroll
        for (int s = 0; s < 3; ++s) id[s] = a.cells[(size_t)tc * 16 + it_row[s]];
        live = t < t1;
    };
    // the loaded value is not touched before it is written to LDS (a mask applied at load time would
    // wait for the load at once); the ids travel with it and mask it there
    auto fetch = [&](const int (&id)[3], bool live, f32x4 (&v)[3], int (&idv)[3]) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int c = (live && it_ok[s]) ? id[s] : -1;
            v[s] = ld4(it_base[s] + (size_t)max(c, 0) * it_stride[s]);
            idv[s] = c;
        }
    };
    auto stash = [&](int buf, const f32x4 (&v)[3], const int (&idv)[3]) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
            st4(lds + (size_t)buf * tile_floats + it_dst[s], idv[s] >= 0 ? v[s] : (f32x4){0.f, 0.f, 0.f, 0.f});
    };

    // software pipeline: tile t multiplies from LDS while tile t+1 is written to the other LDS buffer,
    // the loads of tiles t+2 .. t+RTZ_DEPTH travel in registers and the cell ids of tile
    // t+RTZ_DEPTH+1 are on their way
    constexpr int RTZ_DEPTH = 2;
    int id_n[3], idv[RTZ_DEPTH][3], key_nxt = 0;
    bool live_n = false;
    f32x4 v[RTZ_DEPTH][3];
    __syncthreads();
    if (t0 < t1) {
        ids_of(t0, id_n, live_n);
        fetch(id_n, live_n, v[0], idv[0]);
        stash(0, v[0], idv[0]);
#pragma unroll
        for (int dpt = 0; dpt < RTZ_DEPTH; ++dpt) {
            ids_of(t0 + 1 + dpt, id_n, live_n);
            fetch(id_n, live_n, v[dpt], idv[dpt]);          // tiles t0+1 .. t0+RTZ_DEPTH travelling
        }
        ids_of(t0 + 1 + RTZ_DEPTH, id_n, live_n);
    }
#ifdef RTZ_PROF
    unsigned long long pf_[6] = {0, 0, 0, 0, 0, 0}, pt_ = __builtin_amdgcn_s_memtime();
#endif
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        RTZ_STAMP(0)
        __syncthreads();          // tile t is complete in lds[buf]; nobody reads lds[buf ^ 1] any more
        RTZ_STAMP(1)
        stash(buf ^ 1, v[0], idv[0]);     // tile t+1 (its loads were issued RTZ_DEPTH iterations ago)
#pragma unroll
        for (int dpt = 0; dpt + 1 < RTZ_DEPTH; ++dpt)
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) { v[dpt][s3] = v[dpt + 1][s3]; idv[dpt][s3] = idv[dpt + 1][s3]; }
        fetch(id_n, live_n, v[RTZ_DEPTH - 1], idv[RTZ_DEPTH - 1]);        // tile t+1+RTZ_DEPTH
        ids_of(t + 2 + RTZ_DEPTH, id_n, live_n);      // ids for the fetch of the next iteration
        RTZ_STAMP(2)
        int key;
        if (task_g >= 0) {
            key = task_g;                         // a task is one group, block 0
        } else {
            if (((t - t0) & 255) == 0) {          // refill the key window (workgroup-uniform)
                __syncthreads();
                if (t + tid < t1) {
                    int b = 0;
                    if (a.blk_start) while (t + tid >= bs_l[b + 1]) ++b;   // lists are block-major
                    grp_l[tid] = b * a.G + a.tile_grp[t + tid];
                }
                __syncthreads();
                key = grp_l[0];
            } else {
                key = key_nxt;
            }
            key_nxt = grp_l[(t - t0 + 1) & 255];   // next tile's key: its LDS latency runs under this tile's work
        }
        if (key != cur_key) {
            flush(true);
            cur_key = key;
        } else if (ONES && ((t - t0) & 15) == 15) {
            flush(false);
        }
        RTZ_STAMP(3)
        const float* Rt = lds + (size_t)buf * tile_floats;
        const float* Zt = Rt + 16 * LDR;
        float tsum[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) tsum[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float bv = Zt[(4 * ks + q) * LDZ + 16 * nt + c16];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int mt = ms + i * SPLIT;
                if (mt < MT) {
                    const float av = Rt[(4 * ks + q) * LDR + 16 * mt + c16];
                    if (!ONES) tsum[i] += av;
                    acc[i] = MFMA16(av, bv, acc[i]);
                }
            }
        }
        if (!ONES && sums) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) csum[i] += (double)tsum[i];
        }
        RTZ_STAMP(4)
    }
#ifdef RTZ_PROF
    if (lane == 0) { for (int k = 0; k < 5; ++k) atomicAdd(&g_rtz_prof[k], pf_[k]); atomicAdd(&g_rtz_prof[5], (unsigned long long)(t1 - t0)); }
#endif
    flush(true);
    float* slab = a.slab + (size_t)wg * (MT * NTD * 256);
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int mt = ms + i * SPLIT;
        if (mt < MT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[((mt * NTD + nt) * 4 + r) * 64 + lane] = acc[i][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_rtz_wide: k_rtz2's scheme for K up to 208 / d up to 208 (13 x 13 output tiles): eight waves,
// wave w owns the PC column blocks w and w + 8 and all cluster tiles (26 accumulators of 4 registers),
// tiles of 16 cells staged through double-buffered LDS with 16-byte row loads, one workgroup per CU.
// ------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(512, 2) void k_rtz_wide(RtzArgs a) {
    const int NTD = a.ntd;                          // PC column blocks (<= 16)
    constexpr int NB = 2;                           // PC column blocks per wave: wave w owns blocks w, w + 8 and ALL cluster tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int LDR = a.ldr, LDZ = a.ldz;
    const int tile_floats = 16 * (LDR + LDZ);
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the wave's column blocks are wave-uniform
    int c16 = lane & 15, q = lane >> 4;                         // (refreshed per tile, see the loop)
    const int wg = blockIdx.x;

    int t0, t1;
    if (a.task_tile0) {
        if (wg >= a.ntasks) return;
        t0 = a.task_tile0[wg];
        t1 = a.task_tile1[wg];
    } else {
        const int n_tiles = a.blk_start ? a.blk_start[a.nblk] : a.n_tiles;
        const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
        t0 = min(wg * per, n_tiles);
        t1 = min(t0 + per, n_tiles);
    }

    // this thread's (up to 4) 16-byte pieces of a tile: fixed (array, row, column), only the cell changes
    const int kp4 = a.Kp >> 2, dp4 = a.dp >> 2;
    const int nR = 16 * kp4, nZ = 16 * dp4;
    int it_row[4], it_src[4], it_dst[4];   // row in the tile; float offset inside the source row (-1: none; bit 30: Z); LDS float offset
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = tid + 512 * s;
        if (i < nR) {
            it_row[s] = i / kp4;
            it_src[s] = 4 * (i - it_row[s] * kp4);
            it_dst[s] = it_row[s] * LDR + it_src[s];
        } else if (i < nR + nZ) {
            const int j = i - nR;
            it_row[s] = j / dp4;
            const int c = 4 * (j - it_row[s] * dp4);
            it_src[s] = c | (1 << 30);
            it_dst[s] = 16 * LDR + it_row[s] * LDZ + c;
        } else {
            it_row[s] = 0;
            it_src[s] = -1;
            it_dst[s] = 0;
        }
    }
    for (int i = tid; i < 2 * tile_floats; i += 512) lds[i] = 0.f;   // the padding columns stay zero
    // per-tile bookkeeping (group, block) comes from LDS copies: a dependent global load per tile
    // would cost a full memory round trip on the critical path
    int* grp_l = reinterpret_cast<int*>(lds + 2 * tile_floats);      // 256 tile groups
    int* bs_l = grp_l + 256;                                          // nblk + 1 tile offsets (room for 256: HMX_MAX_BLOCKS = 250)
    const int task_g = a.task_tile0 ? a.task_grp[wg] : -1;
    if (a.blk_start)
        for (int i = tid; i <= a.nblk; i += 512) bs_l[i] = a.blk_start[i];

    f32x4 acc[NB][MT];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int MTW = MT;
    // column sums of R (the removal sums): four values per tile in fp32, tiles in fp64 -- a long fp32
    // accumulation would drop the many tiny entries of R and bias O by ~5e-8 per round
    double csum[MT];
#pragma unroll
    for (int i = 0; i < MTW; ++i) csum[i] = 0.0;
    int cur_g = -1, cur_b = 0;
    const bool sums = wv == 0 && a.S_out != nullptr;
    auto flush = [&]() {
        if (cur_g < 0 || !sums) return;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mt = i;
            double sv = csum[i];
            sv += __shfl_xor(sv, 16, 64);
            sv += __shfl_xor(sv, 32, 64);
            const int k = 16 * mt + c16;
            if (q == 0 && mt < MT && k < a.K && sv != 0.0) atomicAdd(&a.S_out[((size_t)cur_b * a.G + cur_g) * a.K16 + k], sv);
            csum[i] = 0.0;
        }
    };
    auto ids_of = [&](int t, int (&id)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) id[s] = (t < t1 && it_src[s] != -1) ? a.cells[(size_t)t * 16 + it_row[s]] : -1;
    };
    auto fetch = [&](const int (&id)[4], f32x4 (&v)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (id[s] >= 0) {
                const bool isz = (it_src[s] >> 30) & 1;
                const int off = it_src[s] & 0xFFFFFF;
                v[s] = isz ? ld4(a.Z + (size_t)id[s] * a.dp + off) : ld4(a.R + (size_t)id[s] * a.Kp + off);
            }
        }
    };
    auto stash = [&](int buf, const f32x4 (&v)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (it_src[s] != -1) st4(lds + (size_t)buf * tile_floats + it_dst[s], v[s]);
    };

    // software pipeline: tile t multiplies from LDS while tile t+1 is written to the other LDS buffer,
    // the loads of tiles t+2 .. t+RTZ_DEPTH travel in registers and the cell ids of tile
    // t+RTZ_DEPTH+1 are on their way
    constexpr int RTZ_DEPTH = 2;
    int id_n[4];
    f32x4 v[RTZ_DEPTH][4];
    __syncthreads();
    if (t0 < t1) {
        ids_of(t0, id_n);
        fetch(id_n, v[0]);
        stash(0, v[0]);
#pragma unroll
        for (int dpt = 0; dpt < RTZ_DEPTH; ++dpt) {
            ids_of(t0 + 1 + dpt, id_n);
            fetch(id_n, v[dpt]);          // tiles t0+1 .. t0+RTZ_DEPTH travelling
        }
        ids_of(t0 + 1 + RTZ_DEPTH, id_n);
    }
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        // the lane's coordinates pass through an empty asm once per tile: the fragment addresses derived from them are
        // recomputed instead of being kept across the loop (MT = 13: 8 of them were spilled, every reload an
        // s_waitcnt vmcnt(0) in front of the loads in flight; scripts/kernel_audit.py)
        asm volatile("" : "+v"(lane), "+v"(c16), "+v"(q));
        __syncthreads();          // tile t is complete in lds[buf]; nobody reads lds[buf ^ 1] any more
        stash(buf ^ 1, v[0]);     // tile t+1 (its loads were issued RTZ_DEPTH iterations ago)
#pragma unroll
        for (int dpt = 0; dpt + 1 < RTZ_DEPTH; ++dpt)
#pragma unroll
            for (int s3 = 0; s3 < 4; ++s3) v[dpt][s3] = v[dpt + 1][s3];
        fetch(id_n, v[RTZ_DEPTH - 1]);        // tile t+1+RTZ_DEPTH
        ids_of(t + 2 + RTZ_DEPTH, id_n);      // ids for the fetch of the next iteration
        int g = task_g;
        if (task_g < 0) {
            if (((t - t0) & 255) == 0) {          // refill the group window (workgroup-uniform)
                __syncthreads();
                if (tid < 256 && t + tid < t1) grp_l[tid] = a.tile_grp[t + tid];
                __syncthreads();
            }
            g = grp_l[(t - t0) & 255];
        }
        int b = cur_b;
        if (a.blk_start) {
            while (t >= bs_l[b + 1]) ++b;         // lists are block-major, so b only grows
        }
        if (g != cur_g || b != cur_b) {
            flush();
            cur_g = g;
            cur_b = b;
        }
        const float* Rt = lds + (size_t)buf * tile_floats;
        const float* Zt = Rt + 16 * LDR;
        float tsum[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) tsum[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float bv[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nt = wv + 8 * j;
                bv[j] = (nt < NTD) ? Zt[(4 * ks + q) * LDZ + 16 * nt + c16] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = Rt[(4 * ks + q) * LDR + 16 * i + c16];
                tsum[i] += av;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (wv + 8 * j < NTD) acc[j][i] = MFMA16(av, bv[j], acc[j][i]);   // wave-uniform
            }
        }
        if (sums) {
#pragma unroll
            for (int i = 0; i < MT; ++i) csum[i] += (double)tsum[i];
        }
    }
    flush();
    float* slab = a.slab + (size_t)wg * (MT * NTD * 256);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int nt = wv + 8 * j;
        if (nt < NTD) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[((i * NTD + nt) * 4 + r) * 64 + lane] = acc[j][i][r];
        }
    }
}

// Sum the per-workgroup slabs of k_rtz2 in fp64 (same contract as k_rtz_reduce).
__global__ __launch_bounds__(256) void k_rtz2_reduce(const float* __restrict__ slab, int nslabs, int MT, int NTD, int K16,
                                                     int ld, double* __restrict__ out, const int* __restrict__ task_grp,
                                                     int seg_len) {
    const int per = MT * NTD * 256;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const int lane = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int mt = tile / NTD, nt = tile % NTD;
    const int k = 16 * mt + 4 * (lane >> 4) + r;
    const int j = 16 * nt + (lane & 15);
    if (k >= K16 || j >= ld) return;
    const int w0 = blockIdx.y * seg_len, w1 = min(w0 + seg_len, nslabs);
    int g = -1;
    double acc = 0.0;
    for (int w = w0; w < w1; ++w) {
        const int gw = task_grp ? task_grp[w] : 0;
        if (gw != g) {
            if (g >= 0 && acc != 0.0) atomicAdd(&out[((size_t)g * K16 + k) * ld + j], acc);
            g = gw;
            acc = 0.0;
        }
        acc += (double)slab[(size_t)w * per + e];
    }
    if (g >= 0 && acc != 0.0) atomicAdd(&out[((size_t)g * K16 + k) * ld + j], acc);
}

// Sum the per-wave slabs of k_rtz in fp64.  blockIdx.x covers the slab elements, blockIdx.y a
// segment of the slabs (waves / tasks); partial sums meet in fp64 atomics, whose order does not
// change an fp32-rounded result.  `out` must be zeroed by the caller.
//   task_grp == null: out[k*ld + j]              += sum over the segment's waves  (centroid numerator)
//   task_grp != null: out[(g*K16 + k)*ld + j]    += sum over the segment's tasks of group g (ridge)
template <int MTW, int NTW>
__global__ __launch_bounds__(256) void k_rtz_reduce(const float* __restrict__ slab, int nwaves, int nsub, int ntd,
                                                    int K16, int ld, double* __restrict__ out,
                                                    const int* __restrict__ task_grp, int seg_len) {
    const int per_sub = MTW * NTW * 256;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nsub * per_sub) return;
    const int sub = idx / per_sub, e = idx % per_sub;
    const int lane = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int mt = tile / NTW, nt = tile % NTW;
    const int nsub_n = (ntd + NTW - 1) / NTW;
    const int k = 16 * ((sub / nsub_n) * MTW + mt) + 4 * (lane >> 4) + r;
    const int j = 16 * ((sub % nsub_n) * NTW + nt) + (lane & 15);
    if (k >= K16 || j >= ld) return;
    const int w0 = blockIdx.y * seg_len, w1 = min(w0 + seg_len, nwaves);
    int g = -1;
    double acc = 0.0;
    for (int w = w0; w < w1; ++w) {
        const int gw = task_grp ? task_grp[w] : 0;
        if (gw != g) {
            if (g >= 0 && acc != 0.0) atomicAdd(&out[((size_t)g * K16 + k) * ld + j], acc);
            g = gw;
            acc = 0.0;
        }
        acc += (double)slab[((size_t)w * nsub + sub) * per_sub + e];
    }
    if (g >= 0 && acc != 0.0) atomicAdd(&out[((size_t)g * K16 + k) * ld + j], acc);
}

// ------------------------------------------------------------------------------------------
// Per-block diversity table (harmony.py:491-499), one workgroup per cluster k.
//   O_use[g] = O_prev[g] (+ S_new of the previous block) - S_old of this block
//   E[k,b]   = T_use * Pr_b[b]              (harmony.py:491, kept as mass T instead of K x B)
//   ratio    = clamp(E / clamp(O + E, 1e-8), 1e-8, 1) ** theta_b
//   rp[g][k] = sum_v ratio[col(g, v)]        (= (ratio_powered @ Phi) for the cells of group g)
// LDS: B doubles (column sums), B floats (powered ratio), reduction scratch.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_block_table(TableArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ocol = reinterpret_cast<double*>(smem);           // B
    double* red = ocol + a.B;                                  // 128
    float* rpc = reinterpret_cast<float*>(red + 128);          // B
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    for (int b = tid; b < a.B; b += blockDim.x) ocol[b] = 0.0;
    __syncthreads();
    double tpart = 0.0;
    for (int g = tid; g < a.G; g += blockDim.x) {
        const size_t i = (size_t)g * a.K16 + k;
        double o = a.O_prev[i];
        if (a.S_add) o += a.S_add[i];
        if (a.S_sub) o -= a.S_sub[i];
        if (a.O_out) a.O_out[i] = o;
        tpart += o;
        if (a.V == 1) {
            ocol[a.group_cols[g]] = o;
        } else {
            for (int v = 0; v < a.V; ++v) atomicAdd(&ocol[a.group_cols[g * a.V + v]], o);
        }
    }
    red[tid] = tpart;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double T = red[0];
    if (tid == 0 && a.T_out) a.T_out[k] = T;
    if (a.rp) {
        for (int b = tid; b < a.B; b += blockDim.x) {
            const float O = (float)ocol[b];
            const float E = (float)T * a.Pr_b[b];
            const float oe = fmaxf(O + E, 1e-8f);                       // :495-496
            const float ratio = fminf(fmaxf(E / oe, 1e-8f), 1.0f);      // :497-498
            rpc[b] = powf(ratio, a.theta[b]);                           // :499
        }
        __syncthreads();
        for (int g = tid; g < a.G; g += blockDim.x) {
            float s = 0.f;
            for (int v = 0; v < a.V; ++v) s += rpc[a.group_cols[g * a.V + v]];
            a.rp[(size_t)g * a.K16 + k] = s;
            a.lrp[(size_t)g * a.K16 + k] = logf(s);
        }
    }
    if (a.obj_cross) {
        // cross-entropy term of the objective (harmony.py:405-411) collapsed to K x B:
        //   sum_b sigma_k theta_b log((O_c + E_c)/E_c) * (R Phi^T)[k,b]
        double part = 0.0;
        const float sig = a.sigma[k];
        for (int b = tid; b < a.B; b += blockDim.x) {
            const float O = (float)ocol[b];
            const float Oc = fmaxf(O, 1e-8f);                           // :407
            const float Ec = fmaxf((float)T * a.Pr_b[b], 1e-8f);        // :408
            const float tl = a.theta[b] * logf((Oc + Ec) / Ec);         // :409-410
            part += (double)(sig * O * tl);
        }
        __syncthreads();
        red[tid] = part;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) atomicAdd(a.obj_cross, red[0]);
    }
}

// ------------------------------------------------------------------------------------------
// Group column sums of R over a static tile list (exact O after hmx_set(R)).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_sums(const float* __restrict__ R, int Kp, int K, int K16,
                                                    const int* __restrict__ cells, const int* __restrict__ tile_grp,
                                                    int n_tiles, double* __restrict__ Ogrp) {
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const int g = tile_grp[t];
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < 16; ++i) {
            const int cell = cells[(size_t)t * 16 + i];
            if (cell >= 0) s += R[(size_t)cell * Kp + k];
        }
        if (s != 0.f) atomicAdd(&Ogrp[(size_t)g * K16 + k], (double)s);
    }
}

// ------------------------------------------------------------------------------------------
// Ridge solve, one workgroup per cluster k (harmony.py:541-565), single batch variable.
// The system (Phi_moe diag(R_k) Phi_moe^T + diag(lambda)) W = Phi_moe diag(R_k) Z^T is an
// arrowhead matrix; eliminating the batch rows gives, with O_b = sum_i R_ki [b(i)=b],
// S_b = sum_{i in b} R_ki z_i  and c_b = lambda_b / (O_b + lambda_b):
//     w_0 = (sum_b c_b S_b) / (lambda_0 + sum_b c_b O_b)
//     w_b = (S_b - O_b w_0) / (O_b + lambda_b)
// evaluated in fp64 (cancellation-free form).  Row 0 is dropped (harmony.py:565).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ridge_solve_v1(RidgeSolveArgs a) {
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    // lambda_b: fixed, or alpha * E[k,b] = alpha * T[k] * Pr_b[b] (harmony.py:587-591)
    for (int j = tid; j < a.ldw; j += blockDim.x) {
        double num = 0.0, den = a.lambda_est ? 0.0 : (double)a.lamb[0];
        if (j < a.d) {
            for (int b = 0; b < a.B; ++b) {
                const double lam = a.lambda_est ? (double)(a.alpha * ((float)a.T[k] * a.Pr_b[b])) : (double)a.lamb[b + 1];
                const double O = a.Ox[(size_t)b * a.K16 + k];
                const double c = lam / (O + lam);
                num += c * a.S[((size_t)b * a.K16 + k) * a.lds + j];
                den += c * O;
            }
        }
        const double w0 = (j < a.d && k < a.K) ? num / den : 0.0;
        for (int b = 0; b < a.B; ++b) {
            float w = 0.f;
            if (j < a.d && k < a.K) {
                const double lam = a.lambda_est ? (double)(a.alpha * ((float)a.T[k] * a.Pr_b[b])) : (double)a.lamb[b + 1];
                const double O = a.Ox[(size_t)b * a.K16 + k];
                w = (float)((a.S[((size_t)b * a.K16 + k) * a.lds + j] - O * w0) / (O + lam));
            }
            a.W[((size_t)b * a.K16 + k) * a.ldw + j] = w;
        }
    }
}

// General case (several batch variables): dense (B+1) x (B+1) system per cluster, Gauss-Jordan
// with partial pivoting in fp64 on an augmented matrix [cov | rhs] held in global scratch.
// One workgroup per cluster.  Group tables -> column tables:
//   cov[0][0] = sum_g Ox[g] + lambda_0, cov[0][b+1] = Ocol[b], cov[b+1][c+1] = sum_{g has b,c} Ox[g]
//   rhs[0] = sum_g S_g, rhs[b+1] = sum_{g has b} S_g;   W_eff[g] = sum_v w[col(g,v)+1]
__global__ __launch_bounds__(256) void k_ridge_solve_general(RidgeSolveArgs a) {
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.B + 1;
    const int wcols = n + a.d;
    double* M = a.scratch + (size_t)k * n * wcols;
    __shared__ int piv_row;
    __shared__ double piv_val;
    for (int i = tid; i < n * wcols; i += blockDim.x) M[i] = 0.0;
    __threadfence_block();
    __syncthreads();
    if (k >= a.K) {
        for (int g = 0; g < a.G; ++g)
            for (int j = tid; j < a.ldw; j += blockDim.x) a.W[((size_t)g * a.K16 + k) * a.ldw + j] = 0.f;
        return;
    }
    // assemble (serial over groups, parallel over columns: no atomics)
    for (int g = 0; g < a.G; ++g) {
        const double O = a.Ox[(size_t)g * a.K16 + k];
        const int* cols = a.group_cols + g * a.V;
        if (tid == 0) {
            M[0] += O;
            for (int v = 0; v < a.V; ++v) {
                const int b = cols[v] + 1;
                M[b] += O;
                M[(size_t)b * wcols] += O;
                for (int u = 0; u < a.V; ++u) M[(size_t)b * wcols + cols[u] + 1] += O;
            }
        }
        for (int j = tid; j < a.d; j += blockDim.x) {
            const double s = a.S[((size_t)g * a.K16 + k) * a.lds + j];
            M[n + j] += s;
            for (int v = 0; v < a.V; ++v) M[(size_t)(cols[v] + 1) * wcols + n + j] += s;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int b = tid; b < n; b += blockDim.x) {
        double lam;
        if (a.lambda_est) lam = (b == 0) ? 0.0 : (double)(a.alpha * ((float)a.T[k] * a.Pr_b[b - 1]));
        else lam = (double)a.lamb[b];
        M[(size_t)b * wcols + b] += lam;
    }
    __threadfence_block();
    __syncthreads();
    // Gauss-Jordan
    for (int c = 0; c < n; ++c) {
        if (tid == 0) {
            int best = c;
            double bv = fabs(M[(size_t)c * wcols + c]);
            for (int r = c + 1; r < n; ++r) {
                const double v = fabs(M[(size_t)r * wcols + c]);
                if (v > bv) { bv = v; best = r; }
            }
            piv_row = best;
        }
        __syncthreads();
        const int pr = piv_row;
        if (pr != c) {
            for (int j = tid; j < wcols; j += blockDim.x) {
                const double t = M[(size_t)c * wcols + j];
                M[(size_t)c * wcols + j] = M[(size_t)pr * wcols + j];
                M[(size_t)pr * wcols + j] = t;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) piv_val = M[(size_t)c * wcols + c];
        __syncthreads();
        const double inv = 1.0 / piv_val;
        for (int j = tid; j < wcols; j += blockDim.x) M[(size_t)c * wcols + j] *= inv;
        __threadfence_block();
        __syncthreads();
        // eliminate column c from every other row; thread owns (row, column-strip)
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = M[(size_t)r * wcols + c];
            __syncthreads();
            if (f != 0.0)
                for (int j = tid; j < wcols; j += blockDim.x) M[(size_t)r * wcols + j] -= f * M[(size_t)c * wcols + j];
            __threadfence_block();
            __syncthreads();
        }
    }
    for (int g = 0; g < a.G; ++g) {
        const int* cols = a.group_cols + g * a.V;
        for (int j = tid; j < a.ldw; j += blockDim.x) {
            double w = 0.0;
            if (j < a.d)
                for (int v = 0; v < a.V; ++v) w += M[(size_t)(cols[v] + 1) * wcols + n + j];
            a.W[((size_t)g * a.K16 + k) * a.ldw + j] = (float)w;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Ridge apply (harmony.py:566, 569):  Z_corr_i = Z_orig_i - sum_k R_ik W[g(i)][k][:],
// then Z_cos_i = Z_corr_i / ||Z_corr_i||.
// C[pc][cell] = W_g^T (d x K) . R^T (K x cells): A = W_g (4-byte loads, L1/L2), B = R rows
// gathered 16 B per lane.  A lane ends with PCs {16mt + 4q + r} of cell c16.
// ------------------------------------------------------------------------------------------
template <int MTD, int NT>
__global__ __launch_bounds__(256) void k_ridge_apply(ApplyArgs a) {
    const int lane = threadIdx.x & 63;
    const int c16 = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    int per = (a.n_tiles + nwaves - 1) / nwaves;
    per = ((per + NT - 1) / NT) * NT;
    const int t0 = wave * per, t1 = min(t0 + per, a.n_tiles);
    const int kb_n = (a.Kp + 15) >> 4;
    const int mtn = a.mtd;

    for (int t = t0; t < t1; t += NT) {
        // tiles of one step may belong to different groups: process same-group runs together
        int cell[NT], grp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int tt = t + nt;
            cell[nt] = (tt < t1) ? a.cells[(size_t)tt * 16 + c16] : -1;
            grp[nt] = (tt < t1) ? a.tile_grp[tt] : -1;
        }
        f32x4 acc[MTD][NT];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < kb_n; ++kb) {
            const int col0 = 16 * kb + 4 * q;
            f32x4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = (cell[nt] >= 0 && col0 < a.Kp) ? ld4(a.R + (size_t)cell[nt] * a.Kp + col0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (grp[nt] < 0) continue;
                const float* Wg = a.W + (size_t)grp[nt] * a.K16 * a.ldw;
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) {
                    if (mt < mtn) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float wa = Wg[(size_t)(col0 + i) * a.ldw + 16 * mt + c16];
                            acc[mt][nt] = MFMA16(wa, b[nt][i], acc[mt][nt]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (grp[nt] < 0) continue;
            const bool live = cell[nt] >= 0;
            const size_t row = (size_t)(live ? cell[nt] : 0) * a.dp;
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                if (mt < mtn) {
                    const int col = 16 * mt + 4 * q;
                    if (live && col < a.dp) {
                        const f32x4 zo = ld4(a.Zorig + row + col);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[mt][nt][r] = zo[r] - acc[mt][nt][r];
                            ss += acc[mt][nt][r] * acc[mt][nt][r];
                        }
                        st4(a.Zcorr + row + col, acc[mt][nt]);
                    }
                }
            }
            ss = wave_sum_q(ss);
            const float nrm = sqrtf(ss);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                if (mt < mtn) {
                    const int col = 16 * mt + 4 * q;
                    if (live && col < a.dp) {
                        f32x4 zc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) zc[r] = acc[mt][nt][r] / nrm;
                        st4(a.Zcos + row + col, zc);
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// k_ridge_apply2: the same correction for K <= 112, d <= 64.  One workgroup per task (a run of
// tiles of ONE group, the ridge-statistics tasks): the group's W (K16 x d, <= 36 KB) is staged in
// LDS once, with a row stride of 16 (mod 32) floats so that the A-fragment reads are conflict-free;
// every wave then streams pairs of tiles: 16-byte R row pieces straight into B-fragment layout,
// MTD x 2 accumulators, Z_orig row pieces for the epilogue, Z_corr / Z_cos rows out as 16-byte stores.
// ------------------------------------------------------------------------------------------
template <int MTD, int KB>
__global__ __launch_bounds__(256, 3) void k_ridge_apply2(ApplyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Wl = reinterpret_cast<float*>(smem);                 // K16 x LDW
    const int LDW = a.ldw_lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int task = blockIdx.x;
    if (task >= a.ntasks) return;
    const int t0 = a.task_tile0[task], t1 = a.task_tile1[task], g = a.task_grp[task];
    {
        const float* Wg = a.W + (size_t)g * a.K16 * a.ldw;
        const int w4 = (16 * MTD) >> 2;
        for (int i = tid; i < a.K16 * w4; i += 256) {
            const int row = i / w4, c4 = i - row * w4;
            st4(Wl + (size_t)row * LDW + 4 * c4, ld4(Wg + (size_t)row * a.ldw + 4 * c4));
        }
    }
    __syncthreads();
    for (int t = t0 + 2 * wv; t < t1; t += 8) {
        int cell[2];
        f32x4 b[2][KB];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            cell[n] = (t + n < t1) ? a.cells[(size_t)(t + n) * 16 + c16] : -1;
            const float* rr = a.R + (size_t)(cell[n] >= 0 ? cell[n] : 0) * a.Kp;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int col0 = 16 * kb + 4 * q;
                b[n][kb] = (cell[n] >= 0 && col0 < a.Kp) ? ld4(rr + col0) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 zo[2][MTD];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int col = 16 * mt + 4 * q;
                zo[n][mt] = (cell[n] >= 0 && col < a.dp) ? ld4(a.Zorig + (size_t)cell[n] * a.dp + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        f32x4 acc[MTD][2];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) acc[mt][0] = acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* wr = Wl + (size_t)(16 * kb + 4 * q + i) * LDW + c16;
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) {
                    const float wa = wr[16 * mt];
                    acc[mt][0] = MFMA16(wa, b[0][kb][i], acc[mt][0]);
                    acc[mt][1] = MFMA16(wa, b[1][kb][i], acc[mt][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one column block's W fragments live at a time
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bool live = cell[n] >= 0;
            const size_t row = (size_t)(live ? cell[n] : 0) * a.dp;
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                acc[mt][n] = zo[n][mt] - acc[mt][n];                               // :566
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += acc[mt][n][r] * acc[mt][n][r];
            }
            ss = wave_sum_q(ss);
            const float nrm = sqrtf(ss);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int col = 16 * mt + 4 * q;
                if (live && col < a.dp) {
                    st4(a.Zcorr + row + col, acc[mt][n]);
                    f32x4 zc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) zc[r] = acc[mt][n][r] / nrm;       // :569
                    st4(a.Zcos + row + col, zc);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Lloyd iterations of the initial k-means (harmony.py:369-373 runs sklearn KMeans on the host:
// 18 s at 1M cells) on the device, for jobs too large for that.  Euclidean k-means on the
// unit-length rows of Z_cos, as sklearn does: label = argmin ||z - c||^2 = argmax (z.c - ||c||^2/2),
// centroid = mean of its members.
//   k_kmeans_step   : workgroups of 8 waves walk the static tile list; centres (K16 x d) and their
//                     half squared norms in LDS; one tile per wave per step: gather, f32 MFMA, per-cell
//                     argmax (28 scores per lane, then two xor-shuffles); members' rows are added to an
//                     LDS table of sums with ds_add_f32, flushed once per workgroup to a slab.
//   k_kmeans_update : one workgroup per cluster sums the slabs in fp64 -> new centre (empty clusters keep
//                     theirs), half squared norm.
// Seeding (k-means++) stays on the host, on a subsample (harmonypy_amd/harmony.py).
// ------------------------------------------------------------------------------------------
struct KmeansArgs {
    const float* Zcos;     // N x dp
    const float* C;        // K16 x ldy centres (rows >= K zero)
    const float* hn;       // K16 half squared norms (+inf for pads)
    const int* cells;      // static list
    int n_tiles;
    float* slab;           // [wgs][K16 x ldc] sums, then [wgs][K16] counts
    int K, K16, dp, ldy, ldy_lds, ldc;
};

// Member sums without per-element atomics: the tile's rows go through a per-wave LDS tile, the hard
// assignment becomes a one-hot matrix (clusters x cells) built from the winners, and
// sums += onehot . Z runs on the MFMA pipe into 7 x 4 accumulator tiles per wave (K16 x 64 floats);
// the accumulators meet in the workgroup's LDS table once, at the end.  (52 ds_add_f32 per cell made
// the first version of this kernel LDS-atomic-bound: 340 us per iteration at 1M cells.)
#define KM_LDZ 68             /* row stride of the per-wave Z tile: 64 dims + 4 (conflict-free column reads) */
template <int MT>
__global__ __launch_bounds__(512, 1) void k_kmeans_step(KmeansArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K16 = 16 * MT;
    const int LDY = a.ldy_lds, LDC = a.ldc;
    float* Ys = reinterpret_cast<float*>(smem);            // K16 x LDY
    float* hn = Ys + (size_t)K16 * LDY;                    // K16
    float* Cs = hn + K16;                                  // K16 x LDC sums of member rows
    float* cnt = Cs + (size_t)K16 * LDC;                   // K16 member counts
    float* Zt_all = cnt + K16;                             // 8 waves x 16 x KM_LDZ: the waves' Z tiles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    float* Zt = Zt_all + (size_t)wv * 16 * KM_LDZ;
    const int c4n = a.dp >> 2, kb_full = a.dp >> 4, tail = c4n - 4 * kb_full;
    const int ntd = (a.dp + 15) >> 4;                      // 16-dim column blocks (<= 4)
    for (int i = tid; i < K16 * c4n; i += 512) {
        const int row = i / c4n, c4 = i - row * c4n;
        st4(Ys + (size_t)row * LDY + 4 * c4, ld4(a.C + (size_t)row * a.ldy + 4 * c4));
    }
    for (int i = tid; i < K16; i += 512) { hn[i] = a.hn[i]; cnt[i] = 0.f; }
    for (int i = tid; i < K16 * LDC; i += 512) Cs[i] = 0.f;
    for (int i = tid; i < 8 * 16 * KM_LDZ; i += 512) Zt_all[i] = 0.f;   // the columns beyond dp stay zero
    __syncthreads();
    f32x4 sacc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) sacc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the next tile's cell ids are requested at the top of an iteration and its rows once the ids are there
    // (after the distance product), so both round trips run under the MFMAs of the current tile
    auto load_rows = [&](int cell, f32x4 (&zr4)[4], float (&zt)[3]) {
        const float* zr = a.Zcos + (size_t)(cell >= 0 ? cell : 0) * a.dp;
#pragma unroll
        for (int j = 0; j < 4; ++j) zr4[j] = (j < kb_full) ? ld4(zr + 16 * j + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) zt[s2] = (s2 < tail) ? zr[16 * kb_full + 4 * s2 + q] : 0.f;
    };
    const int t_first = blockIdx.x * 8 + wv, t_step = gridDim.x * 8;
    int cell_n = t_first < a.n_tiles ? a.cells[(size_t)t_first * 16 + c16] : -1;
    f32x4 zrow_n[4];
    float ztail_n[3];
    load_rows(cell_n, zrow_n, ztail_n);
    for (int t = t_first; t < a.n_tiles; t += t_step) {
        const int cell = cell_n;
        const bool live = cell >= 0;
        f32x4 zrow[4];
        float ztail[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) zrow[j] = zrow_n[j];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) ztail[s2] = ztail_n[s2];
        const int t_next = t + t_step;
        cell_n = t_next < a.n_tiles ? a.cells[(size_t)t_next * 16 + c16] : -1;
        // the rows also go to this wave's LDS tile (cell-major), from where the second product reads columns
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < kb_full) st4(Zt + c16 * KM_LDZ + 16 * j + 4 * q, zrow[j]);
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
            if (s2 < tail) Zt[c16 * KM_LDZ + 16 * kb_full + 4 * s2 + q] = ztail[s2];
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < kb_full) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 ya = ld4(Ys + (size_t)(16 * mt + c16) * LDY + 16 * j + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[mt] = MFMA16(ya[i], zrow[j][i], acc[mt]);
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            if (s2 < tail) {
                const int col = 16 * kb_full + 4 * s2 + q;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA16(Ys[(size_t)(16 * mt + c16) * LDY + col], ztail[s2], acc[mt]);
            }
        }
        float best = -INFINITY;
        int bk = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 h = ld4(hn + 16 * mt + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = acc[mt][r] - h[r];
                const int k = 16 * mt + 4 * q + r;
                if (sc > best) { best = sc; bk = k; }
            }
        }
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {   // the four q-lanes of a cell hold disjoint clusters
            const float ob = __shfl_xor(best, m, 64);
            const int ok = __shfl_xor(bk, m, 64);
            if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; }
        }
        if (!live) bk = -1;                    // padding cells belong to no cluster
        load_rows(cell_n, zrow_n, ztail_n);    // next tile's rows: their latency runs under the second product
        if (live && q == 0) atomicAdd(cnt + bk, 1.0f);
        // sums += onehot(16 cells -> K16 clusters)^T . Z tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int win = __shfl(bk, 4 * ks + q, 64);              // winner of cell 4ks+q (held by lane 4ks+q)
            float bz[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bz[nt] = (nt < ntd) ? Zt[(4 * ks + q) * KM_LDZ + 16 * nt + c16] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float oh = (win == 16 * mt + c16) ? 1.f : 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (nt < ntd) sacc[mt][nt] = MFMA16(oh, bz[nt], sacc[mt][nt]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the tile is rewritten by the next iteration
        __builtin_amdgcn_wave_barrier();
    }
    // the waves' accumulators meet in the workgroup's table: lane (c16, q) holds sums[16mt+4q+r][16nt+c16]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = 16 * nt + c16;
            if (nt < ntd && col < a.dp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = sacc[mt][nt][r];
                    if (v != 0.f) atomicAdd(Cs + (size_t)(16 * mt + 4 * q + r) * LDC + col, v);
                }
            }
        }
    __syncthreads();
    float* out = a.slab + (size_t)blockIdx.x * K16 * LDC;
    for (int i = tid; i < K16 * LDC; i += 512) out[i] = Cs[i];
    float* outc = a.slab + (size_t)gridDim.x * K16 * LDC + (size_t)blockIdx.x * K16;
    for (int i = tid; i < K16; i += 512) outc[i] = cnt[i];
}

// sums[k][0..d) and counts[k] over the workgroup slabs, in fp64 (one workgroup per cluster)
__global__ __launch_bounds__(256) void k_kmeans_sums(const float* __restrict__ slab, int wgs, int K16, int ldc, int d,
                                                     double* __restrict__ sums /* K16 x (d+1) */) {
    __shared__ double red[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    for (int j = 0; j <= d; ++j) {
        double acc = 0.0;
        for (int w = tid; w < wgs; w += 256)
            acc += (j < d) ? (double)slab[((size_t)w * K16 + k) * ldc + j] : (double)slab[(size_t)wgs * K16 * ldc + (size_t)w * K16 + k];
        red[tid] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) sums[(size_t)k * (d + 1) + j] = red[0];
        __syncthreads();
    }
}

// member sums and counts of the wide-shape Lloyd step from the R^T.Z statistics of the one-hot assignment
// (Sr: G x K16 x ld sums per batch group, Oxr: G x K16 member counts per group)
__global__ __launch_bounds__(256) void k_kmeans_sums_from_stats(const double* __restrict__ Sr, const double* __restrict__ Oxr, int G,
                                                                int K16, int ld, int d, double* __restrict__ sums /* K16 x (d+1) */) {
    const int k = blockIdx.x;
    for (int j = threadIdx.x; j <= d; j += 256) {
        double s = 0.0;
        for (int g = 0; g < G; ++g) s += (j < d) ? Sr[((size_t)g * K16 + k) * ld + j] : Oxr[(size_t)g * K16 + k];
        sums[(size_t)k * (d + 1) + j] = s;
    }
}

// new centres from the (job-wide) sums: mean of the members; empty clusters keep their centre
__global__ __launch_bounds__(64) void k_kmeans_update(const double* __restrict__ sums, float* __restrict__ C, float* __restrict__ hn,
                                                      int K, int d, int ldy) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const double n = (k < K) ? sums[(size_t)k * (d + 1) + d] : 0.0;
    float ss = 0.f;
    for (int j = lane; j < ldy; j += 64) {
        float c = C[(size_t)k * ldy + j];
        if (k < K && j < d && n > 0.0) c = (float)(sums[(size_t)k * (d + 1) + j] / n);
        if (k >= K || j >= d) c = 0.f;
        C[(size_t)k * ldy + j] = c;
        ss += c * c;
    }
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if (lane == 0) hn[k] = (k < K) ? 0.5f * ss : INFINITY;
}

// ------------------------------------------------------------------------------------------
// k_ridge_apply_wide: the correction for K up to 208 / d up to 208.  One workgroup (8 waves) per task
// (tiles of one group); the group's W (K16 x d, up to 170 KB) is walked in k-steps of 16 clusters: a
// step's 16 rows (16 x dp floats, 13 KB) are staged in LDS, double buffered, the next step's rows and
// R pieces travelling in registers; every wave multiplies its own tile (52 MFMAs per step), then
// subtracts from its Z_orig rows and renormalises (harmony.py:566, 569).
// ------------------------------------------------------------------------------------------
template <int MTD>
__global__ __launch_bounds__(64 * WIDE_WAVES, 2) void k_ridge_apply_wide(ApplyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Wsh = reinterpret_cast<float*>(smem);                 // 2 x 16 x LDW
    const int LDW = a.ldw_lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int task = blockIdx.x;
    if (task >= a.ntasks) return;
    const int t0 = a.task_tile0[task], t1 = a.task_tile1[task], g = a.task_grp[task];
    const float* Wg = a.W + (size_t)g * a.K16 * a.ldw;
    const int nkb = a.K16 >> 4;
    const int w4 = (16 * MTD) >> 2;                              // 16-byte pieces of a W row
    constexpr int WPT = (16 * 16 * MTD / 4 + 64 * WIDE_WAVES - 1) / (64 * WIDE_WAVES);
    int stage = 0;
    for (int base = t0; base < t1; base += WIDE_WAVES) {        // workgroup-uniform trip count
        const int t = base + wv;
        const bool has = t < t1;
        const int cell = has ? a.cells[(size_t)t * 16 + c16] : -1;
        const bool live = cell >= 0;
        const float* rr = a.R + (size_t)(live ? cell : 0) * a.Kp;
        f32x4 acc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 wnext[WPT], bnext;
        auto fetch_step = [&](int kb) {
#pragma unroll
            for (int p2 = 0; p2 < WPT; ++p2) {
                const int i = tid + 64 * WIDE_WAVES * p2;
                const int row = i / w4, c4 = i - row * w4;
                wnext[p2] = (i < 16 * w4) ? ld4(Wg + (size_t)(16 * kb + row) * a.ldw + 4 * c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int col0 = 16 * kb + 4 * q;
            bnext = (live && col0 < a.Kp) ? ld4(rr + col0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        fetch_step(0);
        for (int kb = 0; kb < nkb; ++kb) {
            const f32x4 b = bnext;
#pragma unroll
            for (int p2 = 0; p2 < WPT; ++p2) {
                const int i = tid + 64 * WIDE_WAVES * p2;
                const int row = i / w4, c4 = i - row * w4;
                if (i < 16 * w4) st4(Wsh + (size_t)stage * 16 * LDW + row * LDW + 4 * c4, wnext[p2]);
            }
            if (kb + 1 < nkb) fetch_step(kb + 1);
            __syncthreads();
            const float* Wst = Wsh + (size_t)stage * 16 * LDW;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* wr = Wst + (4 * q + i) * LDW + c16;
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) acc[mt] = MFMA16(wr[16 * mt], b[i], acc[mt]);
            }
            stage ^= 1;
        }
        if (has) {
            const size_t row = (size_t)(live ? cell : 0) * a.dp;
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int col = 16 * mt + 4 * q;
                const f32x4 zo = (live && col < a.dp) ? ld4(a.Zorig + row + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[mt] = zo - acc[mt];                                               // :566
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += acc[mt][r] * acc[mt][r];
            }
            ss = wave_sum_q(ss);
            const float nrm = sqrtf(ss);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int col = 16 * mt + 4 * q;
                if (live && col < a.dp) {
                    st4(a.Zcorr + row + col, acc[mt]);
                    f32x4 zc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) zc[r] = acc[mt][r] / nrm;             // :569
                    st4(a.Zcos + row + col, zc);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The wide correction on the bf16 matrix pipe (hmx_device.h: fp32 operands as three exact bf16 terms, six products).
// k_ridge_apply_wide multiplies with the f32-input MFMA from scalar LDS reads, one tile per wave: 36 % pipe busy and 46 %
// LDS-conflict cycles in round 4's counters, 2.0 ms per pass at the configs[4] shard against an HBM floor of 0.5 ms.  Here:
//   * k_w_planes (one small launch behind the ridge solve): W of every group as the A FRAGMENTS of
//     v_mfma_f32_16x16x32_bf16, split once: Wf[g][step s][plane h, m, l][PC tile mt][lane][8 bf16] -- row i = PC 16 mt + c16,
//     k slot j of lane (c16, q) = cluster 32 s + 8 q + j (zeros past K16).  A (plane, tile) fragment is 1 KB contiguous:
//     one LDS-DMA request brings it, one conflict-free 16-byte read per lane hands it to the matrix pipe;
//   * k_ridge_apply_wideb: one workgroup of eight waves per CU walks a task's tiles sixteen at a time (two per wave: a
//     fragment read feeds two tiles' products), W steps through a two-slot ring (3 MTD KB per slot), the R values of a step
//     (two 16-byte loads per tile and lane) are split in registers; one vmcnt(0) + one barrier per step, the requests of
//     step s+1 go out behind the first PC tile of step s.  Epilogue as in k_ridge_apply_wide (:566, :569).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_w_planes(const float* __restrict__ W, int K16, int ldw, int mtd, int ns, unsigned* __restrict__ Wf) {
    const int lane = threadIdx.x, c16 = lane & 15, q = lane >> 4;
    const int s = blockIdx.x / mtd, mt = blockIdx.x - s * mtd, g = blockIdx.y;
    const float* Wg = W + (size_t)g * K16 * ldw + 16 * mt + c16;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * s + 8 * q + j;
        x[j] = (k < K16) ? Wg[(size_t)k * ldw] : 0.f;
    }
    u32x4 pl[3];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned h, m, l;
        bf16_split3((f32x2){x[2 * p], x[2 * p + 1]}, h, m, l);
        pl[0][p] = h; pl[1][p] = m; pl[2][p] = l;
    }
#pragma unroll
    for (int pn = 0; pn < 3; ++pn)
        *reinterpret_cast<u32x4*>(Wf + ((((size_t)g * ns + s) * 3 + pn) * mtd + mt) * 256 + 4 * lane) = pl[pn];
}

#define APPLYB_WAVES 8
template <int MTD>
__global__ __launch_bounds__(64 * APPLYB_WAVES, 1) void k_ridge_apply_wideb(ApplyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* ring = reinterpret_cast<unsigned*>(smem);              // 2 slots x 3 planes x MTD tiles x 256 dwords
    constexpr int SLOT = 3 * MTD * 256;                              // dwords per slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q = lane >> 4;
    const int task = blockIdx.x;
    if (task >= a.ntasks) return;
    const int t0 = a.task_tile0[task], t1 = a.task_tile1[task], g = a.task_grp[task];
    const int ns = (a.K16 + 31) >> 5;                                // k-steps of 32 clusters
    auto u64 = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    };
    const unsigned long long wsrc = u64((unsigned long long)(a.Wf + (size_t)g * ns * SLOT));
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring);
    const unsigned voff = 16u * (unsigned)lane;
    auto request = [&](int s) {                                      // this wave's fragments of step s: pieces wv, wv + 8, ... of 3 MTD
        const unsigned long long src0 = wsrc + (unsigned long long)s * (SLOT * 4);
        const unsigned zone0 = ring0 + (unsigned)(s & 1) * (SLOT * 4);
#pragma unroll
        for (int j = 0; j < (3 * MTD + APPLYB_WAVES - 1) / APPLYB_WAVES; ++j) {
            const int p = wv + APPLYB_WAVES * j;                     // wave-uniform
            if (p < 3 * MTD) {
                const unsigned long long src = src0 + 1024ull * p;
                const unsigned zone = zone0 + 1024u * p;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(zone) : "memory", "m0");
            }
        }
    };
    for (int base = t0; base < t1; base += 2 * APPLYB_WAVES) {       // workgroup-uniform trip count
        const int ta = base + 2 * wv;
        const bool has0 = ta < t1, has1 = ta + 1 < t1;               // wave-uniform
        const int cell0 = has0 ? a.cells[(size_t)ta * 16 + c16] : -1;
        const int cell1 = has1 ? a.cells[(size_t)(ta + 1) * 16 + c16] : -1;
        const bool live0 = cell0 >= 0, live1 = cell1 >= 0;
        const float* rr0 = a.R + (size_t)(live0 ? cell0 : 0) * a.Kp + 8 * q;
        const float* rr1 = a.R + (size_t)(live1 ? cell1 : 0) * a.Kp + 8 * q;
        f32x4 acc0[MTD], acc1[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) { acc0[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        f32x4 rv[4];                                                 // raw R values of the coming step (dead once the step has split them)
        auto load_r = [&](int s) {                                   // ordinary loads, pinned where they are written; clusters past the row are zeros
            __builtin_amdgcn_sched_barrier(0);
            const int col = 32 * s + 8 * q;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            rv[0] = (live0 && col < a.Kp) ? ld4(rr0 + 32 * s) : zero;
            rv[1] = (live0 && col + 4 < a.Kp) ? ld4(rr0 + 32 * s + 4) : zero;
            rv[2] = (live1 && col < a.Kp) ? ld4(rr1 + 32 * s) : zero;
            rv[3] = (live1 && col + 4 < a.Kp) ? ld4(rr1 + 32 * s + 4) : zero;
            __builtin_amdgcn_sched_barrier(0);
        };
        wg_barrier_lds();                                            // nobody reads the ring any more (the pass before)
        request(0);
        load_r(0);
        auto split4 = [&](const f32x4& lo, const f32x4& hi, u32x4 (&pl)[3]) {
            unsigned h, m, l;
            bf16_split3((f32x2){lo[0], lo[1]}, h, m, l); pl[0][0] = h; pl[1][0] = m; pl[2][0] = l;
            bf16_split3((f32x2){lo[2], lo[3]}, h, m, l); pl[0][1] = h; pl[1][1] = m; pl[2][1] = l;
            bf16_split3((f32x2){hi[0], hi[1]}, h, m, l); pl[0][2] = h; pl[1][2] = m; pl[2][2] = l;
            bf16_split3((f32x2){hi[2], hi[3]}, h, m, l); pl[0][3] = h; pl[1][3] = m; pl[2][3] = l;
        };
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's fragments and R values of step s (requested a whole step ago)
            wg_barrier_lds();                                        // everybody's fragments of step s are in; nobody reads step s-1 any more
            const unsigned* slot = ring + (size_t)(s & 1) * SLOT + 4 * lane;
            u32x4 bp0[3], bp1[3];                                    // the two tiles' B planes of this step
            split4(rv[0], rv[1], bp0);
            split4(rv[2], rv[3], bp1);
            u32x4 wp[2][3];                                          // A planes of the current / the next PC tile
            auto fetch = [&](int mt, u32x4 (&pl)[3]) {
#pragma unroll
                for (int pn = 0; pn < 3; ++pn) pl[pn] = ld4u(slot + (pn * MTD + mt) * 256);
            };
            auto products = [&](int mt, const u32x4 (&pl)[3]) {      // smallest terms first; the two tiles alternate
                acc0[mt] = MFMA_BF16(pl[2], bp0[0], acc0[mt]);  acc1[mt] = MFMA_BF16(pl[2], bp1[0], acc1[mt]);
                acc0[mt] = MFMA_BF16(pl[0], bp0[2], acc0[mt]);  acc1[mt] = MFMA_BF16(pl[0], bp1[2], acc1[mt]);
                acc0[mt] = MFMA_BF16(pl[1], bp0[1], acc0[mt]);  acc1[mt] = MFMA_BF16(pl[1], bp1[1], acc1[mt]);
                acc0[mt] = MFMA_BF16(pl[1], bp0[0], acc0[mt]);  acc1[mt] = MFMA_BF16(pl[1], bp1[0], acc1[mt]);
                acc0[mt] = MFMA_BF16(pl[0], bp0[1], acc0[mt]);  acc1[mt] = MFMA_BF16(pl[0], bp1[1], acc1[mt]);
                acc0[mt] = MFMA_BF16(pl[0], bp0[0], acc0[mt]);  acc1[mt] = MFMA_BF16(pl[0], bp1[0], acc1[mt]);
            };
            fetch(0, wp[0]);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                if (mt + 1 < MTD) fetch(mt + 1, wp[(mt + 1) & 1]);
                products(mt, wp[mt & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (mt == 0 && s + 1 < ns) {                         // behind the first PC tile: slot (s+1) & 1 is free, rv[] is split
                    request(s + 1);
                    load_r(s + 1);
                }
            }
        }
        // ---- Z_corr = Z_orig - W^T (Phi_moe R) (:566), Z_cos = its unit rows (:569) ----
        auto finish = [&](bool has, bool live, int cell, f32x4 (&acc)[MTD]) {
            if (!has) return;
            const size_t row = (size_t)(live ? cell : 0) * a.dp;
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int col = 16 * mt + 4 * q;
                const f32x4 zo = (live && col < a.dp) ? ld4(a.Zorig + row + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[mt] = zo - acc[mt];
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += acc[mt][r] * acc[mt][r];
            }
            ss = wave_sum_q(ss);
            const float nrm = sqrtf(ss);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int col = 16 * mt + 4 * q;
                if (live && col < a.dp) {
                    st4(a.Zcorr + row + col, acc[mt]);
                    f32x4 zc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) zc[r] = acc[mt][r] / nrm;
                    st4(a.Zcos + row + col, zc);
                }
            }
        };
        finish(has0, live0, cell0, acc0);
        finish(has1, live1, cell1, acc1);
    }
}

// ------------------------------------------------------------------------------------------
// Device-side update order (replaces torch.randperm + the gather/argsort of harmony.py:471-480,
// 512-513 when the caller does not supply an order).
//
// A keyed bijection pi on [0, Ng) (cycle-walking Feistel network; Ng = cells of the whole job)
// plays the role of the random permutation: position p of the order holds global cell pi(p);
// block b = positions [b*cpb, (b+1)*cpb), the last block takes the remainder (harmony.py:482-484).
// Every cell finds its own position as p = pi^-1(global id): no rank of a sharded job needs the
// others' cells, and the blocks do not depend on the sharding.  Three passes turn that into the
// engine's lists -- cells of a block regrouped by batch group, every (block, group) run padded
// to 16 -- without atomics whose order could change the result:
//   count   : one wave per chunk of cells, per-chunk histogram over key = block*G + group
//   scan    : exclusive scan over chunks per key (one workgroup per key)
//   runs    : run/tile starts, block_tile_start, padding, tile groups
//   scatter : every cell goes to run_start + (#earlier cells with its key)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// inverse of the 6-round Feistel network  (l, r) -> (r, l ^ F_i(r)), i = 0..5, with cycle walking
__device__ __forceinline__ uint32_t feistel_position(uint32_t x, uint32_t n, int half_bits, uint32_t k0, uint32_t k1) {
    const uint32_t mask = (1u << half_bits) - 1u;
    do {
        uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            const uint32_t f = mix32(l * 0x9E3779B1u + k0 + (uint32_t)i * k1) & mask;
            const uint32_t pl = r ^ f;
            r = l;
            l = pl;
        }
        x = (l << half_bits) | r;
    } while (x >= n);
    return x;
}
__device__ __forceinline__ int group_of_cell(const int* __restrict__ gstart, int G, int cell) {
    int lo = 0, hi = G;  // gstart[g] <= cell < gstart[g+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (gstart[mid] <= cell) lo = mid; else hi = mid;
    }
    return lo;
}

#define ORDER_CHUNK 256  /* cells per wave: short chunks = many waves to hide the serial LDS chain */

// mode 0: write per-chunk histograms; mode 1: scatter cells using chunk offsets.
// chunk_tab is laid out [key][chunk] so that the scan below reads contiguous counts.
template <int MODE>
__global__ __launch_bounds__(64) void k_order_pass(OrderArgs a) {
    if (a.frozen && *a.frozen) return;   // a sweep timed out and the host has not looked yet: these may be the lists it will replay
    extern __shared__ int cnt[];  // nkeys running counters of this chunk
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int nkeys = a.nblk * a.G;
    for (int i = lane; i < nkeys; i += 64) cnt[i] = (MODE == 1) ? a.chunk_tab[(size_t)i * nchunks + chunk] : 0;
    __syncthreads();
    int key_bits = 1;
    while ((1 << key_bits) < nkeys) ++key_bits;
    const int64_t base = (int64_t)chunk * ORDER_CHUNK;
    for (int s = 0; s < ORDER_CHUNK / 64; ++s) {
        const int64_t ci = base + s * 64 + lane;
        const bool live = ci < a.N;
        const int cell = (int)ci;
        int key = 0;
        if (live) {
            const uint32_t gid = a.global_id ? (uint32_t)a.global_id[cell] : (uint32_t)cell;
            const int64_t p = feistel_position(gid, (uint32_t)a.Ng, a.half_bits, a.key0, a.key1);
            const int b = (a.cpb > 0) ? (int)min((int64_t)(p / a.cpb), (int64_t)(a.nblk - 1)) : a.nblk - 1;
            const int g = group_of_cell(a.gstart, a.G, cell);
            key = b * a.G + g;
            // the histogram pass walks the cells in storage order: the block ids the streaming R^T.Z pass wants
            // (static tile order) are consecutive bytes here -- no scatter from the list
            if (MODE == 0 && a.tile_blk) a.tile_blk[(size_t)16 * a.s_tile_start[g] + (cell - a.gstart[g])] = (unsigned char)b;
        }
        if (MODE == 0) {                      // a histogram needs no order
            if (live) atomicAdd(&cnt[key], 1);
            continue;
        }
        // lanes of this step that share the key: one ballot per key bit instead of one round per
        // distinct key; the rank among them in cell order is a population count
        unsigned long long same = __ballot(live);
        for (int bit = 0; bit < key_bits; ++bit) {
            const unsigned long long set = __ballot((key >> bit) & 1);
            same &= ((key >> bit) & 1) ? set : ~set;
        }
        if (live) {
            const int rank = __popcll(same & ((1ull << lane) - 1ull));
            const int before = cnt[key];
            a.cells[a.run_start[key] + before + rank] = cell;
            if ((same >> lane) <= 1ull) cnt[key] = before + __popcll(same);   // the group's last lane moves the counter on
        }
        __syncthreads();
    }
    if (MODE == 0) {
        __syncthreads();
        for (int i = lane; i < nkeys; i += 64) a.chunk_tab[(size_t)i * nchunks + chunk] = cnt[i];
    }
}

// One workgroup per key: exclusive scan of the per-chunk counts (in place) and the run length.
__global__ __launch_bounds__(256) void k_order_scan(OrderArgs a, int nchunks) {
    if (a.frozen && *a.frozen) return;   // a sweep timed out and the host has not looked yet: these may be the lists it will replay
    __shared__ int part[256];
    const int key = blockIdx.x, tid = threadIdx.x;
    int* tab = a.chunk_tab + (size_t)key * nchunks;
    const int per = (nchunks + 255) / 256;
    const int c0 = min(tid * per, nchunks), c1 = min(c0 + per, nchunks);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += tab[c];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // inclusive Hillis-Steele scan of the 256 partial sums
        const int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;                   // exclusive prefix of this thread's chunk range
    for (int c = c0; c < c1; ++c) {
        const int v = tab[c];
        tab[c] = run;
        run += v;
    }
    if (tid == 255) a.run_count[key] = part[255];
}

// One workgroup per key (block, group): its run start (every workgroup sums the padded lengths of the
// keys before its own -- a few hundred integers), block_tile_start, the run's padding and tile groups.
__global__ __launch_bounds__(256) void k_order_runs(OrderArgs a) {
    if (a.frozen && *a.frozen) return;   // a sweep timed out and the host has not looked yet: these may be the lists it will replay
    __shared__ int part[256];
    const int nkeys = a.nblk * a.G;
    const int key = blockIdx.x, tid = threadIdx.x;
    int sum = 0;
    for (int k = tid; k < key; k += 256) sum += ((a.run_count[k] + 15) / 16) * 16;
    part[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) part[tid] += part[tid + s];
        __syncthreads();
    }
    const int start = part[0];
    const int n = a.run_count[key];
    const int padded = ((n + 15) / 16) * 16;
    if (tid == 0) {
        a.run_start[key] = start;
        if (a.run_tiles) a.run_tiles[key] = start / 16;                       // (block, group) runs in tiles: k_round's group-affine map
        if (key % a.G == 0) a.blk_start[key / a.G] = start / 16;
        if (key == nkeys - 1) {
            a.blk_start[a.nblk] = (start + padded) / 16;
            if (a.run_tiles) a.run_tiles[nkeys] = (start + padded) / 16;
        }
    }
    for (int i = n + tid; i < padded; i += 256) a.cells[start + i] = -1;
    for (int t = tid; t < padded / 16; t += 256) a.tile_grp[start / 16 + t] = key % a.G;
}

// rows[i] of a row-major float array -> dense n_rows x cols
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, int ld, int cols, const int* __restrict__ rows,
                                                     int n_rows, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[i] = src[(size_t)rows[r] * ld + c];
}

// Upload path: rows of the caller's matrix (n x d, caller's order) into the engine's padded,
// group-sorted layout: dst[i][0..dp) = src[source_row[i]][0..d), 0 beyond d.
__global__ __launch_bounds__(256) void k_load_rows(const float* __restrict__ src, int d, const int* __restrict__ source_row,
                                                   float* __restrict__ dst, int dp, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * dp) return;
    const int64_t r = i / dp;
    const int c = (int)(i - r * dp);
    const int64_t sr = source_row ? (int64_t)source_row[r] : r;
    dst[i] = c < d ? src[sr * d + c] : 0.f;
}

// ------------------------------------------------------------------------------------------
// k-means++ seeding on the device (the greedy k-means++ of sklearn 1.7's `_kmeans_plusplus`, which
// the reference reaches through KMeans(init='k-means++'), harmony.py:370): the first centre uniform,
// then per centre SEED_TRIALS candidates drawn with probability proportional to the squared distance
// to the closest centre so far; the candidate that leaves the smallest total is kept.
// Everything that decides an index is integer arithmetic: squared distances are accumulated in a
// fixed order without contraction (bit-equal to a NumPy float32 loop), potentials are 32.32
// fixed-point sums (order-independent, so atomics are exact), the draws come from a counter-based
// generator, and a draw is located by an exact prefix search.  oracle/kmeans_seed.py restates it.
// Three small launches per centre: pick (1 workgroup) -> eval (one thread per point) -> commit.
// ------------------------------------------------------------------------------------------
#define SEED_TRIALS 8          // slots per step; the first n_trials are used
__device__ __forceinline__ unsigned long long seed_rand(unsigned long long seed, int step, int trial) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(1 + step * SEED_TRIALS + trial);
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ unsigned long long seed_fx(float v) { return (unsigned long long)(v * 4294967296.0f); }

// 256-thread inclusive scan of one unsigned long long per thread (LDS scratch of 256 words)
__device__ __forceinline__ unsigned long long seed_scan256(unsigned long long v, unsigned long long* sh, int tid) {
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long o = tid >= off ? sh[tid - off] : 0ull;
        __syncthreads();
        sh[tid] += o;
        __syncthreads();
    }
    const unsigned long long r = sh[tid];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_seed_transpose(const float* __restrict__ X, int n, int d, float* __restrict__ Xt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * d) return;
    const int c = (int)(i / n), r = (int)(i - (int64_t)c * n);
    Xt[i] = X[(size_t)r * d + c];
}

// candidates of one step -> a.cand[step*SEED_TRIALS + j]
__global__ __launch_bounds__(256) void k_seed_pick(SeedArgs a, int step) {
    __shared__ unsigned long long sh[256];
    __shared__ unsigned long long tot_s;
    __shared__ int chunk_s;
    __shared__ unsigned long long rem_s;
    const int tid = threadIdx.x;
    int* cand = a.cand + step * SEED_TRIALS;
    if (step == 0) {
        if (tid == 0) cand[0] = (int)__umul64hi(seed_rand(a.seed, 0, 0), (unsigned long long)a.n);
        return;
    }
    const int nchunks = (a.n + 255) / 256;
    // prefix over the chunk sums (nchunks <= 256 * 16: a thread owns up to 16 consecutive chunks)
    const int per = (nchunks + 255) / 256;
    unsigned long long mine = 0;
    for (int c = 0; c < per; ++c) { const int ch = tid * per + c; if (ch < nchunks) mine += a.chunk_sum[ch]; }
    const unsigned long long incl = seed_scan256(mine, sh, tid);
    if (tid == 255) tot_s = incl;
    __syncthreads();
    const unsigned long long total = tot_s;
    for (int j = 0; j < a.n_trials; ++j) {
        const unsigned long long u = seed_rand(a.seed, step, j);
        if (total == 0ull) {                                       // every point coincides with a centre
            if (tid == 0) cand[j] = (int)__umul64hi(u, (unsigned long long)a.n);
            continue;
        }
        const unsigned long long target = __umul64hi(u, total);    // uniform in [0, total)
        const unsigned long long excl = incl - mine;
        if (target >= excl && target < incl) {                     // exactly one thread
            unsigned long long acc = excl;
            int ch = tid * per;
            for (int c = 0; c < per; ++c, ++ch) {
                const unsigned long long v = a.chunk_sum[ch];
                if (target < acc + v) break;
                acc += v;
            }
            chunk_s = ch; rem_s = target - acc;
        }
        __syncthreads();
        const int ch = chunk_s;
        const unsigned long long rem = rem_s;
        const int i = ch * 256 + tid;
        const unsigned long long v = i < a.n ? seed_fx(a.closest[i]) : 0ull;
        const unsigned long long inc2 = seed_scan256(v, sh, tid);
        if (rem >= inc2 - v && rem < inc2) cand[j] = i;
        __syncthreads();
    }
}

// squared distance of every point to every candidate of the step, kept as min(closest, .)
__global__ __launch_bounds__(256) void k_seed_eval(SeedArgs a, int step) {
    extern __shared__ float cs[];                                   // n_trials x d candidate rows
    __shared__ unsigned long long red[4 * SEED_TRIALS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nc = step == 0 ? 1 : a.n_trials;
    const int* cand = a.cand + step * SEED_TRIALS;
    for (int i = tid; i < nc * a.d; i += 256) { const int j = i / a.d, c = i - j * a.d; cs[i] = a.X[(size_t)cand[j] * a.d + c]; }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    const bool live = i < a.n;
    float acc[SEED_TRIALS];
#pragma unroll
    for (int j = 0; j < SEED_TRIALS; ++j) acc[j] = 0.f;
    for (int c = 0; c < a.d; ++c) {
        const float x = live ? a.Xt[(size_t)c * a.n + i] : 0.f;
#pragma unroll
        for (int j = 0; j < SEED_TRIALS; ++j)
            if (j < nc) { const float df = __fsub_rn(x, cs[j * a.d + c]); acc[j] = __fadd_rn(acc[j], __fmul_rn(df, df)); }
    }
    const float cl = (live && step > 0) ? a.closest[i] : __builtin_inff();
#pragma unroll
    for (int j = 0; j < SEED_TRIALS; ++j) {
        if (j >= nc) break;
        const float m = fminf(cl, acc[j]);
        if (live) a.cand_min[(size_t)j * a.n + i] = m;
        unsigned long long f = live ? seed_fx(m) : 0ull;
        for (int off = 32; off; off >>= 1) f += __shfl_xor(f, off);
        if (lane == 0) red[wv * SEED_TRIALS + j] = f;
    }
    __syncthreads();
    if (tid < nc) atomicAdd(a.pots + step * SEED_TRIALS + tid, red[tid] + red[SEED_TRIALS + tid] + red[2 * SEED_TRIALS + tid] + red[3 * SEED_TRIALS + tid]);
}

// keep the best candidate: closest <- its minima, chunk sums for the next draw, centre row out
__global__ __launch_bounds__(256) void k_seed_commit(SeedArgs a, int step) {
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nc = step == 0 ? 1 : a.n_trials;
    int best = 0;
    unsigned long long bp = a.pots[step * SEED_TRIALS];
    for (int j = 1; j < nc; ++j) { const unsigned long long pj = a.pots[step * SEED_TRIALS + j]; if (pj < bp) { bp = pj; best = j; } }
    const int i = blockIdx.x * 256 + tid;
    float m = 0.f;
    if (i < a.n) { m = a.cand_min[(size_t)best * a.n + i]; a.closest[i] = m; }
    unsigned long long f = i < a.n ? seed_fx(m) : 0ull;
    for (int off = 32; off; off >>= 1) f += __shfl_xor(f, off);
    if (lane == 0) red[wv] = f;
    __syncthreads();
    if (tid == 0) a.chunk_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0) {
        const int src = a.cand[step * SEED_TRIALS + best];
        if (tid == 0) a.chosen[step] = src;
        for (int c = tid; c < a.d; c += 256) a.centers[(size_t)step * a.d + c] = a.X[(size_t)src * a.d + c];
    }
}

// ------------------------------------------------------------------------------------------
// host-side launchers (called from hmx_capi.cpp through hmx_internal.h)
// ------------------------------------------------------------------------------------------
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void launch_normalize_rows(const float* Z, float* Zc, int64_t N, int dp, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(k_normalize_rows, dim3(cdiv(N, 16)), dim3(256), 0, s, Z, Zc, N, dp);
}

void launch_y_normalize(const float* src, float* dst, int K, int K16, int d, int ldy, hipStream_t s) {
    hipLaunchKernelGGL(k_y_normalize<float>, dim3(K16), dim3(64), 0, s, src, dst, K, d, ldy);
}
void launch_y_normalize_d(const double* src, float* dst, int K, int K16, int d, int ldy, hipStream_t s) {
    hipLaunchKernelGGL(k_y_normalize<double>, dim3(K16), dim3(64), 0, s, src, dst, K, d, ldy);
}

int assign_grid(int ntiles, int nt_per_step, int max_wgs) {
    const int steps = cdiv(ntiles, nt_per_step);
    int wgs = cdiv(steps, 4);
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs < 1) wgs = 1;
    return wgs;
}

static int lds_ldy(int dp) { return ((dp >> 2) & 1) ? dp : dp + 4; }  // (LDY/4) odd: spreads ds_read_b128 rows over banks

template <int MT>
static void launch_assign_lds(const AssignArgs& a, bool penalty, int wgs, size_t sm, hipStream_t s) {
    if (penalty) hipLaunchKernelGGL((k_assign_lds<MT, true>), dim3(wgs), dim3(64 * ASSIGN_WAVES), sm, s, a);
    else hipLaunchKernelGGL((k_assign_lds<MT, false>), dim3(wgs), dim3(64 * ASSIGN_WAVES), sm, s, a);
}

size_t assign_wide3_lds_bytes(int mt) {
    const size_t K16 = 16 * (size_t)mt;
    return (size_t)2 * 3 * mt * 1024 + (2 * K16 + 2 * WIDE3_SLOTS * K16) * sizeof(float) + (WIDE3_SLOTS * K16 + 2 * WIDE3_WAVES + WIDE3_TSUB * K16) * sizeof(double) +
           5 * WIDE3_SLOTS * sizeof(int);
}
// the whole wide sweep in one persistent launch (k_sweep_wide3): one batch variable, at most 32 groups, lists with run offsets
size_t sweep_wide3_lds_bytes(int mt, int nblk) {
    const size_t K16 = 16 * (size_t)mt;
    return (size_t)2 * 3 * mt * 1024 + (2 * K16 + 2 * WIDE3_SLOTS * K16) * sizeof(float) + (WIDE3_SLOTS * K16 + 2 * WIDE3_WAVES + WIDE3_TSUB * K16) * sizeof(double) +
           (4 * WIDE3_SLOTS + 64 + nblk + 2) * sizeof(int);
}
bool sweep_wide3_ok(int mt, int dp, int V, int G, int nblk) {
    return V == 1 && G <= 32 && dp % 16 == 0 && mt >= 1 && mt <= 13 && (mt > 7 || dp > 64) && sweep_wide3_lds_bytes(mt, nblk) <= 160 * 1024;
}
int launch_sweep_wide3(const AssignArgs& a, int wgs, hipStream_t s) {
    if (!sweep_wide3_ok(a.mt, a.dp, 1, a.G, a.nblk) || !a.Yf || !a.run_tiles || !a.fail || !a.O_priv || !a.O_out || wgs < 1) return -1;
    const size_t sm = sweep_wide3_lds_bytes(a.mt, a.nblk);
#define HMX_SWEEP3_CASE(M)                                                                                              \
    case M: {                                                                                                         \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_wide3<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((k_sweep_wide3<M>), dim3(wgs), dim3(64 * WIDE3_WAVES), sm, s, a);                           \
        sweep3_prof_dump(wgs, a.nblk, s);                                                                             \
    } break;
    switch (a.mt) {
        HMX_SWEEP3_CASE(1) HMX_SWEEP3_CASE(2) HMX_SWEEP3_CASE(3) HMX_SWEEP3_CASE(4) HMX_SWEEP3_CASE(5) HMX_SWEEP3_CASE(6) HMX_SWEEP3_CASE(7)
        HMX_SWEEP3_CASE(8) HMX_SWEEP3_CASE(9) HMX_SWEEP3_CASE(10) HMX_SWEEP3_CASE(11) HMX_SWEEP3_CASE(12) HMX_SWEEP3_CASE(13)
        default: return -1;
    }
#undef HMX_SWEEP3_CASE
    return 0;
}
// the wide block assignment can build its own diversity table (one batch variable, the bf16-pipe instance)
bool assign_wide3_fuses_table(int mt, int dp, int V) { return V == 1 && dp % 16 == 0 && mt >= 1 && mt <= 13 && (mt > 7 || dp > 64) && assign_wide3_lds_bytes(mt) <= 160 * 1024; }

int launch_assign(const AssignArgs& a_in, bool penalty, int max_wgs, hipStream_t s) {
    AssignArgs a = a_in;
    const int ntiles = a.tile_end - a.tile_begin;
    if (ntiles <= 0) return 0;
    // LDS-resident path: Y, sigma, wave tiles (+ rp, lrp, fp64 block sums when they fit)
    const int LDY = lds_ldy(a.dp);
    const size_t base_bytes = ((size_t)a.K16 * LDY + 2 * a.K16 + (size_t)ASSIGN_WAVES * 16 * LDY) * 4 + ASSIGN_WAVES * 16;
    const size_t tab_bytes = (size_t)a.G * a.K16 * (penalty ? 16 : 8);
    if (a.mt <= 7 && a.dp <= 64 && base_bytes <= 100 * 1024) {
        a.ldy_lds = LDY;
        a.tables_in_lds = (base_bytes + tab_bytes <= 150 * 1024) ? 1 : 0;
        const size_t sm = base_bytes + tab_bytes;   // layout keeps the table slots even when unused
        if (sm <= 160 * 1024) {
            const int wgs = cdiv(ntiles, ASSIGN_WAVES);   // one tile per wave
            switch (a.mt) {
                case 1: launch_assign_lds<1>(a, penalty, wgs, sm, s); break;
                case 2: launch_assign_lds<2>(a, penalty, wgs, sm, s); break;
                case 3: launch_assign_lds<3>(a, penalty, wgs, sm, s); break;
                case 4: launch_assign_lds<4>(a, penalty, wgs, sm, s); break;
                case 5: launch_assign_lds<5>(a, penalty, wgs, sm, s); break;
                case 6: launch_assign_lds<6>(a, penalty, wgs, sm, s); break;
                default: launch_assign_lds<7>(a, penalty, wgs, sm, s); break;
            }
            return 0;
        }
    }
    if (a.dp % 16 == 0 && a.dp <= 208 && a.mt >= 1 && a.mt <= 13 && (a.mt > 7 || a.dp > 64)) {
        // wide shapes: k-step staged centroid columns, 8 tiles per workgroup pass (K, d <= 208; beyond: the generic kernel below)
        const size_t gk_bytes = (size_t)a.G * a.K16 * sizeof(double);
        a.tables_in_lds = gk_bytes <= 64 * 1024 ? 1 : 0;
        const size_t sm = ((size_t)2 * a.K16 * 20 + 2 * a.K16) * sizeof(float) + (a.tables_in_lds ? gk_bytes : 0) + 2 * WIDE_WAVES * sizeof(double);
        // the penalised block assignment of the round loop on the bf16 matrix pipe, centroids pre-split into fragments
        // (k_assign_wide3: eight waves, two tiles each, one workgroup per CU); engines created under HMX_ROUND_F32=1 and the
        // assignments without a penalty (init_cluster, the device Lloyd) take the f32-input kernel k_assign_wide below
        if (a.fuse_table && !(penalty && !a.hn && a.bf16_pipe && a.Yf && assign_wide3_lds_bytes(a.mt) <= 160 * 1024)) return -1;   // only k_assign_wide3 builds its own table
        if (penalty && !a.hn && a.bf16_pipe && a.Yf) {
            const size_t sm3 = assign_wide3_lds_bytes(a.mt);
            const int wgs3 = cdiv(ntiles, WIDE3_SLOTS);
#define HMX_WIDE3_CASE(M)                                                                                               \
    case M: {                                                                                                         \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_assign_wide3<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((k_assign_wide3<M>), dim3(wgs3), dim3(64 * WIDE3_WAVES), sm3, s, a);                        \
        wide3_prof_dump(wgs3, s);                                                                                     \
    } break;
            if (sm3 <= 160 * 1024) {
                switch (a.mt) {
                    HMX_WIDE3_CASE(1) HMX_WIDE3_CASE(2) HMX_WIDE3_CASE(3) HMX_WIDE3_CASE(4) HMX_WIDE3_CASE(5) HMX_WIDE3_CASE(6) HMX_WIDE3_CASE(7)
                    HMX_WIDE3_CASE(8) HMX_WIDE3_CASE(9) HMX_WIDE3_CASE(10) HMX_WIDE3_CASE(11) HMX_WIDE3_CASE(12) HMX_WIDE3_CASE(13)
                }
                return 1;
            }
#undef HMX_WIDE3_CASE
        }
        const int wgs = std::max(1, std::min(2 * 256, cdiv(ntiles, WIDE_WAVES)));
#define HMX_WIDE_CASE(M)                                                                                          \
    case M:                                                                                                       \
        if (penalty) hipLaunchKernelGGL((k_assign_wide<M, true>), dim3(wgs), dim3(64 * WIDE_WAVES), sm, s, a);    \
        else hipLaunchKernelGGL((k_assign_wide<M, false>), dim3(wgs), dim3(64 * WIDE_WAVES), sm, s, a);           \
        break;
        switch (a.mt) {
            HMX_WIDE_CASE(1) HMX_WIDE_CASE(2) HMX_WIDE_CASE(3) HMX_WIDE_CASE(4) HMX_WIDE_CASE(5) HMX_WIDE_CASE(6) HMX_WIDE_CASE(7)
            HMX_WIDE_CASE(8) HMX_WIDE_CASE(9) HMX_WIDE_CASE(10) HMX_WIDE_CASE(11) HMX_WIDE_CASE(12) HMX_WIDE_CASE(13)
        }
#undef HMX_WIDE_CASE
        return 0;
    }
    if (a.mt <= 7) {
        constexpr int NT = 2;
        const int wgs = assign_grid(ntiles, NT, max_wgs);
        if (penalty) hipLaunchKernelGGL((k_assign<7, NT, true>), dim3(wgs), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_assign<7, NT, false>), dim3(wgs), dim3(256), 0, s, a);
    } else if (a.mt <= 13) {
        constexpr int NT = 1;
        const int wgs = assign_grid(ntiles, NT, max_wgs);
        if (penalty) hipLaunchKernelGGL((k_assign<13, NT, true>), dim3(wgs), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_assign<13, NT, false>), dim3(wgs), dim3(256), 0, s, a);
    } else if (a.mt <= 20) {   // up to 320 clusters: the generic kernel's widest instance (51 KB of staged centroid columns)
        constexpr int NT = 1;
        const int wgs = assign_grid(ntiles, NT, max_wgs);
        if (penalty) hipLaunchKernelGGL((k_assign<20, NT, true>), dim3(wgs), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_assign<20, NT, false>), dim3(wgs), dim3(256), 0, s, a);
    } else {
        return -1;
    }
    return 0;
}

size_t round_lds_bytes(int K16, int dp, int G, int B, int V, bool bf3, bool ga, int nblk) {
    const size_t GK = (size_t)(ga ? 1 : G) * K16;   // group-affine map: a workgroup keeps the tables of its own group only
    // sigma, -1/sigma, rp, lrp, rpc (V > 1) | O, S, T, objective scratch (fp64) | Pr_b, theta, group_cols (G x V), bgrp, run offsets and
    // lengths (2 x (nblk + 3)), one flag | landing zones
    const size_t ys = bf3 ? (size_t)3 * K16 * bf3_ldb(dp / 4) : (size_t)K16 * lds_ldy(dp);   // centroids: three bf16 planes, or fp32 rows
    return (ys + 2 * (size_t)K16 + 2 * GK + (V == 1 ? 0 : (size_t)K16 * B)) * 4 + (2 * GK + K16 + 2 * ROUND_WAVES) * 8 +
           (3 * (size_t)B + (size_t)G * (size_t)(V > 8 ? V : 8) + 2 * ((size_t)nblk + 3) + 2) * 4 + 16 + (size_t)ROUND_WAVES * ROUND_TPW * 16 * dp * 4
        ;
}

size_t peer_box_doubles(int n_ranks, size_t GK) { return box_flags(n_ranks, GK) + 2 * (size_t)n_ranks + 2 * (size_t)n_ranks + 8; }

// Self-test of the peer boxes, with time-outs: `iters` exchange cycles (8; HMX_PEER_SELFTEST_ITERS soaks) of exactly the pattern k_round
// uses -- a payload of system-scope stores into every rank's box, s_waitcnt vmcnt(0), barrier, then
// one flag word per peer -- and on the receiving side: poll the flag words, then read every rank's
// payload with system-scope loads and compare.  A payload that is not complete when its flag is
// visible (ordering over xGMI), a mapping that does not reach the peer, or a peer that never
// answers all give result 0.  Acknowledge words keep a fast rank from overwriting a payload that a
// slow rank is still checking.
__global__ __launch_bounds__(256) void k_peer_selftest(double* const* peer_box, double* my_box, int n_ranks, int rank, size_t GK,
                                                      unsigned long long token, int iters, unsigned* result) {
    __shared__ int bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t tok0 = box_flags(n_ranks, GK) + 2 * (size_t)n_ranks;   // token words, then acknowledge words
    const size_t ack0 = tok0 + n_ranks;
    const int npay = (int)min((size_t)1024, GK);
    unsigned long long* mine = reinterpret_cast<unsigned long long*>(my_box);
    if (tid == 0) bad = 0;
    __syncthreads();
    auto wait_all = [&](size_t word0, unsigned long long want) {   // wave 0: all ranks' words in MY box == want
        if (wv == 0) {
            unsigned spins = 0;
            while (true) {
                const bool ok = lane >= n_ranks || ld_sys(mine + word0 + lane) == want;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 21)) { if (lane == 0) bad = 1; break; }
            }
        }
        __syncthreads();
    };
    for (int it = 0; it < iters && !bad; ++it) {
        const unsigned long long tk = token * 65536ull + (unsigned long long)it;
        const int par = it & 1;
        for (int r = 0; r < n_ranks; ++r)
#if HMX_ROUND_RETURNING
            for (int i = tid; i < npay; i += 256) xchg_sys(peer_box[r] + box_data(n_ranks, GK, par, rank) + i, (double)(tk % 1000003ull) + i);
#else
            for (int i = tid; i < npay; i += 256) st_sys(peer_box[r] + box_data(n_ranks, GK, par, rank) + i, (double)(tk % 1000003ull) + i);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < n_ranks) st_sys(reinterpret_cast<unsigned long long*>(peer_box[tid]) + tok0 + rank, tk);
        wait_all(tok0, tk);
        if (bad) break;
        int wrong = 0;
        for (int r = 0; r < n_ranks; ++r)
            for (int i = tid; i < npay; i += 256)
                wrong |= ld_sys(my_box + box_data(n_ranks, GK, par, r) + i) != (double)(tk % 1000003ull) + i;
        if (wrong) bad = 1;
        // the stale-line case: touch the lines of the OTHER parity -- the ones every peer overwrites in the next cycle -- with the
        // loads the sweep kernel uses, BEFORE the acknowledge that lets the peers go on.  Whatever copy of those lines this GPU
        // keeps from now on is stale by construction; the next cycle's check must still see the peers' new payload.
        double touch = 0.0;
        for (int r = 0; r < n_ranks; ++r)
            for (int i = tid; i < npay; i += 256) touch += ld_sys(my_box + box_data(n_ranks, GK, par ^ 1, r) + i);
        asm volatile("" ::"v"(touch));
        __syncthreads();
        if (tid < n_ranks) st_sys(reinterpret_cast<unsigned long long*>(peer_box[tid]) + ack0 + rank, tk);
        wait_all(ack0, tk);
    }
    __syncthreads();
    if (tid == 0) *result = bad ? 0u : 1u;
}

void launch_peer_selftest(double* const* peer_box, double* my_box, int n_ranks, int rank, size_t GK, unsigned long long token, int iters,
                          unsigned* result, hipStream_t s) {
    hipLaunchKernelGGL(k_peer_selftest, dim3(1), dim3(256), 0, s, peer_box, my_box, n_ranks, rank, GK, token, iters, result);
}

// k_round is compiled for Z_cos rows of 32, 52 and 64 floats (d <= 32, <= 52, <= 64: the engine pads
// its rows to the next of these) and 1..7 cluster tiles.
int round_row_floats(int d) { return d <= 32 ? 32 : d <= 52 ? 52 : d <= 64 ? 64 : 0; }

template <int MT, int KS, bool BF3>
static void launch_round_t(const RoundArgs& a, int wgs, size_t sm, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_round<MT, KS, BF3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_round<MT, KS, BF3>), dim3(wgs), dim3(ROUND_THREADS), sm, s, a);
}
template <int KS, bool BF3>
static void launch_round_ks(const RoundArgs& a, int mt, int wgs, size_t sm, hipStream_t s) {
    switch (mt) {
        case 1: launch_round_t<1, KS, BF3>(a, wgs, sm, s); break;
        case 2: launch_round_t<2, KS, BF3>(a, wgs, sm, s); break;
        case 3: launch_round_t<3, KS, BF3>(a, wgs, sm, s); break;
        case 4: launch_round_t<4, KS, BF3>(a, wgs, sm, s); break;
        case 5: launch_round_t<5, KS, BF3>(a, wgs, sm, s); break;
        case 6: launch_round_t<6, KS, BF3>(a, wgs, sm, s); break;
        default: launch_round_t<7, KS, BF3>(a, wgs, sm, s); break;
    }
}
// The bf16-pipe form of the distance GEMM (round_compute_bf3) keeps the centroids as three bf16 planes: 25 KB more LDS at
// K = 100, d = 50 than the f32-input form, whose instances stay for the shapes that need the room for their tables (21 to 30
// batch groups at that K), for blocks larger than the grid carries (`extra_tiles`: most tiles then go through the unpipelined
// extra-tile loop, one tile per wave at a time, where the split is not hidden -- all 10 M cells of configs[3] on one GPU:
// 3.22 ms per sweep against 3.03, profiles/r04_ab_k_round_bf16_pipe.txt) -- and for engines created under HMX_ROUND_F32=1 (`allow_bf16` false), the switch of the A/B runs and of the direct A/B test.
bool round_uses_bf16_pipe(int K16, int dp, int G, int B, int V, bool extra_tiles, bool allow_bf16, bool ga, int nblk) {
    return HMX_ROUND_BF3 && HMX_ROUND_EXP2 && allow_bf16 && !extra_tiles && round_lds_bytes(K16, dp, G, B, V, true, ga, nblk) <= HMX_ROUND_LDS_LIMIT;
}

int launch_round(const RoundArgs& a_in, int mt, int wgs, hipStream_t s, bool extra_tiles, bool allow_bf16) {
    RoundArgs a = a_in;
    a.ldy_lds = lds_ldy(a.dp);
    const bool ga = a.ga != 0;
    if (ga && (a.V != 1 || a.n_ranks > 1 || !a.run_start || !a.wg_map || a.ga_slots < 1 || a.ga_slots > HMX_GA_SLOTS)) return -1;
    const bool bf3 = round_uses_bf16_pipe(a.K16, a.dp, a.G, a.B, a.V, extra_tiles, allow_bf16, ga, a.nblk);
    const size_t sm = round_lds_bytes(a.K16, a.dp, a.G, a.B, a.V, bf3, ga, a.nblk);
    if (mt < 1 || mt > 7 || sm > HMX_ROUND_LDS_LIMIT) return -1;
    switch (a.dp) {
        case 32: if (bf3) launch_round_ks<8, true>(a, mt, wgs, sm, s); else launch_round_ks<8, false>(a, mt, wgs, sm, s); break;
        case 52: if (bf3) launch_round_ks<13, true>(a, mt, wgs, sm, s); else launch_round_ks<13, false>(a, mt, wgs, sm, s); break;
        case 64: if (bf3) launch_round_ks<16, true>(a, mt, wgs, sm, s); else launch_round_ks<16, false>(a, mt, wgs, sm, s); break;
        default: return -1;
    }
    return 0;
}

// k_rtz2 applies when a tile's rows fit its LDS layout: K <= 112 (7 cluster tiles), d <= 64
bool rtz2_ok(int mt, int dp) { return mt >= 1 && mt <= 7 && (dp == 32 || dp == 52 || dp == 64); }
int rtz2_slab_floats(int mt, int dp) { return mt * (dp == 32 ? 2 : 4) * 256; }

template <int NTD, bool ONES>
static void launch_rtz2_n(RtzArgs& a, int mt, int wgs, size_t sm, hipStream_t s) {
    switch (mt) {
        case 1: hipLaunchKernelGGL((k_rtz2<1, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
        case 2: hipLaunchKernelGGL((k_rtz2<2, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
        case 3: hipLaunchKernelGGL((k_rtz2<3, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
        case 4: hipLaunchKernelGGL((k_rtz2<4, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
        case 5: hipLaunchKernelGGL((k_rtz2<5, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
        case 6: hipLaunchKernelGGL((k_rtz2<6, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
        default: hipLaunchKernelGGL((k_rtz2<7, NTD, ONES>), dim3(wgs), dim3(256), sm, s, a); break;
    }
}

#ifdef RTZ_PROF
void rtz_prof_dump() {
    unsigned long long h[8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rtz_prof), sizeof h);
    const double n = (double)h[5];   // wave-tiles
    fprintf(stderr, "[k_rtz2 prof] cycles per tile per wave: pre-barrier %.0f, barrier %.0f, stash+fetch issue %.0f, bookkeeping %.0f, LDS reads+MFMA %.0f\n",
            h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rtz_prof), z, sizeof z);
}
#endif

void launch_rtz2(const RtzArgs& a_in, int wgs, hipStream_t s) {
    RtzArgs a = a_in;
    const int ntd = a.dp == 32 ? 2 : 4;
    a.ldr = ((a.K16 + 31) / 32) * 32 + 16;          // row strides = 16 (mod 32) floats: conflict-free fragment reads
    a.ldz = ((16 * ntd + 31) / 32) * 32 + 16;
    const size_t sm = (size_t)2 * 16 * (a.ldr + a.ldz) * sizeof(float) + (256 + 256) * sizeof(int) + (size_t)a.K16 * sizeof(double);
    if (ntd == 2) launch_rtz2_n<2, false>(a, a.mt, wgs, sm, s);
    else if (a.dp < 64) launch_rtz2_n<4, true>(a, a.mt, wgs, sm, s);     // a padding column carries the column sums of R
    else launch_rtz2_n<4, false>(a, a.mt, wgs, sm, s);
}

void launch_rtz2_reduce(const float* slab, int nslabs, int mt, int dp, int K16, int ld, double* out, const int* task_grp,
                        hipStream_t s) {
    const int ntd = dp == 32 ? 2 : 4;
    const int seg_len = 8;   // slabs summed serially by one thread: short chains, the loads are latency-bound
    hipLaunchKernelGGL(k_rtz2_reduce, dim3(cdiv(mt * ntd * 256, 256), cdiv(nslabs, seg_len)), dim3(256), 0, s, slab, nslabs, mt,
                       ntd, K16, ld, out, task_grp, seg_len);
}

// ---- device k-means (Lloyd) ---------------------------------------------------------------------
size_t kmeans_slab_floats(int wgs, int K16, int dp) { return (size_t)wgs * K16 * (lds_ldy(dp) + 1); }

int launch_kmeans_step(const float* Zcos, const float* C, const float* hn, const int* cells, int n_tiles, float* slab, int K,
                       int K16, int dp, int ldy, int wgs, hipStream_t s) {
    KmeansArgs a{};
    a.Zcos = Zcos; a.C = C; a.hn = hn; a.cells = cells; a.n_tiles = n_tiles; a.slab = slab;
    a.K = K; a.K16 = K16; a.dp = dp; a.ldy = ldy; a.ldy_lds = lds_ldy(dp); a.ldc = lds_ldy(dp);
    const int mt = K16 / 16;
    if (mt < 1 || mt > 7 || dp > 64) return -1;
    const size_t sm = ((size_t)K16 * a.ldy_lds + K16 + (size_t)K16 * a.ldc + K16 + (size_t)8 * 16 * KM_LDZ) * sizeof(float);
    switch (mt) {
        case 1: hipLaunchKernelGGL((k_kmeans_step<1>), dim3(wgs), dim3(512), sm, s, a); break;
        case 2: hipLaunchKernelGGL((k_kmeans_step<2>), dim3(wgs), dim3(512), sm, s, a); break;
        case 3: hipLaunchKernelGGL((k_kmeans_step<3>), dim3(wgs), dim3(512), sm, s, a); break;
        case 4: hipLaunchKernelGGL((k_kmeans_step<4>), dim3(wgs), dim3(512), sm, s, a); break;
        case 5: hipLaunchKernelGGL((k_kmeans_step<5>), dim3(wgs), dim3(512), sm, s, a); break;
        case 6: hipLaunchKernelGGL((k_kmeans_step<6>), dim3(wgs), dim3(512), sm, s, a); break;
        default: hipLaunchKernelGGL((k_kmeans_step<7>), dim3(wgs), dim3(512), sm, s, a); break;
    }
    return 0;
}

void launch_kmeans_sums(const float* slab, int wgs, int K16, int dp, int d, double* sums, hipStream_t s) {
    hipLaunchKernelGGL(k_kmeans_sums, dim3(K16), dim3(256), 0, s, slab, wgs, K16, lds_ldy(dp), d, sums);
}

void launch_kmeans_sums_from_stats(const double* Sr, const double* Oxr, int G, int K16, int ld, int d, double* sums, hipStream_t s) {
    hipLaunchKernelGGL(k_kmeans_sums_from_stats, dim3(K16), dim3(256), 0, s, Sr, Oxr, G, K16, ld, d, sums);
}

void launch_kmeans_update(const double* sums, float* C, float* hn, int K, int K16, int d, int ldy, hipStream_t s) {
    hipLaunchKernelGGL(k_kmeans_update, dim3(K16), dim3(64), 0, s, sums, C, hn, K, d, ldy);
}

bool rtz_wide_ok(int mt, int dp) { return mt >= 1 && mt <= 13 && dp % 16 == 0 && dp <= 208 && (mt > 7 || dp > 64); }
int rtz_wide_slab_floats(int mt, int dp) { return mt * (dp / 16) * 256; }

void launch_rtz_wide(const RtzArgs& a_in, int wgs, hipStream_t s) {
    RtzArgs a = a_in;
    const int ntd = a.dp / 16;
    a.ntd = ntd;
    a.ldr = ((a.K16 + 31) / 32) * 32 + 16;
    a.ldz = ((16 * ntd + 31) / 32) * 32 + 16;
    const size_t sm = (size_t)2 * 16 * (a.ldr + a.ldz) * sizeof(float) + (256 + 256) * sizeof(int);
    switch (a.mt) {
        case 1: hipLaunchKernelGGL((k_rtz_wide<1>), dim3(wgs), dim3(512), sm, s, a); break;
        case 2: hipLaunchKernelGGL((k_rtz_wide<2>), dim3(wgs), dim3(512), sm, s, a); break;
        case 3: hipLaunchKernelGGL((k_rtz_wide<3>), dim3(wgs), dim3(512), sm, s, a); break;
        case 4: hipLaunchKernelGGL((k_rtz_wide<4>), dim3(wgs), dim3(512), sm, s, a); break;
        case 5: hipLaunchKernelGGL((k_rtz_wide<5>), dim3(wgs), dim3(512), sm, s, a); break;
        case 6: hipLaunchKernelGGL((k_rtz_wide<6>), dim3(wgs), dim3(512), sm, s, a); break;
        case 7: hipLaunchKernelGGL((k_rtz_wide<7>), dim3(wgs), dim3(512), sm, s, a); break;
        case 8: hipLaunchKernelGGL((k_rtz_wide<8>), dim3(wgs), dim3(512), sm, s, a); break;
        case 9: hipLaunchKernelGGL((k_rtz_wide<9>), dim3(wgs), dim3(512), sm, s, a); break;
        case 10: hipLaunchKernelGGL((k_rtz_wide<10>), dim3(wgs), dim3(512), sm, s, a); break;
        case 11: hipLaunchKernelGGL((k_rtz_wide<11>), dim3(wgs), dim3(512), sm, s, a); break;
        case 12: hipLaunchKernelGGL((k_rtz_wide<12>), dim3(wgs), dim3(512), sm, s, a); break;
        default: hipLaunchKernelGGL((k_rtz_wide<13>), dim3(wgs), dim3(512), sm, s, a); break;
    }
}

void launch_rtz_wide_reduce(const float* slab, int nslabs, int mt, int dp, int K16, int ld, double* out, const int* task_grp,
                            hipStream_t s) {
    const int ntd = dp / 16;
    const int seg_len = 8;
    hipLaunchKernelGGL(k_rtz2_reduce, dim3(cdiv(mt * ntd * 256, 256), cdiv(nslabs, seg_len)), dim3(256), 0, s, slab, nslabs, mt,
                       ntd, K16, ld, out, task_grp, seg_len);
}

void rtz_geometry(int mt, int ntd, int* nsub, int* slab_per_wave) {
    const int nsub_m = cdiv(mt, HMX_RTZ_MTW), nsub_n = cdiv(ntd, HMX_RTZ_NTW);
    *nsub = nsub_m * nsub_n;
    *slab_per_wave = (*nsub) * HMX_RTZ_MTW * HMX_RTZ_NTW * 256;
}

void launch_rtz(const RtzArgs& a, int wgs, hipStream_t s) {
    int nsub, spw;
    rtz_geometry(a.mt, a.ntd, &nsub, &spw);
    hipLaunchKernelGGL((k_rtz<HMX_RTZ_MTW, HMX_RTZ_NTW>), dim3(wgs, nsub), dim3(256), 0, s, a);
}

void launch_rtz_reduce(const float* slab, int nwaves, int mt, int ntd, int K16, int ld, double* out,
                       const int* task_grp, hipStream_t s) {
    int nsub, spw;
    rtz_geometry(mt, ntd, &nsub, &spw);
    const int seg_len = 32;   // slabs summed by one thread before its fp64 atomic
    hipLaunchKernelGGL((k_rtz_reduce<HMX_RTZ_MTW, HMX_RTZ_NTW>), dim3(cdiv(spw, 256), cdiv(nwaves, seg_len)), dim3(256), 0, s,
                       slab, nwaves, nsub, ntd, K16, ld, out, task_grp, seg_len);
}

void launch_block_table(const TableArgs& a, int K16, hipStream_t s) {
    const size_t sm = (size_t)a.B * 8 + 128 * 8 + (size_t)a.B * 4 + 16;
    hipLaunchKernelGGL(k_block_table, dim3(K16), dim3(128), sm, s, a);
}

void launch_group_sums(const float* R, int Kp, int K, int K16, const int* cells, const int* tile_grp, int n_tiles,
                       double* Ogrp, hipStream_t s) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(k_group_sums, dim3(n_tiles), dim3(128), 0, s, R, Kp, K, K16, cells, tile_grp, n_tiles, Ogrp);
}

void launch_ridge_solve(const RidgeSolveArgs& a, hipStream_t s) {
    if (a.V == 1) hipLaunchKernelGGL(k_ridge_solve_v1, dim3(a.K16), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_ridge_solve_general, dim3(a.K16), dim3(256), 0, s, a);
}

void launch_order(const OrderArgs& a, hipStream_t s) {
    const int nchunks = cdiv(a.N, ORDER_CHUNK);
    const size_t sm = (size_t)a.nblk * a.G * sizeof(int);
    hipLaunchKernelGGL(k_order_pass<0>, dim3(nchunks), dim3(64), sm, s, a);
    hipLaunchKernelGGL(k_order_scan, dim3(a.nblk * a.G), dim3(256), 0, s, a, nchunks);
    hipLaunchKernelGGL(k_order_runs, dim3(a.nblk * a.G), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_order_pass<1>, dim3(nchunks), dim3(64), sm, s, a);
}

int order_chunks(int64_t N) { return cdiv(N, ORDER_CHUNK); }

void launch_gather_rows(const float* src, int ld, int cols, const int* rows, int n_rows, float* dst, hipStream_t s) {
    if (n_rows <= 0) return;
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv((int64_t)n_rows * cols, 256)), dim3(256), 0, s, src, ld, cols, rows, n_rows, dst);
}

template <int MTD>
static void launch_apply2_k(const ApplyArgs& a, int kb, size_t sm, hipStream_t s) {
    switch (kb) {
        case 1: hipLaunchKernelGGL((k_ridge_apply2<MTD, 1>), dim3(a.ntasks), dim3(256), sm, s, a); break;
        case 2: hipLaunchKernelGGL((k_ridge_apply2<MTD, 2>), dim3(a.ntasks), dim3(256), sm, s, a); break;
        case 3: hipLaunchKernelGGL((k_ridge_apply2<MTD, 3>), dim3(a.ntasks), dim3(256), sm, s, a); break;
        case 4: hipLaunchKernelGGL((k_ridge_apply2<MTD, 4>), dim3(a.ntasks), dim3(256), sm, s, a); break;
        case 5: hipLaunchKernelGGL((k_ridge_apply2<MTD, 5>), dim3(a.ntasks), dim3(256), sm, s, a); break;
        case 6: hipLaunchKernelGGL((k_ridge_apply2<MTD, 6>), dim3(a.ntasks), dim3(256), sm, s, a); break;
        default: hipLaunchKernelGGL((k_ridge_apply2<MTD, 7>), dim3(a.ntasks), dim3(256), sm, s, a); break;
    }
}

void launch_load_rows(const float* src, int d, const int* source_row, float* dst, int dp, int64_t N, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(k_load_rows, dim3(cdiv(N * dp, 256)), dim3(256), 0, s, src, d, source_row, dst, dp, N);
}

void launch_kmeans_seed(const SeedArgs& a, int K, hipStream_t s) {
    const int wgs = cdiv(a.n, 256);
    hipLaunchKernelGGL(k_seed_transpose, dim3(cdiv((int64_t)a.n * a.d, 256)), dim3(256), 0, s, a.X, a.n, a.d, a.Xt);
    const size_t sm = (size_t)SEED_TRIALS * a.d * sizeof(float);
    for (int step = 0; step < K; ++step) {
        hipLaunchKernelGGL(k_seed_pick, dim3(1), dim3(256), 0, s, a, step);
        hipLaunchKernelGGL(k_seed_eval, dim3(wgs), dim3(256), sm, s, a, step);
        hipLaunchKernelGGL(k_seed_commit, dim3(wgs), dim3(256), 0, s, a, step);
    }
}

size_t y_planes_dwords(int K16, int dp) { return (size_t)((dp + 31) / 32) * 3 * (K16 / 16) * 256; }
void launch_y_planes(const float* Y, int K16, int ldy, int dp, unsigned* Yf, hipStream_t s) {
    const int ns = (dp + 31) / 32, mt = K16 / 16;
    hipLaunchKernelGGL(k_y_planes, dim3(ns * mt), dim3(64), 0, s, Y, ldy, mt, Yf);
}
size_t w_planes_dwords(int G, int K16, int dp) { return (size_t)G * ((K16 + 31) / 32) * 3 * (dp / 16) * 256; }
void launch_w_planes(const float* W, int G, int K16, int ldw, int dp, unsigned* Wf, hipStream_t s) {
    const int ns = (K16 + 31) / 32, mtd = dp / 16;
    hipLaunchKernelGGL(k_w_planes, dim3(ns * mtd, G), dim3(64), 0, s, W, K16, ldw, mtd, ns, Wf);
}

int launch_ridge_apply(const ApplyArgs& a_in, int max_wgs, hipStream_t s) {
    ApplyArgs a = a_in;
    if (a.n_tiles <= 0) return 0;
    if (a.task_tile0 && a.Wf && rtz_wide_ok((a.K16 + 15) / 16, a.dp)) {   // the bf16-pipe instance: W comes as fragments (launch_w_planes)
        const int mtd = a.dp / 16;
        const size_t sm = (size_t)2 * 3 * mtd * 1024;
#define HMX_AWB(M)                                                                                                     \
    case M: {                                                                                                         \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ridge_apply_wideb<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((k_ridge_apply_wideb<M>), dim3(a.ntasks), dim3(64 * APPLYB_WAVES), sm, s, a);               \
    } break;
        if (mtd < 1 || mtd > 13 || sm > (size_t)80 * 1024) return -1;   // (13 column tiles: 78 KB, the attribute's 80 KB)
        switch (mtd) {
            HMX_AWB(1) HMX_AWB(2) HMX_AWB(3) HMX_AWB(4) HMX_AWB(5) HMX_AWB(6) HMX_AWB(7) HMX_AWB(8) HMX_AWB(9) HMX_AWB(10) HMX_AWB(11) HMX_AWB(12)
            HMX_AWB(13)   // (configs[4]: d = 200 -> 13 tiles, 78 KB of dynamic LDS: above the 64 KB default, so the attribute is set like everywhere else)
        }
#undef HMX_AWB
        return 0;
    }
    if (a.task_tile0 && rtz_wide_ok((a.K16 + 15) / 16, a.dp)) {
        const int mtd = a.dp / 16;
        a.ldw_lds = ((16 * mtd + 31) / 32) * 32 + 16;
        const size_t sm = (size_t)2 * 16 * a.ldw_lds * sizeof(float);
#define HMX_AW(M) case M: hipLaunchKernelGGL((k_ridge_apply_wide<M>), dim3(a.ntasks), dim3(64 * WIDE_WAVES), sm, s, a); break;
        switch (mtd) {
            HMX_AW(1) HMX_AW(2) HMX_AW(3) HMX_AW(4) HMX_AW(5) HMX_AW(6) HMX_AW(7) HMX_AW(8) HMX_AW(9) HMX_AW(10) HMX_AW(11) HMX_AW(12)
            default: hipLaunchKernelGGL((k_ridge_apply_wide<13>), dim3(a.ntasks), dim3(64 * WIDE_WAVES), sm, s, a); break;
        }
#undef HMX_AW
        return 0;
    }
    if (a.task_tile0 && rtz2_ok((a.K16 + 15) / 16, a.dp)) {
        const int mtd = a.dp == 32 ? 2 : 4;
        a.ldw_lds = ((16 * mtd + 31) / 32) * 32 + 16;
        const size_t sm = (size_t)a.K16 * a.ldw_lds * sizeof(float);
        if (mtd == 2) launch_apply2_k<2>(a, a.K16 / 16, sm, s);
        else launch_apply2_k<4>(a, a.K16 / 16, sm, s);
        return 0;
    }
    constexpr int NT = 2;
    const int wgs = assign_grid(a.n_tiles, NT, max_wgs);
    if (a.mtd <= 4) hipLaunchKernelGGL((k_ridge_apply<4, NT>), dim3(wgs), dim3(256), 0, s, a);
    else if (a.mtd <= 13) hipLaunchKernelGGL((k_ridge_apply<13, NT>), dim3(wgs), dim3(256), 0, s, a);
    else if (a.mtd <= 20) hipLaunchKernelGGL((k_ridge_apply<20, 1>), dim3(assign_grid(a.n_tiles, 1, max_wgs)), dim3(256), 0, s, a);   // up to 320 PCs
    else return -1;
    return 0;
}
