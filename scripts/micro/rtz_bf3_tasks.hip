// Micro-benchmark, third step: rtz_bf3_stream.hip with k_rtz3's TASK structure, ready to move into
// csrc/hmx_rtz3.hip -- cells in group-sorted storage order with ragged group ends, static 16-cell tiles with 16 block-id bytes
// each (255 = padding), tasks = (first tile, end tile, first cell, group's end cell, tile stride), a wave takes the tiles
// t0 + w + stride i.  The unit of the bf16 multiply is a PAIR of the wave's tiles (any two: they need not be neighbours in
// memory), each in its own LDS buffer; four tile buffers per wave = two pairs, one travelling while the other is multiplied.
// A tile's rows past its group's end are zeroed in LDS (as k_rtz3 does); a missing second tile contributes zeros.
// -DADJACENT=1: a wave's unit is a pair of NEIGHBOURING tiles (t, t + 1) = 32 consecutive cells: one contiguous 19.5 KB piece,
// 21 requests instead of 24, two pair buffers per wave; the group's last pair may be half a pair (rows past the end zeroed).
//   hipcc --offload-arch=gfx950 -O3 [-DADJACENT=1] -I harmonypy_amd/csrc scripts/micro/rtz_bf3_tasks.hip -o build/micro/rtz_bf3_tasks
#ifndef ADJACENT
#define ADJACENT 0
#endif
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hmx_device.h"

constexpr int MT = 7, KP = 100, KS = 13, DP = 52, NT = 5, H = MT / 4, REM = MT % 4;
constexpr int UROWS = ADJACENT ? 32 : 16;                            // rows of a streaming unit (a tile, or a pair of neighbours)
constexpr int R_BYTES = UROWS * KP * 4, Z_BYTES = UROWS * DP * 4;    // a unit's rows: contiguous in memory
constexpr int BUF_BYTES = R_BYTES + Z_BYTES + UROWS;                 // | its block ids
constexpr int NR = (R_BYTES + 1023) / 1024, NZ = (Z_BYTES + 1023) / 1024, NI = NR + NZ + 1;   // requests per tile: 7 + 4 + 1
constexpr int WAVES = 4;

__device__ __forceinline__ void dma16(const void* base_, unsigned voff, unsigned zone) {   // as in csrc/hmx_rtz3.hip
    const unsigned long long bits = (unsigned long long)base_;
    const void* base = (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bits >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bits));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

struct Tasks { const int *t0, *t1, *c0, *cend, *stride; };
__global__ __launch_bounds__(64 * WAVES, 1) void k_tasks(const float* __restrict__ Rg, const float* __restrict__ Zg,
                                                         const unsigned char* __restrict__ tile_blk, Tasks T, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = blockIdx.x;
    const unsigned lane16 = 16 * lane;
    unsigned char* mybuf = smem + (size_t)wv * (ADJACENT ? 2 : 4) * BUF_BYTES;        // pair (i & 1) [, tile u of the pair]
    const unsigned zone0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)mybuf);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int t0 = T.t0[task], t1 = T.t1[task], c_first = T.c0[task], c_end = T.cend[task];
    const int stride = __builtin_amdgcn_readfirstlane(T.stride[task]);
#if ADJACENT
    // this wave's pairs: tiles t0 + 2 (wv + stride i), + 1
    const int n_mine = (t1 - t0 - 2 * wv + 2 * stride - 1) / (2 * stride);
    const int c_mine = c_first + 32 * wv;
    auto request_pair = [&](int i, int par) {
        const size_t cell0 = (size_t)c_mine + (size_t)32 * stride * i;
        const unsigned char* r = reinterpret_cast<const unsigned char*>(Rg) + cell0 * (KP * 4);
        const unsigned char* z = reinterpret_cast<const unsigned char*>(Zg) + cell0 * (DP * 4);
        const unsigned char* id = tile_blk + (size_t)16 * (t0 + 2 * (wv + (size_t)stride * i));
        const unsigned zb = zone0 + (unsigned)par * (unsigned)BUF_BYTES;
#pragma unroll
        for (int it = 0; it < NR; ++it)
            if (1024 * (it + 1) <= R_BYTES || 1024 * it + (int)lane16 < R_BYTES) dma16(r + 1024 * it, lane16, zb + 1024u * it);
#pragma unroll
        for (int it = 0; it < NZ; ++it)
            if (1024 * (it + 1) <= Z_BYTES || 1024 * it + (int)lane16 < Z_BYTES) dma16(z + 1024 * it, lane16, zb + (unsigned)R_BYTES + 1024u * it);
        if (lane < 2) dma16(id, lane16, zb + (unsigned)(R_BYTES + Z_BYTES));
    };
    constexpr int NPAIR = NI;
#else
    const int n_tiles = (t1 - t0 - wv + stride - 1) / stride;                         // this wave's tiles: t0 + wv + stride i
    const int n_mine = (n_tiles + 1) / 2;                                             // ... taken two at a time
    const int c_mine = c_first + 16 * wv;
    // the requests of the wave's tile `ti` into buffer `b` (0..3)
    auto request = [&](int ti, int b) {
        const size_t cell0 = (size_t)c_mine + (size_t)16 * stride * ti;
        const unsigned char* r = reinterpret_cast<const unsigned char*>(Rg) + cell0 * (KP * 4);
        const unsigned char* z = reinterpret_cast<const unsigned char*>(Zg) + cell0 * (DP * 4);
        const unsigned char* id = tile_blk + (size_t)16 * (t0 + wv + (size_t)stride * ti);
        const unsigned zb = zone0 + (unsigned)b * (unsigned)BUF_BYTES;
#pragma unroll
        for (int it = 0; it < NR; ++it)
            if (1024 * (it + 1) <= R_BYTES || 1024 * it + (int)lane16 < R_BYTES) dma16(r + 1024 * it, lane16, zb + 1024u * it);
#pragma unroll
        for (int it = 0; it < NZ; ++it)
            if (1024 * (it + 1) <= Z_BYTES || 1024 * it + (int)lane16 < Z_BYTES) dma16(z + 1024 * it, lane16, zb + (unsigned)R_BYTES + 1024u * it);
        if (lane == 0) dma16(id, lane16, zb + (unsigned)(R_BYTES + Z_BYTES));
    };
    // a pair's requests are always 2 NI instructions: a missing second tile requests the first one again (into the spare
    // buffer; it is never read), so that the counted waits do not depend on the parity of the wave's tile count
    auto request_pair = [&](int i, int par) {
        request(2 * i, 2 * par);
        request(2 * i + 1 < n_tiles ? 2 * i + 1 : 2 * i, 2 * par + 1);
    };
    constexpr int NPAIR = 2 * NI;
#endif
    if (n_mine > 0) request_pair(0, 0);
    if (n_mine > 1) request_pair(1, 1);
    for (int i = 0; i < n_mine; ++i) {
        asm volatile("" ::: "memory");
        if (i + 1 < n_mine) wait_vmcnt<NPAIR>(); else wait_vmcnt<0>();               // pair i has landed (pair i+1 may travel on)
        asm volatile("" ::: "memory");
#if ADJACENT
        unsigned char* pb = mybuf + (size_t)(i & 1) * BUF_BYTES;
        {   // rows past the group's end hold other cells (or the slack behind the arrays): they count for nothing
            const int c0 = c_mine + 32 * stride * i;
            const int n_live = min(32, c_end - c0);
            if (n_live < 32) {
                float* Rt = reinterpret_cast<float*>(pb);
                float* Zt = reinterpret_cast<float*>(pb + R_BYTES);
                for (int j = n_live * KP + lane; j < 32 * KP; j += 64) Rt[j] = 0.f;
                for (int j = n_live * DP + lane; j < 32 * DP; j += 64) Zt[j] = 0.f;
            }
        }
        const float* R = reinterpret_cast<const float*>(pb) + 8 * q * KP;            // lane (c16, q): cells 8 q .. + 8 of the pair
        const float* Z = reinterpret_cast<const float*>(pb + R_BYTES) + 8 * q * DP;
        const unsigned char* ids = pb + R_BYTES + Z_BYTES + 8 * q;
#else
        const bool has2 = 2 * i + 1 < n_tiles;                                       // wave-uniform
        unsigned char* pb = mybuf + (size_t)(2 * (i & 1)) * BUF_BYTES;
        // rows past the group's end hold other cells (or the slack behind the arrays): they count for nothing
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int c0 = c_mine + 16 * stride * (2 * i + u);
            const int n_live = (u == 0 || has2) ? min(16, c_end - c0) : 0;
            if (n_live < 16) {
                float* Rt = reinterpret_cast<float*>(pb + (size_t)u * BUF_BYTES);
                float* Zt = reinterpret_cast<float*>(pb + (size_t)u * BUF_BYTES + R_BYTES);
                for (int j = n_live * KP + lane; j < 16 * KP; j += 64) Rt[j] = 0.f;
                for (int j = n_live * DP + lane; j < 16 * DP; j += 64) Zt[j] = 0.f;
                if (n_live == 0 && lane < 4) reinterpret_cast<unsigned*>(pb + (size_t)u * BUF_BYTES + R_BYTES + Z_BYTES)[lane] = 0xFFFFFFFFu;
            }
        }
        // lane (c16, q): cells 8 (q & 1) .. + 8 of tile q >> 1 of the pair
        const unsigned char* buf = pb + (size_t)(q >> 1) * BUF_BYTES;
        const float* R = reinterpret_cast<const float*>(buf) + 8 * (q & 1) * KP;
        const float* Z = reinterpret_cast<const float*>(buf + R_BYTES) + 8 * (q & 1) * DP;
        const unsigned char* ids = buf + R_BYTES + Z_BYTES + 8 * (q & 1);
#endif
        // ---- the pair's product (scripts/micro/rtz_bf3.hip, MODE 1): k slot j of lane (c16, q) <-> cell 8 q + j ------------
        u32x4 bh[4], bm[4], bl[4], oh;
        {
            int bid[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bid[j] = ids[j];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x2 z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = *reinterpret_cast<const f32x2*>(Z + j * DP + 4 * min(c16, KS - 1) + 2 * half);
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int nt = 2 * half + n2;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        f32x2 x;
                        x.x = (c16 < KS) ? z[2 * p][n2] : ((bid[2 * p] == 4 * c16 + nt - DP) ? 1.f : 0.f);
                        x.y = (c16 < KS) ? z[2 * p + 1][n2] : ((bid[2 * p + 1] == 4 * c16 + nt - DP) ? 1.f : 0.f);
                        unsigned h, m, l;
                        bf16_split3(x, h, m, l);
                        bh[nt][p] = h; bm[nt][p] = m; bl[nt][p] = l;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
                oh[p] = ((bid[2 * p] == (64 - DP) + c16) ? 0x3F80u : 0u) | ((bid[2 * p + 1] == (64 - DP) + c16) ? 0x3F800000u : 0u);
        }
        __builtin_amdgcn_sched_barrier(0);
        auto one_mt = [&](int mt, const float (&a)[8]) {
            u32x4 ah, am, al;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned h, m, l;
                bf16_split3((f32x2){a[2 * p], a[2 * p + 1]}, h, m, l);
                ah[p] = h; am[p] = m; al[p] = l;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[mt][nt] = MFMA_BF16(al, bh[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(ah, bl[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(am, bm[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(am, bh[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(ah, bm[nt], acc[mt][nt]);
                acc[mt][nt] = MFMA_BF16(ah, bh[nt], acc[mt][nt]);
            }
            acc[mt][4] = MFMA_BF16(al, oh, acc[mt][4]);
            acc[mt][4] = MFMA_BF16(am, oh, acc[mt][4]);
            acc[mt][4] = MFMA_BF16(ah, oh, acc[mt][4]);
        };
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x2*>(R + j * KP + 4 * c16 + 2 * half);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                float a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = v[j][jj];
                one_mt(2 * half + jj, a);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int jj = 0; jj < REM; ++jj) {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = 64 * H + REM * c16 + jj;                              // clusters 100..111 do not exist: no read past the row
                a[j] = R[j * KP + min(col, KP - 1)];
                if (col >= KP) a[j] = 0.f;
            }
            one_mt(4 * H + jj, a);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the buffer is in registers: hand it to the pair after next ------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 2 < n_mine) request_pair(i + 2, i & 1);
    }
    float* o = out + ((size_t)task * WAVES + wv) * (MT * NT * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(o + (mt * NT + nt) * 256 + 4 * lane) = acc[mt][nt];
}

int main() {
    const int G = 8, tpg = 32, ntasks = G * tpg, nwaves = ntasks * WAVES;
    const size_t n_big = (1 << 20) - 57, n_small = (1 << 16) + 203;     // ragged: no group is a multiple of 16
    std::vector<float> R((n_big + 64) * KP, 0.f), Z((n_big + 64) * DP, 0.f);
    std::vector<unsigned char> ids(n_big + 64);
    srand(7);
    for (size_t c = 0; c < n_big + 64; ++c) {                            // (the slack behind the arrays holds numbers too)
        double s = 0, n2 = 0;
        float* r = &R[c * KP];
        float* z = &Z[c * DP];
        for (int k = 0; k < KP; ++k) { r[k] = (float)std::exp(6.0 * rand() / RAND_MAX); s += r[k]; }
        for (int k = 0; k < KP; ++k) r[k] = (float)(r[k] / s);
        for (int j = 0; j < 50; ++j) { z[j] = (float)(rand() / (double)RAND_MAX - 0.5); n2 += (double)z[j] * z[j]; }
        for (int j = 0; j < 50; ++j) z[j] = (float)(z[j] / std::sqrt(n2));
        z[50] = z[51] = 0.f;
        ids[c] = (unsigned char)(rand() % 20);
    }
    float *dR, *dZ, *dout;
    unsigned char* dB;
    int* dT;
    hipMalloc(&dR, R.size() * 4); hipMalloc(&dZ, Z.size() * 4);
    hipMalloc(&dB, n_big + 16 * (G + 64)); hipMalloc(&dT, 5 * ntasks * 4);
    hipMalloc(&dout, (size_t)nwaves * MT * NT * 256 * 4);
    hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dZ, Z.data(), Z.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)WAVES * (ADJACENT ? 2 : 4) * BUF_BYTES;
    const int TU = ADJACENT ? 2 : 1;                                     // tiles of a wave's unit
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tasks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // groups of unequal, non-multiple-of-16 sizes; static tiles; the group's tasks sweep its tiles together (k_rtz3's grid-stride form)
    auto setup = [&](size_t n, bool contiguous) {
        std::vector<size_t> gs(G + 1, 0);
        for (int g = 0; g < G; ++g) gs[g + 1] = gs[g] + n / G + (g % 3) * 5 - 5;
        gs[G] = n;
        std::vector<int> tile_start(G + 1, 0);
        for (int g = 0; g < G; ++g) tile_start[g + 1] = tile_start[g] + (int)((gs[g + 1] - gs[g] + 15) / 16);
        std::vector<unsigned char> blk((size_t)16 * tile_start[G], 255);
        for (int g = 0; g < G; ++g)
            for (size_t c = gs[g]; c < gs[g + 1]; ++c) blk[(size_t)16 * tile_start[g] + (c - gs[g])] = ids[c];
        std::vector<int> t(5 * ntasks);
        for (int g = 0; g < G; ++g) {
            const int nt = tile_start[g + 1] - tile_start[g];
            for (int j = 0; j < tpg; ++j) {
                const int w = g * tpg + j;
                if (contiguous) {                                           // a contiguous run of tiles per task, the waves interleaved in it
                    const int q4 = 4 * TU, per = ((nt + tpg - 1) / tpg + q4 - 1) / q4 * q4, a0 = std::min(nt, j * per), a1 = std::min(nt, a0 + per);
                    t[w] = tile_start[g] + a0; t[ntasks + w] = tile_start[g] + a1; t[2 * ntasks + w] = (int)gs[g] + 16 * a0; t[4 * ntasks + w] = WAVES;
                } else {
                    t[w] = tile_start[g] + TU * WAVES * j; t[ntasks + w] = tile_start[g + 1]; t[2 * ntasks + w] = (int)gs[g] + 16 * TU * WAVES * j; t[4 * ntasks + w] = WAVES * tpg;
                }
                t[3 * ntasks + w] = (int)gs[g + 1];
            }
        }
        hipMemcpy(dB, blk.data(), blk.size(), hipMemcpyHostToDevice);
        hipMemcpy(dT, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    };
    const Tasks T{dT, dT + ntasks, dT + 2 * ntasks, dT + 3 * ntasks, dT + 4 * ntasks};
    auto run = [&]() { hipLaunchKernelGGL(k_tasks, dim3(ntasks), dim3(64 * WAVES), lds, 0, dR, dZ, dB, T, dout); };
    int bad_total = 0;
    for (int contiguous = 0; contiguous < 2; ++contiguous) {
        // ---- numerics on the small pass: slabs summed in float64 vs a float64 reference ------------------------------------------
        setup(n_small, contiguous);
        run();
        hipDeviceSynchronize();
        std::vector<float> o((size_t)nwaves * MT * NT * 256);
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        std::vector<double> ref(112 * 80, 0.0), got(112 * 80, 0.0);
        for (size_t c = 0; c < n_small; ++c)
            for (int k = 0; k < KP; ++k) {
                const double r = R[c * KP + k];
                for (int j = 0; j < 52; ++j) ref[k * 80 + j] += r * Z[c * DP + j];
                ref[k * 80 + 52 + ids[c]] += r;
            }
        for (int w = 0; w < nwaves; ++w)
            for (int mt = 0; mt < MT; ++mt)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 4; ++i) {
                            const int c16 = lane & 15, q = lane >> 4, m = 4 * q + i;
                            const int cluster = mt < 4 * H ? 64 * (mt / 4) + 4 * m + (mt % 4) : 64 * H + REM * m + (mt - 4 * H);
                            const int col = nt < 4 ? 4 * c16 + nt : 64 + c16;
                            got[cluster * 80 + col] += o[((size_t)w * MT * NT + mt * NT + nt) * 256 + 4 * lane + i];
                        }
        double worst = 0, scale = 0;
        int bad = 0;
        for (int k = 0; k < KP; ++k)
            for (int col = 0; col < 72; ++col) {
                const double e = std::fabs(got[k * 80 + col] - ref[k * 80 + col]);
                worst = std::max(worst, e); scale = std::max(scale, std::fabs(ref[k * 80 + col]));
                if (!(e <= 1e-4 * (std::fabs(ref[k * 80 + col]) + 1.0))) ++bad;
            }
        printf("%s tasks, numerics, %zu cells in %d ragged groups: max |err| %.3e of %.3e, %d entries off by more than 1e-4\n",
               contiguous ? "contiguous " : "grid-stride", n_small, G, worst, scale, bad);
        bad_total += bad;
        // ---- the 1 M-cell pass ------------------------------------------------------------------------------------------------------
        setup(n_big, contiguous);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f, sum = 0.f;
        for (int pass = 0; pass < 6; ++pass) {
            hipEventRecord(e0);
            run();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (pass) { best = std::min(best, ms); sum += ms; }
        }
        const double bytes = (double)n_big * (KP * 4 + DP * 4 + 1);
        printf("%s tasks, stream, %zu cells: %.1f us per pass (best %.1f), %.0f MB -> %.2f TB/s; %d tasks, requests per unit of %d rows %d, LDS %zu bytes per workgroup\n",
               contiguous ? "contiguous " : "grid-stride", n_big, sum / 5 * 1e3, best * 1e3, bytes / 1e6, bytes / (sum / 5 * 1e-3) / 1e12, ntasks, UROWS, NI, lds);
    }
    return bad_total != 0;
}
