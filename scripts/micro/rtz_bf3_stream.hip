// Micro-benchmark, second step: the STREAMING form of k_rtz3's product on the bf16 matrix pipe.
// R (N x 100 floats) and Z (N x 52 floats) in storage order, one block id per cell; a wave owns PAIRS of 16-cell tiles
// (one k-step of 32 for v_mfma_f32_16x16x32_bf16), brings them global -> LDS with 1 KB LDS-DMA requests into two private
// pair buffers (one pair travels while the other is multiplied), splits both operands in registers into three exact bf16
// terms and multiplies with 7 x (4 x 6 + 3) MFMAs per pair; one workgroup of four waves per CU (its 4 x 2 buffers take 156 KB).
// Output: per-wave accumulator slabs, summed on the host in float64 and compared with a float64 reference for a small N;
// then the time of a 1 M-cell pass (HIP events) -- to set beside k_rtz3's 168-180 us at C3 (f32-input MFMA, 0.65 of its
// pipe) and the stream's own bound (657 MB at 6.2-6.9 TB/s = 95-106 us).
//   hipcc --offload-arch=gfx950 -O3 -I harmonypy_amd/csrc scripts/micro/rtz_bf3_stream.hip -o build/micro/rtz_bf3_stream
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hmx_device.h"

constexpr int MT = 7, KP = 100, KS = 13, DP = 52, NT = 5, H = MT / 4, REM = MT % 4;
constexpr int R_BYTES = 32 * KP * 4, Z_BYTES = 32 * DP * 4;          // a pair's rows: 12800 + 6656 bytes, contiguous in memory
constexpr int BUF_BYTES = R_BYTES + Z_BYTES + 32;                    // | 32 block ids
constexpr int NR = (R_BYTES + 1023) / 1024, NZ = (Z_BYTES + 1023) / 1024, NI = NR + NZ + 1;   // requests per pair: 13 + 7 + 1
constexpr int WAVES = 4;

__device__ __forceinline__ void dma16(const void* base_, unsigned voff, unsigned zone) {   // as in csrc/hmx_rtz3.hip
    const unsigned long long bits = (unsigned long long)base_;
    const void* base = (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bits >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bits));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

__global__ __launch_bounds__(64 * WAVES, 1) void k_stream(const float* __restrict__ Rg, const float* __restrict__ Zg,
                                                          const unsigned char* __restrict__ idg, int npairs, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = blockIdx.x * WAVES + wv, nwaves = gridDim.x * WAVES;
    const unsigned lane16 = 16 * lane;
    unsigned char* mybuf = smem + (size_t)wv * 2 * BUF_BYTES;
    const unsigned zone0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)mybuf);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_mine = wave < npairs ? (npairs - wave + nwaves - 1) / nwaves : 0;    // pairs wave, wave + nwaves, ...
    // the NI requests of this wave's i-th pair in two parts: its Z rows + block ids (that part of a buffer is free as soon as
    // the B operands are split), its R rows (free after the last cluster tile)
    auto request_z = [&](int i, int par) {
        const size_t p = (size_t)wave + (size_t)i * nwaves;
        const unsigned char* z = reinterpret_cast<const unsigned char*>(Zg) + p * Z_BYTES;
        const unsigned zb = zone0 + (par ? (unsigned)BUF_BYTES : 0u);
#pragma unroll
        for (int it = 0; it < NZ; ++it)
            if (1024 * (it + 1) <= Z_BYTES || 1024 * it + (int)lane16 < Z_BYTES) dma16(z + 1024 * it, lane16, zb + (unsigned)R_BYTES + 1024u * it);
        if (lane < 2) dma16(idg + p * 32, lane16, zb + (unsigned)(R_BYTES + Z_BYTES));
    };
    auto request_r = [&](int i, int par) {
        const size_t p = (size_t)wave + (size_t)i * nwaves;
        const unsigned char* r = reinterpret_cast<const unsigned char*>(Rg) + p * R_BYTES;
        const unsigned zb = zone0 + (par ? (unsigned)BUF_BYTES : 0u);
#pragma unroll
        for (int it = 0; it < NR; ++it)
            if (1024 * (it + 1) <= R_BYTES || 1024 * it + (int)lane16 < R_BYTES) dma16(r + 1024 * it, lane16, zb + 1024u * it);
    };
    if (n_mine > 0) { request_z(0, 0); request_r(0, 0); }
    if (n_mine > 1) { request_z(1, 1); request_r(1, 1); }
    for (int i = 0; i < n_mine; ++i) {
        asm volatile("" ::: "memory");
        if (i + 1 < n_mine) wait_vmcnt<NI>(); else wait_vmcnt<0>();                  // pair i has landed (pair i+1 may travel on)
        asm volatile("" ::: "memory");
        const unsigned char* buf = mybuf + (size_t)(i & 1) * BUF_BYTES;
        const float* R = reinterpret_cast<const float*>(buf);
        const float* Z = reinterpret_cast<const float*>(buf + R_BYTES);
        const unsigned char* ids = buf + R_BYTES + Z_BYTES;
        // ---- the pair's product (scripts/micro/rtz_bf3.hip, MODE 1): k slot j of lane (c16, q) <-> cell 8 q + j ------------
        u32x4 bh[4], bm[4], bl[4], oh;
        {
            int bid[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bid[j] = ids[8 * q + j];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x2 z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = *reinterpret_cast<const f32x2*>(Z + (8 * q + j) * DP + 4 * min(c16, KS - 1) + 2 * half);
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // Z rows and ids are in registers
        if (i + 2 < n_mine) request_z(i + 2, i & 1);
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
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x2*>(R + (8 * q + j) * KP + 4 * c16 + 2 * half);
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
                a[j] = R[(8 * q + j) * KP + min(col, KP - 1)];
                if (col >= KP) a[j] = 0.f;
            }
            one_mt(4 * H + jj, a);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the buffer is in registers: hand it to the pair after next ------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 2 < n_mine) request_r(i + 2, i & 1);
    }
    float* o = out + (size_t)wave * (MT * NT * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(o + (mt * NT + nt) * 256 + 4 * lane) = acc[mt][nt];
}

int main() {
    const int wgs = 256, nwaves = wgs * WAVES;
    const size_t n_big = 1 << 20, n_small = 1 << 16;                   // cells (multiples of 32)
    std::vector<float> R(n_big * KP), Z(n_big * DP);
    std::vector<unsigned char> ids(n_big);
    srand(7);
    for (size_t c = 0; c < n_big; ++c) {
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
    unsigned char* dI;
    hipMalloc(&dR, R.size() * 4 + 4096); hipMalloc(&dZ, Z.size() * 4 + 4096); hipMalloc(&dI, ids.size() + 4096);
    hipMalloc(&dout, (size_t)nwaves * MT * NT * 256 * 4);
    hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dZ, Z.data(), Z.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dI, ids.data(), ids.size(), hipMemcpyHostToDevice);
    const size_t lds = (size_t)WAVES * 2 * BUF_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto run = [&](size_t n) { hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(64 * WAVES), lds, 0, dR, dZ, dI, (int)(n / 32), dout); };
    // ---- numerics on the small pass: slabs summed in float64 vs a float64 reference ----------------------------------------------
    run(n_small);
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
        for (int col = 0; col < 72; ++col) {                               // 52 PC columns (two of them padding) + 20 blocks
            const double e = std::fabs(got[k * 80 + col] - ref[k * 80 + col]);
            worst = std::max(worst, e); scale = std::max(scale, std::fabs(ref[k * 80 + col]));
            if (e > 1e-4 * (std::fabs(ref[k * 80 + col]) + 1.0)) ++bad;
        }
    printf("numerics, %zu cells: max |err| %.3e of %.3e, %d entries off by more than 1e-4\n", n_small, worst, scale, bad);
    // ---- the 1 M-cell pass ------------------------------------------------------------------------------------------------------
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    for (int pass = 0; pass < 6; ++pass) {
        hipEventRecord(e0);
        run(n_big);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (pass) { best = std::min(best, ms); sum += ms; }
    }
    const double bytes = (double)n_big * (KP * 4 + DP * 4 + 1);
    printf("stream, %zu cells: %.1f us per pass (best %.1f), %.0f MB -> %.2f TB/s; requests per pair %d, LDS %zu bytes per workgroup\n", n_big,
           sum / 5 * 1e3, best * 1e3, bytes / 1e6, bytes / (sum / 5 * 1e-3) / 1e12, NI, lds);
    return bad != 0;
}
