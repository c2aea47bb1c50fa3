// Micro-benchmark + numerics check for the NEXT step of DESIGN.md section 8 (1b): k_rtz3's product R^T . [Z | one-hot of the
// update block] of 16-cell tiles -- 7 cluster tiles x (4 PC-column tiles + 1 one-hot tile), the k index runs over cells --
//   MODE 0: as k_rtz3 does it, v_mfma_f32_16x16x4_f32: 4 k-steps x 35 MFMAs of 32 cycles per tile;
//   MODE 1: on the bf16 matrix pipe, BOTH operands split in registers into three exact bf16 terms (hmx_device.h: bf16_split3):
//           one k-step of 32 = a PAIR of tiles, 7 x (4 x 6 + 3) = 189 MFMAs of 16 cycles (a one-hot column is exact in bf16:
//           three products).
// Operands come from LDS with k_rtz3's index maps (cluster 4 c16 + j / 64 + 3 c16 + j per A row, column 4 c16 + nt per B
// column); one pair of tiles is resident per workgroup and re-read every repetition: the arithmetic side alone, two waves
// per SIMD as in k_rtz3.  Prints shader cycles per pair of tiles and wave, the time per "1 M cells" the rate amounts to, and
// the largest deviation of one wave's result from a float64 reference.
//   hipcc --offload-arch=gfx950 -O3 -I harmonypy_amd/csrc scripts/micro/rtz_bf3.hip -o build/micro/rtz_bf3 && build/micro/rtz_bf3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hmx_device.h"

constexpr int MT = 7, KP = 112, KS = 13, DP = 52, NT = 5, H = MT / 4, REM = MT % 4;
constexpr int PAIR_FLOATS = 32 * KP + 32 * DP + 8;   // R rows | Z rows | 32 block ids (bytes)

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ pairs, float* __restrict__ out, unsigned long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[PAIR_FLOATS];
    extern __shared__ float pad[];   // (dynamic LDS only to hold the occupancy down: 90 KB -> one workgroup = one wave per SIMD)
    (void)pad;
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < PAIR_FLOATS; i += 256) lds[i] = pairs[i];
    __syncthreads();
    const float* Rt = lds;
    const float* Zt = lds + 32 * KP;
    const unsigned char* ids = reinterpret_cast<const unsigned char*>(lds + 32 * KP + 32 * DP);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        int zero = 0;
        asm volatile("" : "+v"(zero));   // nothing of the operand reads is hoisted out of the repetition loop
        const float* R = Rt + zero;
        const float* Z = Zt + zero;
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int cell = 16 * t + 4 * q + ks;
                    float a[MT], b[NT];
                    const f32x4 v = ld4(R + cell * KP + 4 * c16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = v[j];
#pragma unroll
                    for (int j = 0; j < REM; ++j) a[4 * H + j] = R[cell * KP + 64 * H + REM * c16 + j];
                    const f32x4 z = ld4(Z + cell * DP + 4 * min(c16, KS - 1));
                    const int bid = ids[cell + zero];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) b[nt] = (c16 < KS) ? z[nt] : ((bid == 4 * c16 + nt - DP) ? 1.f : 0.f);
                    b[4] = (bid == (64 - DP) + c16) ? 1.f : 0.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(a[mt], b[nt], acc[mt][nt]);
                }
        } else {
            // k slot j of lane (c16, q) <-> cell 8 q + j of the pair
            u32x4 bh[4], bm[4], bl[4], oh;   // B planes of the four PC-column tiles, the one-hot tile's only plane
            {
                int bid[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bid[j] = ids[8 * q + j + zero];
#pragma unroll
                for (int half = 0; half < 2; ++half) {   // two column tiles at a time: 8-byte reads, 16 raw registers live
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
                for (int p = 0; p < 4; ++p)   // bf16(1.0) = 0x3F80
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
            for (int half = 0; half < 2; ++half) {   // the four cluster tiles of k_rtz3's 16-byte reads, two at a time (8-byte reads:
                f32x2 v[8];                          // eight 16-byte fragments live at once cost 13 spilled registers)
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
                for (int j = 0; j < 8; ++j) a[j] = R[(8 * q + j) * KP + 64 * H + REM * c16 + jj];
                one_mt(4 * H + jj, a);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float* o = out + ((size_t)blockIdx.x * 4 + (tid >> 6)) * (MT * NT * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(o + (mt * NT + nt) * 256 + 4 * lane) = acc[mt][nt];
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const int reps = 200, wgs = 512;
    std::vector<float> pair(PAIR_FLOATS, 0.f);
    unsigned char* ids = reinterpret_cast<unsigned char*>(pair.data() + 32 * KP + 32 * DP);
    srand(7);
    for (int c = 0; c < 32; ++c) {
        double s = 0, n2 = 0;
        for (int k = 0; k < 100; ++k) { const double e = std::exp(6.0 * rand() / RAND_MAX); pair[c * KP + k] = (float)e; s += e; }
        for (int k = 0; k < 100; ++k) pair[c * KP + k] = (float)(pair[c * KP + k] / s);                 // a row of R: sums to 1
        for (int j = 0; j < 50; ++j) { const double v = rand() / (double)RAND_MAX - 0.5; pair[32 * KP + c * DP + j] = (float)v; n2 += v * v; }
        for (int j = 0; j < 50; ++j) pair[32 * KP + c * DP + j] = (float)(pair[32 * KP + c * DP + j] / std::sqrt(n2));   // a unit row of Z
        ids[c] = (unsigned char)(rand() % 20);
    }
    float *dp, *dout;
    unsigned long long* dc;
    hipMalloc(&dp, PAIR_FLOATS * 4);
    hipMalloc(&dout, (size_t)wgs * 4 * MT * NT * 256 * 4);
    hipMalloc(&dc, 8);
    hipMemcpy(dp, pair.data(), PAIR_FLOATS * 4, hipMemcpyHostToDevice);
    // float64 reference of one repetition: out[cluster][column], columns = 52 PCs then one-hot of blocks 0..27
    std::vector<double> ref(112 * 80, 0.0);
    for (int c = 0; c < 32; ++c)
        for (int k = 0; k < 112; ++k) {
            const double r = pair[c * KP + k];
            for (int j = 0; j < 52; ++j) ref[k * 80 + j] += r * pair[32 * KP + c * DP + j];
            ref[k * 80 + 52 + ids[c]] += r;
        }
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int occ = 2; occ >= 1; --occ)
    for (int mode = 0; mode < 2; ++mode) {
        const size_t dyn = occ == 2 ? 0 : 90 * 1024;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), dyn, 0, dp, dout, dc, reps);
            else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), dyn, 0, dp, dout, dc, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc;
        hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        std::vector<float> o(MT * NT * 256);
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int mt = 0; mt < MT; ++mt)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        const int c16 = lane & 15, q = lane >> 4, m = 4 * q + i;
                        const int cluster = mt < 4 * H ? 64 * (mt / 4) + 4 * m + (mt % 4) : 64 * H + REM * m + (mt - 4 * H);
                        const int col = nt < 4 ? 4 * c16 + nt : 64 + c16;           // 0..51 PCs, 52..63 blocks 0..11, 64.. blocks 12..27
                        const double want = reps * ref[cluster * 80 + col];
                        const double got = o[(mt * NT + nt) * 256 + 4 * lane + i];
                        worst = std::max(worst, std::fabs(got - want));
                        scale = std::max(scale, std::fabs(want));
                    }
        // the whole chip: wgs x 4 waves, each `reps` pairs of tiles = 32 cells
        const double cells = (double)wgs * 4 * reps * 32;
        printf("%d wave(s) per SIMD, %s: %.0f shader cycles per pair of tiles and wave (wave 0), %.3f ms for %.1f M cells -> %.1f us per 1 M cells; max |err| %.3e of %.3e (%.1e relative) after %d accumulated repetitions\n",
               occ, mode == 0 ? "f32-input MFMA   " : "bf16 pipe, x3    ", (double)cyc / reps, ms, cells / 1e6, ms * 1e3 / (cells / 1e6), worst, scale, worst / scale, reps);
    }
    return 0;
}
