// Micro-benchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 on MI355X, whole chip busy.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o build/mfma_rate && build/mfma_rate
// For each configuration: shader cycles per MFMA (s_memtime of one wave), nanoseconds per MFMA and SIMD (wall clock over the
// launch), and the shader clock that follows (cycles / time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NACC, int VALU>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, v = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = MFMA16(a, b, acc[i]);
            if (VALU && i % (NACC / VALU) == 0) v = v * 1.0001f + b;
        }
        a += 1e-6f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// Closer to k_rtz3's stream: 7 x 5 output tiles, A operand per row tile, B operand per column tile.
//   MODE 0: operands fixed in registers.  MODE 1: + operands re-read from LDS every k-step (ds_read_b128 x3 + b32 x3).
//   MODE 2: + B operands built with compare/select.  MODE 3: MODE 0 + an exec-masked, branch-guarded asm block every 11 MFMAs.
template <int MODE>
__global__ __launch_bounds__(512, 1) void k2(float* out, unsigned long long* cyc, int iters, int flag) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* my = lds + wv * 64 * 16;
    for (int i = 0; i < 16; ++i) my[i * 64 + lane] = 1e-3f * (lane + i);
    __syncthreads();
    f32x4 acc[7][5];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[7], b[5];
    for (int i = 0; i < 7; ++i) a[i] = my[i * 64 + lane];
    for (int j = 0; j < 5; ++j) b[j] = my[(7 + j) * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 2) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(my + 4 * lane), v1 = *reinterpret_cast<const f32x4*>(my + 256 + 4 * lane);
            const f32x4 z = *reinterpret_cast<const f32x4*>(my + 512 + 4 * lane);
            a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = v0[3]; a[4] = my[768 + lane]; a[5] = my[832 + lane]; a[6] = my[896 + lane];
            (void)v1;
            if (MODE == 2) {
                const int bid = (it >> 2) & 255;
                for (int j = 0; j < 4; ++j) b[j] = (lane & 15) < 13 ? z[j] : (bid == 4 * (lane & 15) + j - 52 ? 1.f : 0.f);
                b[4] = bid == 12 + (lane & 15) ? 1.f : 0.f;
            } else { b[0] = z[0]; b[1] = z[1]; b[2] = z[2]; b[3] = z[3]; b[4] = v1[0]; }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[i][j] = MFMA16(a[i], b[j], acc[i][j]);
                if (MODE == 3 && (i * 5 + j) % 11 == 10) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (flag) { if (lane < 16) asm volatile("s_nop 0\n\ts_nop 0" ::: "memory"); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// k_rtzw's k-step: 13 A operands (3 x ds_read_b128 + 1 x ds_read_b32, masked by a factor at use time), 2 B operands
// (ds_read_b32, fma), 2 x 13 MFMAs on two accumulator sets; the next k-step's reads are issued before the MFMAs.
// BAR: a workgroup barrier every 4 k-steps (k_rtzw's tile boundary).
template <bool BAR>
__global__ __launch_bounds__(512, 1) void k3(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 24];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* my = lds + wv * 64 * 24;
    for (int i = 0; i < 24; ++i) my[i * 64 + lane] = 1e-3f * (lane + i);
    __syncthreads();
    f32x4 acc0[13], acc1[13];
    for (int i = 0; i < 13; ++i) { acc0[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float raw[2][13], zr[2][2];
    auto rd = [&](int set, int it) {
        const float* p = my + ((it & 3) * 64);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + 4 * (lane & 15)), v1 = *reinterpret_cast<const f32x4*>(p + 256 + 4 * (lane & 15)),
                    v2 = *reinterpret_cast<const f32x4*>(p + 512 + 4 * (lane & 15));
        for (int j = 0; j < 4; ++j) { raw[set][j] = v0[j]; raw[set][4 + j] = v1[j]; raw[set][8 + j] = v2[j]; }
        raw[set][12] = p[768 + lane];
        zr[set][0] = p[832 + lane]; zr[set][1] = p[896 + lane];
    };
    rd(0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it0 = 0; it0 < iters; it0 += 4) {
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const int it = it0 + u4;
        const int set = u4 & 1;
        float am[13];
        const float lm = (lane + it) & 1024 ? 0.f : 1.f;
        for (int i = 0; i < 13; ++i) am[i] = raw[set][i] * lm;
        const float b0 = fmaf(zr[set][0], lm, 0.f), b1 = fmaf(zr[set][1], lm, 1.f);
        __builtin_amdgcn_sched_barrier(0);
        rd(set ^ 1, it + 1);
#pragma unroll
        for (int i = 0; i < 13; ++i) acc0[i] = MFMA16(am[i], b0, acc0[i]);
#pragma unroll
        for (int i = 0; i < 13; ++i) acc1[i] = MFMA16(am[i], b1, acc1[i]);
        if (BAR && u4 == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 13; ++i) s += acc0[i][0] + acc0[i][1] + acc0[i][2] + acc0[i][3] + acc1[i][0] + acc1[i][1] + acc1[i][2] + acc1[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool BAR>
void run3(const char* name, int iters) {
    const int threads = 512, wgs = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)wgs * threads * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k3<BAR>), dim3(wgs), dim3(threads), 0, 0, out, cyc, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k3<BAR>), dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 26 * 2;
    printf("%-60s %6.1f cycles per MFMA per SIMD slot | %6.2f ns | clock %.2f GHz | %.1f TF\n", name, (double)c / ((double)iters * 26) / 2,
           ms * 1e6 / mfma_per_simd, (double)c / (ms * 1e6), 256.0 * 4 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

template <int MODE>
void run2(const char* name, int iters) {
    const int threads = 512, wgs = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)wgs * threads * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<MODE>), dim3(wgs), dim3(threads), 0, 0, out, cyc, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<MODE>), dim3(wgs), dim3(threads), 0, 0, out, cyc, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 35 * 2;
    printf("%-60s %6.1f cycles per MFMA per SIMD slot | %6.2f ns | clock %.2f GHz | %.1f TF\n", name, (double)c / ((double)iters * 35) / 2,
           ms * 1e6 / mfma_per_simd, (double)c / (ms * 1e6), 256.0 * 4 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

template <int NACC, int VALU>
void run(const char* name, int threads, int wgs, int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)wgs * threads * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, VALU>), dim3(wgs), dim3(threads), 0, 0, out, cyc, 10);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, VALU>), dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 64.0 / 4.0 * (wgs / 256.0);
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    printf("%-44s %5.1f waves/SIMD | %6.1f cycles per MFMA of one wave = %5.1f per SIMD slot | %6.2f ns per MFMA and SIMD | clock %.2f GHz | %.1f TF\n",
           name, waves_per_simd, (double)c / ((double)iters * NACC), (double)c / ((double)iters * NACC) / waves_per_simd,
           ms * 1e6 / mfma_per_simd, (double)c / (ms * 1e6), 256.0 * 4 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    run<35, 0>("35 accumulators, MFMA only, 1 wave/SIMD", 256, 256, it);
    run<42, 0>("42 accumulators, MFMA only, 2 waves/SIMD", 512, 256, it);
    run<35, 0>("35 accumulators, MFMA only, 2 waves/SIMD", 512, 256, it);
    run<35, 7>("35 acc + 7 VALU per 35, 2 waves/SIMD", 512, 256, it);
    run<28, 0>("28 accumulators, MFMA only, 2 waves/SIMD", 512, 256, it);
    run<8, 0>("8 accumulators, MFMA only, 2 waves/SIMD", 512, 256, it);
    run<35, 0>("35 acc, MFMA only, 2 waves/SIMD, 64 CUs", 512, 64, it);
    run2<0>("7x5 tiles, distinct A/B registers, 2 waves/SIMD", it);
    run2<1>("7x5 tiles, operands re-read from LDS per k-step", it);
    run2<2>("7x5 tiles, LDS re-read + compare/select B operands", it);
    run2<3>("7x5 tiles, exec-masked branchy asm block every 11 MFMAs", it);
    run3<false>("k_rtzw k-step: 16 LDS reads, 13 mul, 2 x 13 MFMAs", 4 * it);
    run3<true>("k_rtzw k-step + workgroup barrier every 4 k-steps", 4 * it);
    return 0;
}
