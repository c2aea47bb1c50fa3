// Micro-benchmark: HBM read bandwidth of the access patterns the R^T.Z pass can use, whole chip, 640 MB working set.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_rate.hip -o build/stream_rate && build/stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// plain 16-byte loads, grid-stride (the copy-kernel pattern)
__global__ void k_plain(const f32x4* __restrict__ p, size_t n16, float* out) {
    f32x4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}

// LDS-DMA: every wave streams 1 KB pieces into its own LDS ring, DEPTH pieces in flight.
// MODE 0: a wave owns a contiguous range; MODE 1: waves of the whole chip interleave 1 KB pieces... (piece = wave id + waves * i);
// MODE 2: a workgroup's 4 waves interleave pieces inside the workgroup's contiguous range.  NT: non-temporal hint.
template <int DEPTH, int MODE, bool NT>
__global__ __launch_bounds__(256, 2) void k_dma(const char* __restrict__ p, size_t bytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t n_pieces = bytes / 1024, n_waves = (size_t)gridDim.x * 4, gw = (size_t)blockIdx.x * 4 + wv;
    size_t first, step, count;
    if (MODE == 0) { const size_t per = n_pieces / n_waves; first = gw * per; step = 1; count = per; }
    else if (MODE == 1) { first = gw; step = n_waves; count = n_pieces / n_waves; }
    else { const size_t per = n_pieces / gridDim.x; first = (size_t)blockIdx.x * per + wv; step = 4; count = per / 4; }
    const unsigned zone0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + wv * DEPTH * 1024));
    const unsigned voff = 16 * lane;
    auto issue = [&](size_t i) {
        const unsigned long long a = (unsigned long long)(p + (first + i * step) * 1024);
        const void* base = (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a));
        const unsigned zone = zone0 + 1024u * (unsigned)(i % DEPTH);
        if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
    };
    for (size_t i = 0; i < DEPTH && i < count; ++i) issue(i);
    float s = 0.f;
    for (size_t i = 0; i < count; ++i) {
        __builtin_amdgcn_s_waitcnt((DEPTH - 1) & 15 | (7 << 4) | (15 << 8) | (((DEPTH - 1) >> 4) << 14));   // piece i has landed
        asm volatile("" ::: "memory");
        s += reinterpret_cast<const float*>(smem + wv * DEPTH * 1024 + 1024 * (i % DEPTH))[lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + DEPTH < count) issue(i + DEPTH); else asm volatile("s_nop 0");
    }
    // (the tail pieces: the counter only ever allows DEPTH-1 younger ones, so the last ones may be waited for early -- fine for a rate)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (s == 12345.f) out[0] = s;
}

// k_rtz3's own pattern: a wave owns tiles (6400 B of R rows, 3328 B of Z rows, 16 B of block ids, three arrays), NBUF tile
// buffers, a tile requested NBUF-1 tiles ahead with 12 requests (7 + 4 + 1, the last ones partial), consumed whole.
// BLK: with the 16-byte request; work: shader cycles of s_sleep per tile standing in for the arithmetic.
template <int NBUF, bool BLK, bool NT>
__global__ __launch_bounds__(256, 2) void k_tiles(const char* __restrict__ R, const char* __restrict__ Z, const char* __restrict__ B,
                                                  int tiles_per_wg, int work, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TB = 6400 + 3328 + 16, NI = 7 + 4 + (BLK ? 1 : 0);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned voff = 16 * lane;
    const unsigned zone0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + wv * NBUF * TB));
    const size_t t_first = (size_t)blockIdx.x * tiles_per_wg + wv;
    const int n_mine = (tiles_per_wg - wv + 3) / 4;
    auto dma = [&](const char* src, unsigned zone, bool on) {
        const unsigned long long a = (unsigned long long)src;
        const void* base = (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a));
        if (on) {
            if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(zone) : "memory", "m0");
        }
    };
    auto issue = [&](int i) {
        const size_t t = t_first + 4 * (size_t)i;
        const unsigned zb = zone0 + (unsigned)(i % NBUF) * TB;
        for (int p = 0; p < 7; ++p) dma(R + t * 6400 + 1024 * p, zb + 1024 * p, p < 6 || lane < 16);
        for (int p = 0; p < 4; ++p) dma(Z + t * 3328 + 1024 * p, zb + 6400 + 1024 * p, p < 3 || lane < 16);
        if (BLK) dma(B + t * 16, zb + 6400 + 3328, lane == 0);
    };
    for (int i = 0; i < NBUF - 1 && i < n_mine; ++i) issue(i);
    float s = 0.f;
    for (int i = 0; i < n_mine; ++i) {
        if (i + NBUF - 1 < n_mine) issue(i + NBUF - 1);
        constexpr int younger = (NBUF - 1) * NI;
        if (i + NBUF - 1 < n_mine) __builtin_amdgcn_s_waitcnt((younger & 15) | (7 << 4) | (15 << 8) | ((younger >> 4) << 14));
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory");
        s += reinterpret_cast<const float*>(smem + wv * NBUF * TB + (i % NBUF) * TB)[lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int w = 0; w < work; w += 64 * 16) __builtin_amdgcn_s_sleep(16);   // ~64 cycles per unit
    }
    if (s == 12345.f) out[0] = s;
}

template <typename F>
void timeit(const char* name, size_t bytes, F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.1f us  %6.2f TB/s\n", name, ms * 1e3 / 5, bytes * 5 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)640 << 20;
    char* p; float* out;
    (void)hipMalloc(&p, bytes + (1 << 20)); (void)hipMalloc(&out, 64);
    (void)hipMemset(p, 1, bytes);
    timeit("plain 16-byte loads, grid-stride, 2048 x 256", bytes, [&] { hipLaunchKernelGGL(k_plain, dim3(2048), dim3(256), 0, 0, (const f32x4*)p, bytes / 16, out); });
    timeit("plain 16-byte loads, grid-stride, 512 x 256", bytes, [&] { hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, 0, (const f32x4*)p, bytes / 16, out); });
#define DMA(D, M, N, WGS, LABEL) timeit(LABEL, bytes, [&] { hipLaunchKernelGGL((k_dma<D, M, N>), dim3(WGS), dim3(256), 4 * D * 1024, 0, (const char*)p, bytes, out); })
    DMA(8, 0, false, 512, "LDS-DMA depth 8, contiguous range per wave, 512 WGs");
    DMA(16, 0, false, 512, "LDS-DMA depth 16, contiguous range per wave, 512 WGs");
    DMA(16, 2, false, 512, "LDS-DMA depth 16, 4 waves interleave in a WG range, 512 WGs");
    DMA(16, 1, false, 512, "LDS-DMA depth 16, all waves interleave (grid-stride), 512 WGs");
    DMA(8, 1, false, 512, "LDS-DMA depth 8, all waves interleave (grid-stride), 512 WGs");
    DMA(16, 1, true, 512, "LDS-DMA depth 16, grid-stride, nt, 512 WGs");
    DMA(16, 0, true, 512, "LDS-DMA depth 16, contiguous per wave, nt, 512 WGs");
    DMA(16, 1, false, 256, "LDS-DMA depth 16, grid-stride, 256 WGs");
    {   // k_rtz3's pattern: 62500 tiles = 505 workgroups x 124 tiles (C3)
        const int wgs = 505, tpw = 123;
        const size_t tiles = (size_t)wgs * tpw, tb = tiles * (6400 + 3328 + 16);
        char *R2 = p, *Z2 = p + tiles * 6400, *B2 = Z2 + tiles * 3328;
#define TILES(NB, BL, N, W, LABEL) timeit(LABEL, tb, [&] { hipLaunchKernelGGL((k_tiles<NB, BL, N>), dim3(wgs), dim3(256), 4 * NB * (6400 + 3328 + 16), 0, R2, Z2, B2, tpw, W, out); })
        TILES(2, true, false, 0, "k_rtz3 pattern: 2 buffers, request one tile ahead, no work");
        TILES(2, false, false, 0, "k_rtz3 pattern: 2 buffers, without the 16-byte request");
        TILES(2, true, true, 0, "k_rtz3 pattern: 2 buffers, nt");
        TILES(3, true, false, 0, "k_rtz3 pattern: 3 buffers (1 WG/CU fits 2 with 117 KB? no: 1), no work");
        TILES(2, true, false, 4480, "k_rtz3 pattern: 2 buffers, 4480 cycles of work per tile");
        TILES(2, true, true, 4480, "k_rtz3 pattern: 2 buffers, nt, 4480 cycles of work per tile");
        TILES(2, true, false, 8960, "k_rtz3 pattern: 2 buffers, 8960 cycles of work per tile");
    }
    return 0;
}
