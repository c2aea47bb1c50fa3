// LISI (Local Inverse Simpson Index) on gfx950 -- the integration metric of the reference
// (harmonypy/lisi.py:24-133): exact k nearest neighbours, a per-cell perplexity search, and the
// inverse Simpson index of the labels in the neighbourhood.
//
//   k_lisi_colsum / k_lisi_center : centre the float64 input, float32 copy padded to 16-float rows,
//                                   squared norms (padding rows get +inf and never qualify)
//   k_lisi_knn<KS16, QT>          : brute-force neighbour search.  A wave owns 16*QT queries whose
//                                   fragments stay in registers and streams every 16-candidate tile
//                                   through f32 MFMA (dot products), turns them into
//                                   |c|^2 - 2 q.c, and appends the candidates below the query's
//                                   current threshold to the query's list in global memory (an LDS
//                                   counter hands out the slots).  A list that is nearly full is
//                                   sorted by the wave in LDS (bitonic, 256 keys), cut to the best
//                                   128, and the 128th key becomes the new threshold.
//   k_lisi_finish                 : one wave per cell: exact float64 distances to the 128 survivors
//                                   (float32 only preselects; 128 >= 3*perplexity leaves slack for
//                                   rank inversions of the approximation), exact ranking, first
//                                   column dropped (lisi.py:58-60), then compute_simpson
//                                   (lisi.py:83-132) in float64 and 1/simpson per label column.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hmx_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ unsigned long long ld_l2(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // past the CU's L1
}
// Ordering between the lanes of ONE wave through LDS: the DS pipe serves a wave's requests in order, so
// only the compiler has to be kept from moving LDS accesses across the hand-over points.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// float -> unsigned with the same order (negative values included)
__device__ __forceinline__ unsigned order_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ __launch_bounds__(256) void k_lisi_colsum(const double* __restrict__ X, int64_t n, int d, double* __restrict__ sums) {
    const int c = threadIdx.x;
    if (c >= d) return;
    const int64_t r0 = (int64_t)blockIdx.x * 1024, r1 = min(n, r0 + 1024);
    double s = 0.0;
    for (int64_t r = r0; r < r1; ++r) s += X[r * d + c];
    atomicAdd(sums + c, s);
}

// 16 lanes per row; rows >= n are padding
__global__ __launch_bounds__(256) void k_lisi_center(const double* __restrict__ X, int64_t n, int64_t npad, int d, int dp,
                                                     const double* __restrict__ sums, float* __restrict__ X32, float* __restrict__ cn) {
    const int l16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= npad) return;
    float ss = 0.f;
    for (int c = l16; c < dp; c += 16) {
        float v = 0.f;
        if (row < n && c < d) v = (float)(X[row * d + c] - sums[c] / (double)n);
        X32[row * dp + c] = v;
        ss += v * v;
    }
    for (int off = 8; off; off >>= 1) ss += __shfl_xor(ss, off);
    if (l16 == 0) cn[row] = row < n ? ss : __builtin_inff();
}

// ---- wave-level bitonic sort of 256 keys in LDS (ascending) ------------------------------------
__device__ __forceinline__ void wave_sort256(unsigned long long* scr, int lane) {
    for (int k = 2; k <= 256; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = lane + 64 * h;
                const int i = ((p / j) * 2 * j) + (p % j);
                const int l = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long a = scr[i], b = scr[l];
                if ((a > b) == up) { scr[i] = b; scr[l] = a; }
            }
            wave_fence();
        }
    }
}

template <int KS16, int QT>
__global__ __launch_bounds__(64 * LISI_KNN_WAVES) void k_lisi_knn(LisiKnnArgs a) {
    __shared__ unsigned long long scr_all[LISI_KNN_WAVES][LISI_CAP];
    __shared__ int cnt_all[LISI_KNN_WAVES][16 * QT];
    __shared__ float tau_all[LISI_KNN_WAVES][16 * QT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    unsigned long long* scr = scr_all[wv];
    int* cnt = cnt_all[wv];
    float* tau = tau_all[wv];
    const int64_t qbase = ((int64_t)blockIdx.x * LISI_KNN_WAVES + wv) * (16 * QT);
    if (qbase >= a.npad) return;
    for (int i = lane; i < 16 * QT; i += 64) { cnt[i] = 0; tau[i] = __builtin_inff(); }
    wave_fence();

    f32x4 bq[QT][KS16];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int m = 0; m < KS16; ++m) bq[t][m] = ld4(a.X + (size_t)(qbase + 16 * t + c16) * a.dp + 16 * m + 4 * q);
    float th[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) th[t] = __builtin_inff();

    // sort one query's list, keep the best LISI_KEEP, tighten its threshold
    auto compact = [&](int slot) {
        const int c = cnt[slot];
        unsigned long long* lst = a.lists + (size_t)(qbase + slot) * LISI_CAP;
#pragma unroll
        for (int h = 0; h < LISI_CAP / 64; ++h) { const int i = lane + 64 * h; scr[i] = i < c ? ld_l2(lst + i) : ~0ull; }
        wave_fence();
        wave_sort256(scr, lane);
        const int keep = min(c, LISI_KEEP);
#pragma unroll
        for (int h = 0; h < LISI_KEEP / 64; ++h) { const int i = lane + 64 * h; if (i < keep) lst[i] = scr[i]; }
        if (lane == 0) {
            cnt[slot] = keep;
            if (c >= LISI_KEEP) tau[slot] = order_float((unsigned)(scr[LISI_KEEP - 1] >> 32));
        }
        wave_fence();
    };

    // No register prefetch of the next tile: the list stores keep the compiler from counting
    // outstanding loads (loads and stores retire out of order), so the latency of a tile's loads is
    // covered by the other waves of the SIMD instead (2-4 resident, depending on KS16).
    const int ntiles = (int)((a.n + 15) / 16);
    for (int ct = 0; ct < ntiles; ++ct) {
        f32x4 ac[KS16];
#pragma unroll
        for (int m = 0; m < KS16; ++m) ac[m] = ld4(a.X + (size_t)(16 * ct + c16) * a.dp + 16 * m + 4 * q);
        const f32x4 cn4 = ld4(a.cn + 16 * ct + 4 * q);
        f32x4 acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KS16; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t] = MFMA16(ac[m][r], bq[t][m][r], acc[t]);
        bool full = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            unsigned long long* lst = a.lists + (size_t)(qbase + 16 * t + c16) * LISI_CAP;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float key = cn4[r] - 2.f * acc[t][r];          // |c|^2 - 2 q.c (+inf for padding rows)
                if (key < th[t]) {
                    const int slot = atomicAdd(&cnt[16 * t + c16], 1);
                    lst[slot] = ((unsigned long long)order_bits(key) << 32) | (unsigned)(16 * ct + 4 * q + r);
                }
            }
        }
        wave_fence();
#pragma unroll
        for (int t = 0; t < QT; ++t) full |= cnt[16 * t + c16] > LISI_CAP - 16;
        if (__any(full)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the list entries are in L2
            for (int slot = 0; slot < 16 * QT; ++slot)
                if (cnt[slot] > LISI_CAP - 16) compact(slot);           // wave-uniform
#pragma unroll
            for (int t = 0; t < QT; ++t) th[t] = tau[16 * t + c16];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int slot = 0; slot < 16 * QT; ++slot) {
        compact(slot);
        if (lane == 0 && qbase + slot < a.n) a.counts[qbase + slot] = cnt[slot];
    }
}

// ---- exact ranking + compute_simpson ------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(64 * LISI_FIN_WAVES) void k_lisi_finish(LisiFinishArgs a) {
    __shared__ unsigned long long key_all[LISI_FIN_WAVES][LISI_KEEP];
    __shared__ int idx_all[LISI_FIN_WAVES][LISI_KEEP];
    __shared__ double p_all[LISI_FIN_WAVES][LISI_KEEP];
    __shared__ int lab_all[LISI_FIN_WAVES][LISI_KEEP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t cell = (int64_t)blockIdx.x * LISI_FIN_WAVES + wv;
    if (cell >= a.n) return;
    unsigned long long* key = key_all[wv];
    int* idx = idx_all[wv];
    double* P = p_all[wv];
    int* lab = lab_all[wv];
    const int c = a.counts[cell];
    const unsigned long long* lst = a.lists + (size_t)cell * LISI_CAP;
    const double* xq = a.X + (size_t)cell * a.d;
    // exact squared distances to the survivors, float64 from direct differences
#pragma unroll
    for (int h = 0; h < LISI_KEEP / 64; ++h) {
        const int i = lane + 64 * h;
        unsigned long long kb = ~0ull;
        int id = 0x7FFFFFFF;
        if (i < c) {
            id = (int)(unsigned)(lst[i] & 0xFFFFFFFFull);
            const double* xc = a.X + (size_t)id * a.d;
            double s = 0.0;
            for (int k = 0; k < a.d; ++k) { const double df = xc[k] - xq[k]; s += df * df; }
            kb = (unsigned long long)__double_as_longlong(s);           // s >= 0: bit order = value order
        }
        key[i] = kb; idx[i] = id;
    }
    wave_fence();
    // bitonic sort of 128 (distance, index) pairs, ties by index
    for (int k = 2; k <= LISI_KEEP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((lane / j) * 2 * j) + (lane % j);
            const int l = i + j;
            const bool up = (i & k) == 0;
            const unsigned long long ka = key[i], kb = key[l];
            const int ia = idx[i], ib = idx[l];
            const bool gt = ka > kb || (ka == kb && ia > ib);
            if (gt == up) { key[i] = kb; key[l] = ka; idx[i] = ib; idx[l] = ia; }
            wave_fence();
        }
    }
    // neighbours = ranks 1 .. nn-1 (the first column is dropped, lisi.py:58-60)
    const int M = a.nn - 1;
    double D[2];
    int nid[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        D[h] = j < M ? sqrt(__longlong_as_double((long long)key[j + 1])) : 0.0;
        nid[h] = j < M ? idx[j + 1] : 0;
        if (j < M && a.knn_dist) { a.knn_dist[(size_t)cell * M + j] = D[h]; a.knn_idx[(size_t)cell * M + j] = nid[h]; }
    }
    // lisi.py:83-119: search beta so that the entropy of P = exp(-beta D) is log(perplexity)
    const double logU = log(a.perplexity);
    double beta = 1.0, betamin = -__builtin_inf(), betamax = __builtin_inf();
    double H = 0.0, Pn[2];
    auto entropy = [&]() {
        double p[2], s = 0.0, sd = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            p[h] = (lane + 64 * h) < M ? exp(-D[h] * beta) : 0.0;
            s += p[h];
            sd += D[h] * p[h];
        }
        s = wave_sum(s);
        sd = wave_sum(sd);
        if (s == 0.0) { H = 0.0; Pn[0] = Pn[1] = 0.0; }
        else { H = log(s) + beta * sd / s; Pn[0] = p[0] / s; Pn[1] = p[1] / s; }
    };
    entropy();
    double Hdiff = H - logU;
    for (int t = 0; t < 50; ++t) {
        if (fabs(Hdiff) < a.tol) break;
        if (Hdiff > 0) {
            betamin = beta;
            beta = isinf(betamax) ? beta * 2.0 : (beta + betamax) / 2.0;
        } else {
            betamax = beta;
            beta = isinf(betamin) ? beta / 2.0 : (beta + betamin) / 2.0;
        }
        entropy();
        Hdiff = H - logU;
    }
    // lisi.py:120-132: squared probability mass per category = sum_j P_j * (mass of j's category)
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int j = lane + 64 * h; if (j < LISI_KEEP) P[j] = j < M ? Pn[h] : 0.0; }
    wave_fence();
    for (int L = 0; L < a.n_labels; ++L) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { const int j = lane + 64 * h; if (j < M) lab[j] = a.labels[(size_t)L * a.n + nid[h]]; }
        wave_fence();
        double part = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = lane + 64 * h;
            if (j < M) {
                const int mine = lab[j];
                double mass = 0.0;
                for (int m2 = 0; m2 < M; ++m2) mass += (lab[m2] == mine) ? P[m2] : 0.0;
                part += Pn[h] * mass;
            }
        }
        double simpson = wave_sum(part);
        if (H == 0.0) simpson += -1.0;
        if (lane == 0) a.out[(size_t)cell * a.n_labels + L] = 1.0 / simpson;
        wave_fence();
    }
}

template <int KS16>
void launch_knn_qt(const LisiKnnArgs& a, hipStream_t s) {
    constexpr int QT = KS16 <= 4 ? 4 : KS16 <= 8 ? 2 : 1;
    const int64_t waves = (a.npad + 16 * QT - 1) / (16 * QT);
    const int wgs = (int)((waves + LISI_KNN_WAVES - 1) / LISI_KNN_WAVES);
    hipLaunchKernelGGL((k_lisi_knn<KS16, QT>), dim3(wgs), dim3(64 * LISI_KNN_WAVES), 0, s, a);
}

}  // namespace

void launch_lisi_prepare(const double* X, int64_t n, int64_t npad, int d, int dp, double* sums, float* X32, float* cn, hipStream_t s) {
    (void)hipMemsetAsync(sums, 0, sizeof(double) * d, s);
    hipLaunchKernelGGL(k_lisi_colsum, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, X, n, d, sums);
    hipLaunchKernelGGL(k_lisi_center, dim3((unsigned)(npad / 16)), dim3(256), 0, s, X, n, npad, d, dp, sums, X32, cn);
}

int launch_lisi_knn(const LisiKnnArgs& a, hipStream_t s) {
    switch (a.dp / 16) {
        case 1: launch_knn_qt<1>(a, s); break;
        case 2: launch_knn_qt<2>(a, s); break;
        case 3: launch_knn_qt<3>(a, s); break;
        case 4: launch_knn_qt<4>(a, s); break;
        case 5: launch_knn_qt<5>(a, s); break;
        case 6: launch_knn_qt<6>(a, s); break;
        case 7: launch_knn_qt<7>(a, s); break;
        case 8: launch_knn_qt<8>(a, s); break;
        case 9: launch_knn_qt<9>(a, s); break;
        case 10: launch_knn_qt<10>(a, s); break;
        case 11: launch_knn_qt<11>(a, s); break;
        case 12: launch_knn_qt<12>(a, s); break;
        case 13: launch_knn_qt<13>(a, s); break;
        default: return 1;
    }
    return 0;
}

void launch_lisi_finish(const LisiFinishArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_lisi_finish, dim3((unsigned)((a.n + LISI_FIN_WAVES - 1) / LISI_FIN_WAVES)), dim3(64 * LISI_FIN_WAVES), 0, s, a);
}
