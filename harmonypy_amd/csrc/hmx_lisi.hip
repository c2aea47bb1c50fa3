// LISI (Local Inverse Simpson Index) on gfx950 -- the integration metric of the reference
// (harmonypy/lisi.py:24-133): exact k nearest neighbours, a per-cell perplexity search, and the
// inverse Simpson index of the labels in the neighbourhood.
//
//   k_lisi_colsum / k_lisi_center : centre the float64 input, float32 copy padded to 16-float rows,
//                                   squared norms (padding rows get +inf and never qualify)
//   k_lisi_knn<KS16, QT>          : brute-force neighbour search.  A wave owns 16*QT queries whose
//                                   fragments stay in registers and streams every 16-candidate tile
//                                   (shared by the workgroup's waves through LDS) through f32 MFMA, turns them into
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

// ---- wave-level bitonic sort of N keys in LDS (ascending), N / 128 compare-exchanges per lane and step ---------
template <int N>
__device__ __forceinline__ void wave_sort(unsigned long long* scr, int lane) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int h = 0; h < N / 128; ++h) {
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

// Candidate lists come in three sizes (CAP entries, the best CAP / 2 kept by a compaction and ranked exactly afterwards):
// 256 for 3 * perplexity <= 120 neighbours (the reference's default is 90), 1024 up to 504, 4096 up to 2040.  The large
// sizes keep the sort scratch in dynamic LDS (32 KB / 128 KB per workgroup) and so run fewer workgroups per CU.
template <int KS16, int QT, int CAP>
__global__ __launch_bounds__(64 * LISI_KNN_WAVES, CAP == 256 ? 3 : 1) void k_lisi_knn(LisiKnnArgs a) {
    constexpr int KEEP = CAP / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned long long scr_all[];   // waves x CAP
    __shared__ int cnt_all[LISI_KNN_WAVES][16 * QT];
    __shared__ float tau_all[LISI_KNN_WAVES][16 * QT];
    constexpr int LDW = 16 * KS16 + 4;                       // padded row: conflict-free 16-byte fragment reads
    __shared__ __attribute__((aligned(16))) float stage[2][16 * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    unsigned long long* scr = scr_all + (size_t)wv * CAP;
    int* cnt = cnt_all[wv];
    float* tau = tau_all[wv];
    const int64_t qbase = ((int64_t)blockIdx.x * LISI_KNN_WAVES + wv) * (16 * QT);   // < npad: npad is a multiple of 256
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

    // sort one query's list, keep the best KEEP, tighten its threshold
    auto compact = [&](int slot) {
        const int c = cnt[slot];
        unsigned long long* lst = a.lists + (size_t)(qbase + slot) * CAP;
#pragma unroll 4
        for (int h = 0; h < CAP / 64; ++h) { const int i = lane + 64 * h; scr[i] = i < c ? ld_l2(lst + i) : ~0ull; }
        wave_fence();
        wave_sort<CAP>(scr, lane);
        const int keep = min(c, KEEP);
#pragma unroll 4
        for (int h = 0; h < KEEP / 64; ++h) { const int i = lane + 64 * h; if (i < keep) lst[i] = scr[i]; }
        if (lane == 0) {
            cnt[slot] = keep;
            if (c >= KEEP) tau[slot] = order_float((unsigned)(scr[KEEP - 1] >> 32));
        }
        wave_fence();
    };

    // Software pipeline over the candidate tiles.  The workgroup's four waves share every tile
    // through LDS (one global read per workgroup instead of one per wave: the float32 matrix is
    // streamed by every workgroup, so the reuse decides whether the loop is L2/MALL- or MFMA-bound):
    //   iteration t:  issue the global loads of tile t+2's pieces  ->  fragments of tile t from
    //   LDS, MFMAs of tile t  ->  pieces of tile t+1 (issued an iteration ago) into the other LDS
    //   buffer  ->  tile t-1's products into keys, the qualifying ones appended to the
    //   lists  ->  barrier.
    // The list stores are issued from inline assembly on purpose: with ordinary stores in the loop the
    // compiler waits for vmcnt(0) before every use of a loaded value (it cannot bound a counter that
    // loads and stores leave out of order).  Its "at most as many outstanding as loads issued after
    // the wanted one" is still sufficient: loads return in order among themselves, so while a wanted
    // load is outstanding every younger load is too and the counter stays above the bound; a store
    // in flight can only lengthen the wait.
    const int ntiles = (int)((a.n + 15) / 16);
    constexpr int NPC = (64 * KS16 + 64 * LISI_KNN_WAVES - 1) / (64 * LISI_KNN_WAVES);   // 16-byte pieces per thread
    f32x4 pre[2][NPC], af[KS16], acc[2][QT];
    auto fetch_pieces = [&](int ps, int tile) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = tid + 64 * LISI_KNN_WAVES * j;
            const int row = i / (4 * KS16), c4 = i - row * (4 * KS16);
            if (i < 64 * KS16) pre[ps][j] = ld4(a.X + (size_t)(16 * tile + row) * a.dp + 4 * c4);
        }
    };
    auto store_pieces = [&](int ps, int sb) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = tid + 64 * LISI_KNN_WAVES * j;
            const int row = i / (4 * KS16), c4 = i - row * (4 * KS16);
            if (i < 64 * KS16) *reinterpret_cast<f32x4*>(&stage[sb][row * LDW + 4 * c4]) = pre[ps][j];
        }
    };
    auto read_fragments = [&](int sb) {
#pragma unroll
        for (int m = 0; m < KS16; ++m) af[m] = *reinterpret_cast<const f32x4*>(&stage[sb][c16 * LDW + 16 * m + 4 * q]);
    };
    auto multiply = [&](int p) {
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[p][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KS16; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[p][t] = MFMA16(af[m][r], bq[t][m][r], acc[p][t]);
    };
    // Tile products -> keys -> list entries.  Most tiles add nothing once the thresholds are tight: one
    // wave-wide test leaves early.  A qualifying value takes its list position from the query's LDS
    // counter and is stored from inline assembly (see below); the positions handed out tell whether a
    // list is about to overflow, so the counters are not read back.
    auto append = [&](int p, const f32x4 cn4, int tile) {
        bool any = false;
        float key[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                key[t][r] = cn4[r] - 2.f * acc[p][t][r];                 // |c|^2 - 2 q.c (+inf for padding rows)
                any |= key[t][r] < th[t];
            }
        if (!__any(any)) return;                                        // the common case once the thresholds are tight
        // Qualifying values are few and scattered: every lane walks the set bits of its own 16-value
        // mask, so the wave loops as often as its busiest lane has entries (once, typically) instead
        // of branching around sixteen value slots.
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mask |= (key[t][r] < th[t]) ? (1u << (4 * t + r)) : 0u;
        bool full = false;
        while (__any(mask != 0)) {
            if (mask != 0) {
                const int j = __ffs(mask) - 1;
                mask &= mask - 1;
                float kv = key[0][0];
#pragma unroll
                for (int jj = 1; jj < 4 * QT; ++jj) kv = (j == jj) ? key[jj >> 2][jj & 3] : kv;
                const int t = j >> 2, r = j & 3;
                const int slot = atomicAdd(&cnt[16 * t + c16], 1);
                const unsigned long long ent = ((unsigned long long)order_bits(kv) << 32) | (unsigned)(16 * tile + 4 * q + r);
                unsigned long long* dst = a.lists + (size_t)(qbase + 16 * t + c16) * CAP + slot;
                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(ent) : "memory");
                full |= slot >= CAP - 16;                          // the list now holds more than CAP-16 entries
            }
        }
        if (__any(full)) {
            wave_fence();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the list entries are in L2
            for (int slot = 0; slot < 16 * QT; ++slot)
                if (cnt[slot] > CAP - 16) compact(slot);           // wave-uniform
#pragma unroll
            for (int t = 0; t < QT; ++t) th[t] = tau[16 * t + c16];
        }
    };
    // the query fragments must have landed before the loop: a wait for them left inside the loop
    // would also wait for the prefetches issued there
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int m = 0; m < KS16; ++m) asm volatile("" ::"v"(bq[t][m][3]));
    fetch_pieces(0, 0);
    store_pieces(0, 0);
    if (ntiles > 1) fetch_pieces(1, 1);
    __syncthreads();
#ifdef LISI_PROF   // cycle stamps per phase segment (timing experiments only)
    unsigned long long pf[6] = {0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
#define LISI_STAMP(k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); pf[k] += now - pt; pt = now; }
#else
#define LISI_STAMP(k)
#endif
    for (int ct = 0; ct < ntiles; ct += 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int t = ct + p;
            if (t < ntiles) {                                           // workgroup-uniform
                const bool more = t + 1 < ntiles;
                // addresses of both loads first, in registers of their own: the compiler otherwise recycles
                // the first load's address registers for the second and waits for the first load to return
                int cn_off = 16 * (t > 0 ? t - 1 : 0) + 4 * q;
                asm volatile("" : "+v"(cn_off));
                if (t + 2 < ntiles) fetch_pieces(p, t + 2);             // two tiles ahead: a round trip can exceed one MFMA phase
                const f32x4 cn_prev = ld4(a.cn + cn_off);               // norms of the tile appended below
                __builtin_amdgcn_sched_barrier(0);                      // loads first: their latency runs under the MFMAs
                LISI_STAMP(0)
                read_fragments(p);
                multiply(p);
                __builtin_amdgcn_sched_barrier(0);
                LISI_STAMP(1)
                if (more) store_pieces(p ^ 1, p ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                LISI_STAMP(2)
                if (t > 0) append(p ^ 1, cn_prev, t - 1);
                LISI_STAMP(3)
                __syncthreads();
                LISI_STAMP(4)
            }
        }
    }
#ifdef LISI_PROF
    if (lane == 0 && a.prof) for (int k = 0; k < 6; ++k) atomicAdd(a.prof + k, pf[k]);
#endif
    {
        const f32x4 cn_last = ld4(a.cn + 16 * (ntiles - 1) + 4 * q);
        if ((ntiles - 1) & 1) append(1, cn_last, ntiles - 1);
        else append(0, cn_last, ntiles - 1);
    }
    wave_fence();
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

template <int KEEP, int FW>
__global__ __launch_bounds__(64 * FW) void k_lisi_finish(LisiFinishArgs a) {
    constexpr int NH = KEEP / 64;                                       // list positions per lane
    __shared__ unsigned long long key_all[FW][KEEP];
    __shared__ int idx_all[FW][KEEP];
    __shared__ double p_all[FW][KEEP];
    __shared__ int lab_all[FW][KEEP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t cell = (int64_t)blockIdx.x * FW + wv;
    if (cell >= a.n) return;
    unsigned long long* key = key_all[wv];
    int* idx = idx_all[wv];
    double* P = p_all[wv];
    int* lab = lab_all[wv];
    const int c = a.counts[cell];
    const unsigned long long* lst = a.lists + (size_t)cell * (2 * KEEP);
    const double* xq = a.X + (size_t)cell * a.d;
    // exact squared distances to the survivors, float64 from direct differences
#pragma unroll 2
    for (int h = 0; h < NH; ++h) {
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
    // bitonic sort of the KEEP (distance, index) pairs, ties by index
    for (int k = 2; k <= KEEP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int h = 0; h < KEEP / 128; ++h) {
                const int p = lane + 64 * h;
                const int i = ((p / j) * 2 * j) + (p % j);
                const int l = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long ka = key[i], kb = key[l];
                const int ia = idx[i], ib = idx[l];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == up) { key[i] = kb; key[l] = ka; idx[i] = ib; idx[l] = ia; }
            }
            wave_fence();
        }
    }
    // neighbours = ranks 1 .. nn-1 (the first column is dropped, lisi.py:58-60)
    const int M = a.nn - 1;
    double D[NH];
    int nid[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int j = lane + 64 * h;
        D[h] = j < M ? sqrt(__longlong_as_double((long long)key[j + 1])) : 0.0;
        nid[h] = j < M ? idx[j + 1] : 0;
        if (j < M && a.knn_dist) { a.knn_dist[(size_t)cell * M + j] = D[h]; a.knn_idx[(size_t)cell * M + j] = nid[h]; }
    }
    // lisi.py:83-119: search beta so that the entropy of P = exp(-beta D) is log(perplexity)
    const double logU = log(a.perplexity);
    double beta = 1.0, betamin = -__builtin_inf(), betamax = __builtin_inf();
    double H = 0.0, Pn[NH];
    auto entropy = [&]() {
        double p[NH], s = 0.0, sd = 0.0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            p[h] = (lane + 64 * h) < M ? exp(-D[h] * beta) : 0.0;
            s += p[h];
            sd += D[h] * p[h];
        }
        s = wave_sum(s);
        sd = wave_sum(sd);
        if (s == 0.0) {
            H = 0.0;
#pragma unroll
            for (int h = 0; h < NH; ++h) Pn[h] = 0.0;
        } else {
            H = log(s) + beta * sd / s;
#pragma unroll
            for (int h = 0; h < NH; ++h) Pn[h] = p[h] / s;
        }
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
    for (int h = 0; h < NH; ++h) { const int j = lane + 64 * h; P[j] = j < M ? Pn[h] : 0.0; }
    wave_fence();
    for (int L = 0; L < a.n_labels; ++L) {
#pragma unroll
        for (int h = 0; h < NH; ++h) { const int j = lane + 64 * h; if (j < M) lab[j] = a.labels[(size_t)L * a.n + nid[h]]; }
        wave_fence();
        double part = 0.0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
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

template <int KS16, int CAP>
void launch_knn_cap(const LisiKnnArgs& a, hipStream_t s) {
    constexpr int QT = KS16 <= 4 ? 4 : KS16 <= 8 ? 2 : 1;
    const int64_t waves = (a.npad + 16 * QT - 1) / (16 * QT);
    const int wgs = (int)((waves + LISI_KNN_WAVES - 1) / LISI_KNN_WAVES);
    const size_t sm = (size_t)LISI_KNN_WAVES * CAP * sizeof(unsigned long long);
    static bool attr_done = false;
    if (!attr_done && sm > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lisi_knn<KS16, QT, CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_lisi_knn<KS16, QT, CAP>), dim3(wgs), dim3(64 * LISI_KNN_WAVES), sm, s, a);
}
template <int KS16>
void launch_knn_qt(const LisiKnnArgs& a, hipStream_t s) {
    if (a.cap <= 256) launch_knn_cap<KS16, 256>(a, s);
    else if (a.cap <= 1024) launch_knn_cap<KS16, 1024>(a, s);
    else launch_knn_cap<KS16, 4096>(a, s);
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
    if (a.cap <= 256) hipLaunchKernelGGL((k_lisi_finish<128, 4>), dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, a);
    else if (a.cap <= 1024) hipLaunchKernelGGL((k_lisi_finish<512, 4>), dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_lisi_finish<2048, 1>), dim3((unsigned)a.n), dim3(64), 0, s, a);
}

int lisi_list_cap(int nn) { return nn <= 128 - 8 ? 256 : nn <= 512 - 8 ? 1024 : nn <= 2048 - 8 ? 4096 : 0; }
