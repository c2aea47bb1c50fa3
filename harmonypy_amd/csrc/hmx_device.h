// hmx_device.h -- device-side helpers shared by the kernel files (hmx_kernels.hip, hmx_rtz3.hip).
//
// Fragment conventions of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md §3):
//   A operand: lane l holds A[i = l&15][k = l>>4]
//   B operand: lane l holds B[k = l>>4][j = l&15]
//   C/D      : lane l, reg r holds C[row = 4*(l>>4) + r][col = l&15]
// Throughout: c16 = lane & 15, q = lane >> 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float wave_sum_q(float v) {
    // sum over the 4 lanes that share c16 (q = 0..3)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// the same sum without the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) exchange 32-lane halves and
// 16-lane rows between two registers; with both operands = v the two results add up to v[l] + v[l ^ 32] (then ^ 16)
// in every lane.  Two VALU instructions per step, no lgkmcnt wait.
__device__ __forceinline__ float wave_sum_q_swap(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_sum_c16(float v) {
    // sum over the 16 lanes that share q
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_all(double v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- fp32 products on the bf16 matrix pipe ("three-term split") --------------------------------------------------
// gfx950 has no xf32 and its f32-input MFMA runs at the vector rate (16x16x4: 32 cycles per SIMD for 2 x 16 x 16 x 4
// flops); v_mfma_f32_16x16x32_bf16 does eight times the K in half the cycles.  An fp32 value is the EXACT sum of three
// bf16 values, x = h + m + l: h = RNE(x) to 8 significant bits, m = RNE(x - h), l = x - h - m (both subtractions are
// exact, and what is left after m has at most 8 significant bits).  A product of two bf16 values is exact in fp32 and
// the matrix pipe accumulates in fp32, so
//     x y = hh + (hm + mh) + (mm + hl + lh) + [ml + lm + ll],      |[...]| <= 2^-23 |x y|  (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|),
// i.e. six bf16 MFMAs give the fp32 product sum up to 2^-23 of ONE product -- against the rounding of the fp32
// accumulator (2^-24 of the running sum of all d products) that vanishes: in tests/test_split_gemm.py the six-product
// form is as close to the float64 value as the f32-input MFMA, a shade closer in fact (the accumulator is rounded 6
// times per 32 k instead of 8), and adding the dropped terms changes nothing -- at 6 x 16 cycles per 32 k against 8 x 32.
// Fragment conventions of v_mfma_f32_16x16x32_bf16:
//   A operand: lane l holds A[i = l&15][k = 8 (l>>4) .. +8] (8 bf16 = 4 registers), B likewise B[k][j = l&15], C/D as above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ u32x4 ld4u(const unsigned* p) { return *reinterpret_cast<const u32x4*>(p); }
// two floats -> two bf16 in one register (v_cvt_pk_bf16_f32, round to nearest even), and back (exact)
__device__ __forceinline__ unsigned bf16_pack(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 bf16_widen(unsigned p) {
    f32x2 r;
    r.x = __uint_as_float(p << 16);
    r.y = __uint_as_float(p & 0xffff0000u);
    return r;
}
// x = h + m + l exactly (pairs: 3 conversions, 2 packed subtractions, 4 shifts / masks).  (Four single v_sub_f32 instead of
// the two v_pk_add_f32 were measured in round 5 -- packed fp32 instructions are dear beside MFMAs -- and lost: C3 191.6 vs
// 194.8 M cells/s/iteration on one box, profiles/r05_ab_split_single_subs.txt.)
__device__ __forceinline__ void bf16_split3(f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_pack(x);
    const f32x2 r1 = x - bf16_widen(h);
    m = bf16_pack(r1);
    l = bf16_pack(r1 - bf16_widen(m));
}

// sum over the 16 lanes of a DPP row (the lanes that share q); every lane of the row gets the total
#define DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {
    v += DPP_F(v, 0xB1);   // quad_perm [1,0,3,2]
    v += DPP_F(v, 0x4E);   // quad_perm [2,3,0,1]
    v += DPP_F(v, 0x141);  // row_half_mirror
    v += DPP_F(v, 0x140);  // row_mirror
    return v;
}

// exp(x) as 2^(x log2 e) for finite x: the product is split into a rounded head and an fma tail so
// that the relative error stays ~1 ulp over the whole range of arguments (|x| up to ~100 would
// otherwise lose 2^-24 * |x| log2 e).  v_exp_f32 is the hardware exp2.
__device__ __forceinline__ float fast_exp_finite(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
    const float th = x * L2E_HI;
    const float tl = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -th));
    const float p = __builtin_amdgcn_exp2f(th);
    return fmaf(p, tl * 0.693147182464599609375f, p);
}
// r ** t for r in [1e-8, 1], t >= 0 as 2^(t log2 r): v_log_f32 / v_exp_f32 (1 ulp each) with the
// product split into a rounded head and an fma tail.  Relative error ~3e-7 t (measured against powf
// over the clamp range), far below what moves a round decision.
__device__ __forceinline__ float pow_unit(float r, float t) {
    const float l = __builtin_amdgcn_logf(r);            // log2 r <= 0
    const float hi = t * l;
    const float lo = fmaf(t, l, -hi);
    const float p = __builtin_amdgcn_exp2f(hi);
    return fmaf(p, lo * 0.693147182464599609375f, p);
}

// ---- hand-off between workgroups: 8-byte agent-scope atomics on both sides (Guideline 16) ----
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a value handed to other workgroups as a RETURNING read-modify-write at agent scope: the old value comes back from where the
// exchange was performed, so the writer's vmcnt(0) means "performed", not merely "accepted" (a non-returning atomic or a store
// is acknowledged earlier: a flag raised behind it was seen by other XCDs before the data -- measured in round 3 on the persistent
// wide sweep)
__device__ __forceinline__ void xchg_agent(float* p, float v) {
    const float old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
}
__device__ __forceinline__ void xchg_agent(unsigned* p, unsigned v) {
    const unsigned old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
}
// peer boxes (other GPUs over xGMI): 8-byte system-scope atomics on both sides
__device__ __forceinline__ double ld_sys(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void xchg_sys(double* p, double v) {   // returning: performed at the peer when the wave's vmcnt(0) is through (see xchg_agent)
    const double old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("" ::"v"(old));
}
__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// A rank's box: data [parity 2][source rank][G x K16] doubles, then flags [parity 2][source rank] u64,
// then 2 x n_ranks u64 of self-test words.
__device__ __host__ __forceinline__ size_t box_data(int n_ranks, size_t GK, int par, int src) { return ((size_t)par * n_ranks + src) * GK; }
__device__ __host__ __forceinline__ size_t box_flags(int n_ranks, size_t GK) { return (size_t)2 * n_ranks * GK; }

__device__ __forceinline__ void wg_barrier_lds() {
    // workgroup barrier that orders LDS traffic only: global loads already in flight (the next
    // tiles' operands) keep travelling instead of being drained as __syncthreads() would
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
