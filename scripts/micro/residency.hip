// Micro-benchmark: how many workgroups of a (256 threads, ~220 VGPRs, 68 KB LDS) kernel are resident at once?
// Every workgroup counts itself in, then waits (bounded) until `want` have counted in; reports how many it saw.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/residency.hip -o build/residency && build/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_res(unsigned* counter, unsigned* seen, int want, float* out, int iters) {
    extern __shared__ unsigned char smem[];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)threadIdx.x};
    if (threadIdx.x == 0) atomicAdd(counter, 1u);
    unsigned s = 0;
    if (threadIdx.x == 0) {
        for (int spin = 0; spin < 2000000; ++spin) {
            s = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s >= (unsigned)want) break;
            __builtin_amdgcn_s_sleep(4);
        }
        seen[blockIdx.x] = s;
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)it, 1.0f, acc[i], 0, 0, 0);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][3];
    if (t == 12345.f || smem[threadIdx.x] == 77) out[0] = t;
}
template <int NACC>
void run(int wgs, size_t lds, const char* what) {
    unsigned *counter, *seen;
    float* out;
    hipMalloc(&counter, 4); hipMalloc(&seen, wgs * 4); hipMalloc(&out, 4);
    hipMemset(counter, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_res<NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL((k_res<NACC>), dim3(wgs), dim3(256), lds, 0, counter, seen, wgs, out, 4);
    hipDeviceSynchronize();
    std::vector<unsigned> h(wgs);
    hipMemcpy(h.data(), seen, wgs * 4, hipMemcpyDeviceToHost);
    int full = 0; unsigned mn = ~0u;
    for (unsigned v : h) { full += v >= (unsigned)wgs; mn = std::min(mn, v); }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_res<NACC>));
    printf("%-44s %d workgroups, %zu B LDS, %d VGPRs: %d saw all %d counted in (smallest count seen: %u)\n", what, wgs, lds, fa.numRegs, full, wgs, mn);
    hipFree(counter); hipFree(seen); hipFree(out);
}
int main() {
    run<8>(512, 70000, "few registers, 70 KB LDS");
    run<48>(512, 70000, "~200 registers, 70 KB LDS");
    run<52>(490, 70032, "~220 registers, 70 KB LDS, 490 workgroups");
    run<52>(512, 1024, "~220 registers, 1 KB LDS");
    run<56>(512, 1024, "~240 registers, 1 KB LDS");
    return 0;
}
