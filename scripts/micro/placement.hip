// Micro-benchmark: where do the workgroups of a (256 threads, 2 per CU) launch land, and what do the two waves that share a
// SIMD see in HW_ID?  Prints, for the first CUs, the blockIdx values per (XCC, SE, CU), and per SIMD the wave slots.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/placement.hip -o build/placement && build/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(256, 2) void k_place(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // XCC_ID
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);                      // stay resident while the others arrive
    if (lane == 0) {
        unsigned* o = out + ((size_t)blockIdx.x * 4 + wv) * 4;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = (unsigned)(t0 >> 32);
    }
    if (smem[threadIdx.x] == 77) out[0] = 1;
}
int main() {
    const int wgs = 512;
    unsigned* d;
    hipMalloc(&d, (size_t)wgs * 16 * sizeof(unsigned));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_place), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(k_place, dim3(wgs), dim3(256), 79 * 1024, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)wgs * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::vector<int>> cu;   // (xcc, se, sh, cu) -> blockIdx list
    printf("HW_ID of the 4 waves of workgroups 0, 1, 8, 256, 257 (hex), XCC_ID:\n");
    for (int b : {0, 1, 8, 256, 257, 264})
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[((size_t)b * 4 + w) * 4], x = h[((size_t)b * 4 + w) * 4 + 1];
            printf("  wg %3d wave %d: HW_ID %08x  wave_slot %u simd %u pipe %u cu %u sh %u se %u | xcc %u\n", b, w, hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15,
                   (hw >> 12) & 1, (hw >> 13) & 7, x & 15);
        }
    for (int b = 0; b < wgs; ++b) {
        const unsigned hw = h[(size_t)b * 16], x = h[(size_t)b * 16 + 1] & 15;
        cu[{x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15}].push_back(b);
    }
    printf("%zu distinct (xcc, se, sh, cu); first 12:\n", cu.size());
    int n = 0;
    for (auto& kv : cu) {
        if (n++ >= 12) break;
        printf("  xcc %u se %u sh %u cu %2u:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    // same-SIMD pairs: slot parity
    int same_par = 0, diff_par = 0, gen_same = 0, gen_diff = 0;
    for (auto& kv : cu)
        if (kv.second.size() == 2) {
            const int a = kv.second[0], b = kv.second[1];
            gen_same += ((a >> 8) & 1) == ((b >> 8) & 1);
            gen_diff += ((a >> 8) & 1) != ((b >> 8) & 1);
            for (int wa = 0; wa < 4; ++wa)
                for (int wb = 0; wb < 4; ++wb) {
                    const unsigned ha = h[((size_t)a * 4 + wa) * 4], hb = h[((size_t)b * 4 + wb) * 4];
                    if (((ha >> 4) & 3) == ((hb >> 4) & 3)) { if ((ha & 1) == (hb & 1)) ++same_par; else ++diff_par; }
                }
        }
    printf("CUs with two workgroups: blockIdx>>8 parity differs on %d, equal on %d; same-SIMD wave pairs: slot parity differs %d, equal %d\n", gen_diff, gen_same, diff_par, same_par);
    return 0;
}
