// Micro-benchmark: what does one hop of a grid-wide hand-off cost, by placement and by how the waiting side reads?
// n workgroups (one wave each) pass a token round a ring: workgroup i waits until word[i] == round, then publishes
// word[i+1] = round.  Time per hop = kernel time / (rounds x n).  Variants:
//   read:  0 = agent-scope atomic load (global_load sc1, what k_round polls with), 1 = returning atomic OR 0 (executes at
//          the L2 like every read-modify-write), 2 = workgroup-scope load (sc0: may hit the CU's L1 -- expected to hang, bounded)
//   place: stride 1 = consecutive workgroups = consecutive XCDs (round-robin dispatch); stride 8 = all on one XCD
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/handoff.hip -o build/handoff && build/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int READ>
__global__ __launch_bounds__(64) void k_ring(unsigned* word, int n, int stride, int rounds, unsigned* xcc_out, unsigned long long* fail) {
    if (blockIdx.x % stride) return;
    const int me = blockIdx.x / stride, next = (me + 1) % n;
    if (threadIdx.x == 0) xcc_out[me] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 15;
    for (int r = 1; r <= rounds; ++r) {
        const unsigned want = me == 0 ? (unsigned)(r - 1) : (unsigned)r;   // workgroup 0 starts round r when round r-1 came back
        unsigned spins = 0;
        while (true) {
            unsigned v;
            if (READ == 0) v = __hip_atomic_load(word + 32 * me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (READ == 1) v = __hip_atomic_fetch_or(word + 32 * me, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __hip_atomic_load(word + 32 * me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v >= want) break;
            if (++spins > (1u << 22)) { if (threadIdx.x == 0) atomicAdd(fail, 1ull); return; }
        }
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_exchange(word + 32 * next, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));
        }
    }
}
int main() {
    unsigned *word, *xcc; unsigned long long* fail;
    hipMalloc(&word, 64 * 32 * 4); hipMalloc(&xcc, 64 * 4); hipMalloc(&fail, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 2000;
    for (int read = 0; read < 3; ++read)
        for (int stride : {1, 8})
            for (int n : {2, 16}) {
                hipMemset(word, 0, 64 * 32 * 4); hipMemset(fail, 0, 8);
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(word, 0, 64 * 32 * 4);
                    hipEventRecord(e0);
                    if (read == 0) hipLaunchKernelGGL(k_ring<0>, dim3(n * stride), dim3(64), 0, 0, word, n, stride, rounds, xcc, fail);
                    else if (read == 1) hipLaunchKernelGGL(k_ring<1>, dim3(n * stride), dim3(64), 0, 0, word, n, stride, rounds, xcc, fail);
                    else hipLaunchKernelGGL(k_ring<2>, dim3(n * stride), dim3(64), 0, 0, word, n, stride, rounds, xcc, fail);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                unsigned long long f; hipMemcpy(&f, fail, 8, hipMemcpyDeviceToHost);
                std::vector<unsigned> x(64); hipMemcpy(x.data(), xcc, 64 * 4, hipMemcpyDeviceToHost);
                printf("read %d (%s) stride %d n %2d: %.3f us per hop%s   xcc of the first workgroups:", read,
                       read == 0 ? "agent load sc1" : read == 1 ? "returning atomic or" : "workgroup load sc0", stride, n,
                       1e3 * best / (rounds * n), f ? "  [TIMED OUT]" : "");
                for (int i = 0; i < (n < 6 ? n : 6); ++i) printf(" %u", x[i]);
                printf("\n");
            }
    return 0;
}
