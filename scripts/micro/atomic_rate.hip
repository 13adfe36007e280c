// Rate of returned atomic adds from many resident blocks: on one address, and spread over K addresses
// (one cache line each).  Build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void k(unsigned* counters, int n_addr, int iters, unsigned* sink) {
    __shared__ unsigned s_base;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x == 0) s_base = atomicAdd(&counters[(blockIdx.x % n_addr) * 32], 37u);
        __syncthreads();
        acc += s_base;
        __syncthreads();
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main() {
    unsigned *c, *sink;
    (void)hipMalloc(&c, 64 * 32 * sizeof(unsigned));
    (void)hipMalloc(&sink, 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int iters = 2000;
    for (int blocks : {256, 768, 1536})
        for (int n_addr : {1, 2, 4, 8, 16, 64}) {
            (void)hipMemset(c, 0, 64 * 32 * sizeof(unsigned));
            k<<<blocks, 512>>>(c, n_addr, 10, sink);
            (void)hipEventRecord(a);
            k<<<blocks, 512>>>(c, n_addr, iters, sink);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            printf("blocks %4d addresses %2d: %.1f M atomics/s, %.2f us per block iteration\n", blocks, n_addr,
                   (double)blocks * iters / ms / 1e3, ms * 1e3 / iters);
        }
    return 0;
}
