// Rate at which a CU's vector memory path serves the node fetches of a tree walk: every lane reads a whole 128-byte
// record (eight 16-byte loads, as the extend kernel's node step does) at an address of its own, from a table that fits
// the L1 cache (16 KiB), the L2 (256 KiB: the size of the headline scene's tree) or neither (64 MiB).
// Reported per CU and shader cycle: 16-byte lane loads, and wave load instructions; with all 64 lanes active and with
// every second lane switched off (a diverged wave).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int TURNS = 2000;

// WIDTH: bytes per lane and load (4, 8, 16).  SHARE: lanes per record - 1: every lane reads a record of its own (LOADS
// loads from it); 2, 4, 8: that many neighbouring lanes read one record together, each its own 1/SHARE of the loads'
// addresses (what several lanes per ray would do); 64: the whole wave reads one record.
template <int LOADS, int WIDTH, int SHARE>
__global__ __launch_bounds__(256) void k_gather(const char* __restrict__ table, unsigned mask, unsigned lane_stride, unsigned long long* stamps,
                                                float* sink) {
    const unsigned lane = threadIdx.x & 63;
    unsigned idx = (blockIdx.x * 256u + (threadIdx.x / SHARE) * SHARE) * 2654435761u + 12345u;   // the same for the lanes of a group
    const unsigned sub = lane % SHARE;
    const bool active = (threadIdx.x % lane_stride) == 0;
    float acc = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (active) {
        for (int turn = 0; turn < TURNS; ++turn) {
            const char* p = table + (size_t)((idx >> 8) & mask) * 128 + (size_t)sub * (128 / SHARE >= WIDTH ? 128 / SHARE : WIDTH);
            float v[LOADS];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                if (WIDTH == 16) { const float4 x = *(const float4*)(p + (16 * k) % 128); v[k] = x.x + x.w; }
                else if (WIDTH == 8) { const float2 x = *(const float2*)(p + (8 * k) % 128); v[k] = x.x + x.y; }
                else v[k] = *(const float*)(p + (4 * k) % 128);
            }
#pragma unroll
            for (int k = 0; k < LOADS; ++k) acc += v[k];
            idx = idx * 1664525u + 1013904223u;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int LOADS, int WIDTH, int SHARE>
void run(int cus, const char* table, unsigned nodes, const char* what, unsigned long long* d_stamps, float* d_sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (unsigned lane_stride : {1u, 2u}) {
        if (SHARE > 1 && lane_stride > 1) continue;
        const int blocks_per_cu = 6;
        const int blocks = cus * blocks_per_cu;
        k_gather<LOADS, WIDTH, SHARE><<<blocks, 256>>>(table, nodes - 1, lane_stride, d_stamps, d_sink);
        (void)hipEventRecord(e0);
        k_gather<LOADS, WIDTH, SHARE><<<blocks, 256>>>(table, nodes - 1, lane_stride, d_stamps, d_sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * (size_t)blocks * 4);
        (void)hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (size_t w = 0; w < (size_t)blocks * 4; ++w) mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double clock = mhz[mhz.size() / 2] * 1e6;
        const double cycles = ms * 1e-3 * clock;
        const double wave_loads = (double)blocks_per_cu * 4 * TURNS * LOADS;   // per CU
        const double lane_loads = wave_loads * (64 / lane_stride);
        printf("%-18s %d loads x %2d B, %2d lanes per record, %2u lanes/wave active: %6.3f lane loads / cycle / CU, %6.2f cycles per wave load, "
               "%5.1f B / cycle / CU (%.3f ms, %.0f MHz)\n",
               what, LOADS, WIDTH, SHARE, 64 / lane_stride, lane_loads / cycles, cycles / wave_loads, lane_loads * WIDTH / cycles, ms, clock / 1e6);
    }
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_stamps;
    float* d_sink;
    char* table;
    const size_t max_nodes = (64u << 20) / 128;
    (void)hipMalloc(&d_stamps, 2 * (size_t)cus * 8 * 4 * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 16);
    (void)hipMalloc(&table, max_nodes * 128 + 256);
    (void)hipMemset(table, 0, max_nodes * 128 + 256);
    printf("%s, %d CUs, 6 waves per SIMD\n", prop.name, cus);
    const unsigned l1 = (16u << 10) / 128, l2 = (256u << 10) / 128;
    run<8, 16, 1>(cus, table, l1, "16 KiB (L1)", d_stamps, d_sink);
    run<8, 16, 1>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<8, 16, 1>(cus, table, (unsigned)max_nodes, "64 MiB (MALL)", d_stamps, d_sink);
    run<4, 16, 1>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<8, 8, 1>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<8, 4, 1>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<4, 16, 2>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<2, 16, 4>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<1, 16, 8>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<8, 16, 64>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    run<8, 4, 64>(cus, table, l2, "256 KiB (L2)", d_stamps, d_sink);
    return 0;
}
