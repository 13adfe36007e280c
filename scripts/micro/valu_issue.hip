// What a vector instruction of gfx950 costs a SIMD, by opcode and by the number of waves resident on the SIMD.
//
// Each wave runs a stream of independent instructions of one opcode (16 accumulators, so no instruction waits for the one
// before it) and stamps the shader clock before and after; W waves share a SIMD (W blocks of 256 threads per CU, one
// wave of each block per SIMD).  Reported per opcode and W: shader cycles of SIMD time per wave-instruction, from the
// launch's wall time and the clock rate the waves measured (s_memtime against the 100 MHz s_memrealtime), and the
// cycles a single wave took per instruction by its own stamps.  2 means a SIMD retires one such wave-instruction every
// 2 cycles once enough waves are resident (the 32-lane SIMD of CDNA4 at full rate); 4 means it does so every 4 cycles
// whatever is resident.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int UNROLL = 64;   // instructions per loop turn (16 accumulators x 4)
constexpr int TURNS = 4000;

typedef float float2v __attribute__((ext_vector_type(2)));

// One kernel per instruction: ASM is applied to accumulator a[i] (float), p[i] (float pair) or u[i] (unsigned), with the
// loop-invariant operands m, c (float), pm (float pair), um (unsigned).
#define ISSUE_KERNEL(NAME, ASM)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* stamps, float* sink, float seed) {              \
        float a[16];                                                                                                 \
        float2v p[16];                                                                                               \
        unsigned u[16];                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                             \
            a[i] = seed + (float)i + (float)threadIdx.x;                                                             \
            p[i] = float2v{a[i], a[i] * 0.5f};                                                                       \
            u[i] = threadIdx.x * 977u + i;                                                                           \
        }                                                                                                            \
        const float m = 1.0000001f, c = 1e-9f;                                                                       \
        const float2v pm = {1.0000001f, 0.9999999f};                                                                 \
        const unsigned um = 0x9E3779B9u;                                                                             \
        __syncthreads();                                                                                             \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_mov_b64 s[20:21], vcc" :: "v"(a[0]), "v"(m) : "vcc", "s20", "s21");                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();           \
        for (int turn = 0; turn < TURNS; ++turn) {                                                                   \
            _Pragma("unroll") for (int rep = 0; rep < UNROLL / 16; ++rep) {                                          \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                     \
                    asm volatile(ASM : "+v"(a[i]), "+v"(p[i]), "+v"(u[i]) : "v"(m), "v"(c), "v"(pm), "v"(um) : "vcc", "scc", "s20", "s21"); \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();           \
        float s = 0.0f;                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];                    \
        if (s == 12345.678f) sink[0] = s;                                                                            \
        if ((threadIdx.x & 63) == 0) {                                                                               \
            const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);                                                  \
            stamps[2 * w] = t1 - t0;                                                                                 \
            stamps[2 * w + 1] = r1 - r0;                                                                             \
        }                                                                                                            \
    }

// operands: %0 a (float), %1 p (pair), %2 u (unsigned), %3 m, %4 c, %5 pm, %6 um
ISSUE_KERNEL(k_fma, "v_fma_f32 %0, %0, %3, %4")
ISSUE_KERNEL(k_add, "v_add_f32 %0, %0, %4")
ISSUE_KERNEL(k_mul, "v_mul_f32 %0, %0, %3")
ISSUE_KERNEL(k_max, "v_max_f32 %0, %0, %4")
ISSUE_KERNEL(k_max3, "v_max3_f32 %0, %0, %3, %4")
ISSUE_KERNEL(k_min3, "v_min3_f32 %0, %0, %3, %4")
ISSUE_KERNEL(k_pk_mul, "v_pk_mul_f32 %1, %1, %5")
ISSUE_KERNEL(k_pk_add, "v_pk_add_f32 %1, %1, %5")
ISSUE_KERNEL(k_pk_fma, "v_pk_fma_f32 %1, %1, %5, %5")
ISSUE_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %3")
ISSUE_KERNEL(k_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %3")
ISSUE_KERNEL(k_min_u32, "v_min_u32 %2, %2, %6")
ISSUE_KERNEL(k_add_u32, "v_add_u32 %2, %2, %6")
ISSUE_KERNEL(k_and, "v_and_b32 %2, %2, %6")
ISSUE_KERNEL(k_xor, "v_xor_b32 %2, %2, %6")
ISSUE_KERNEL(k_lshl_or, "v_lshl_or_b32 %2, %2, 4, %6")
ISSUE_KERNEL(k_lshl_add, "v_lshl_add_u32 %2, %2, 1, %6")
ISSUE_KERNEL(k_bfe, "v_bfe_u32 %2, %2, 3, 29")
ISSUE_KERNEL(k_mov, "v_mov_b32 %0, %3")
ISSUE_KERNEL(k_mul_lo, "v_mul_lo_u32 %2, %2, %6")
ISSUE_KERNEL(k_mad_u64, "v_mul_hi_u32 %2, %2, %6")
ISSUE_KERNEL(k_rcp, "v_rcp_f32 %0, %0")
ISSUE_KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
ISSUE_KERNEL(k_cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %2")
ISSUE_KERNEL(k_cvt_ubyte3, "v_cvt_f32_ubyte3 %0, %2")
ISSUE_KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %2")
ISSUE_KERNEL(k_cvt_f16_hi, "v_cvt_f32_f16_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
ISSUE_KERNEL(k_cvt_u32, "v_cvt_f32_u32 %0, %2")
ISSUE_KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %3, vcc")
ISSUE_KERNEL(k_perm, "v_perm_b32 %2, %2, %6, %6")
ISSUE_KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
ISSUE_KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %2, %3, %4 op_sel_hi:[1,0,0]")
ISSUE_KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %3, s[20:21]")
ISSUE_KERNEL(k_cndmask_nodep, "v_cndmask_b32 %0, %4, %3, vcc")
ISSUE_KERNEL(k_cndmask_imm, "v_cndmask_b32 %2, -1, %2, vcc")
ISSUE_KERNEL(k_sub, "v_sub_f32 %0, %0, %4")
ISSUE_KERNEL(k_ashr, "v_ashrrev_i32 %2, 31, %2")
ISSUE_KERNEL(k_or, "v_or_b32 %2, %2, %6")
ISSUE_KERNEL(k_and_or, "v_and_or_b32 %2, %2, %6, %6")
ISSUE_KERNEL(k_min_f32, "v_min_f32 %0, %0, %4")
ISSUE_KERNEL(k_fmac, "v_fmac_f32 %0, %3, %4")
ISSUE_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %2, %2, %6, %6")
ISSUE_KERNEL(k_sub_u32, "v_sub_u32 %2, %2, %6")
ISSUE_KERNEL(k_lshrrev, "v_lshrrev_b32 %2, 1, %2")
ISSUE_KERNEL(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %3, vcc")
ISSUE_KERNEL(k_cmp_cndmask_vcc, "v_cmp_lt_u32 vcc, %2, %6\n\tv_cndmask_b32 %0, %0, %3, vcc")
ISSUE_KERNEL(k_cmp_cndmask_sgpr, "v_cmp_lt_u32 s[20:21], %2, %6\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]")
ISSUE_KERNEL(k_cmp_cndmask_e64_vcc, "v_cmp_lt_u32 vcc, %2, %6\n\tv_cndmask_b32_e64 %0, %0, %3, vcc")
ISSUE_KERNEL(k_cmp_2cndmask_vcc, "v_cmp_lt_u32 vcc, %2, %6\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %2, %2, %6, vcc")
ISSUE_KERNEL(k_cmp_2cndmask_sgpr, "v_cmp_lt_u32 s[20:21], %2, %6\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %6, s[20:21]")
ISSUE_KERNEL(k_sort_unit_vcc, "v_cmp_lt_u32 vcc, %2, %6\n\tv_min_u32 %2, %2, %6\n\tv_max_u32 %2, %2, %6\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %0, %3, %0, vcc")
ISSUE_KERNEL(k_sort_unit_sgpr, "v_cmp_lt_u32 s[20:21], %2, %6\n\tv_min_u32 %2, %2, %6\n\tv_max_u32 %2, %2, %6\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]\n\tv_cndmask_b32_e64 %0, %3, %0, s[20:21]")
ISSUE_KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %3, %0")
ISSUE_KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %0, %3, %4")
ISSUE_KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %0, %3, %4")
ISSUE_KERNEL(k_addc_vcc, "v_addc_co_u32 %2, vcc, %2, %6, vcc")
ISSUE_KERNEL(k_addc_sgpr, "v_addc_co_u32_e64 %2, s[20:21], %2, %6, s[20:21]")
ISSUE_KERNEL(k_add_co, "v_add_co_u32 %2, vcc, %2, %6")
ISSUE_KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %1, s[20:21], %2, %6, %1")
ISSUE_KERNEL(k_cmp_class, "v_cmp_class_f32 vcc, %0, %6")
ISSUE_KERNEL(k_readfirstlane, "v_readfirstlane_b32 s20, %0")
ISSUE_KERNEL(k_log, "v_log_f32 %0, %0")
ISSUE_KERNEL(k_exp, "v_exp_f32 %0, %0")
ISSUE_KERNEL(k_rsq, "v_rsq_f32 %0, %0")
ISSUE_KERNEL(k_mul_f64, "v_mul_f64 %1, %1, %1")
ISSUE_KERNEL(k_lshl_b64, "v_lshlrev_b64 %1, 1, %1")
ISSUE_KERNEL(k_bfi, "v_bfi_b32 %2, %6, %2, %6")
ISSUE_KERNEL(k_alignbit, "v_alignbit_b32 %2, %2, %6, 5")
ISSUE_KERNEL(k_med3, "v_med3_f32 %0, %0, %3, %4")
ISSUE_KERNEL(k_s_mov, "s_mov_b32 s20, 0")
ISSUE_KERNEL(k_s_and_b64, "s_and_b64 s[20:21], s[20:21], exec")
ISSUE_KERNEL(k_s_add, "s_add_u32 s20, s20, 1")
ISSUE_KERNEL(k_s_bcnt, "s_bcnt1_i32_b64 s20, exec")
ISSUE_KERNEL(k_s_nop, "s_nop 0")
ISSUE_KERNEL(k_saveexec, "s_and_saveexec_b64 s[20:21], exec\n\ts_or_b64 exec, exec, s[20:21]")
ISSUE_KERNEL(k_valu_salu, "v_add_f32 %0, %0, %4\n\ts_and_b64 s[20:21], s[20:21], exec")
ISSUE_KERNEL(k_mix, "v_fma_f32 %0, %0, %3, %4\n\tv_pk_mul_f32 %1, %1, %5\n\tv_min_u32 %2, %2, %6")

typedef void (*kernel_t)(unsigned long long*, float*, float);

void run(const char* name, kernel_t kernel, int per_asm, int cus, unsigned long long* d_stamps, float* d_sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("%-16s", name);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = cus * waves_per_simd;
        kernel<<<blocks, 256>>>(d_stamps, d_sink, 1.0f);  // warm-up (code load, clocks)
        (void)hipEventRecord(e0);
        kernel<<<blocks, 256>>>(d_stamps, d_sink, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * (size_t)blocks * 4);
        (void)hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, mhz;
        for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
            cyc.push_back((double)h[2 * w]);
            mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(mhz.begin(), mhz.end());
        const double med_mhz = mhz[mhz.size() / 2];
        const double inst = (double)TURNS * UNROLL * per_asm;
        // cycles of SIMD time per wave-instruction: the launch's wall time in shader cycles, divided by the instructions
        // each SIMD was given (waves_per_simd x instructions per wave)
        const double cycles_per_inst = ms * 1e-3 * med_mhz * 1e6 / (waves_per_simd * inst);
        printf("  %dw: %5.2f cyc/inst (%4.0f MHz, wave %4.2f)", waves_per_simd, cycles_per_inst, med_mhz, cyc[cyc.size() / 2] / inst);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_stamps;
    float* d_sink;
    (void)hipMalloc(&d_stamps, 2 * (size_t)cus * 8 * 4 * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 16);
    printf("%s, %d CUs.  Per instruction and number of waves per SIMD (W blocks of 256 threads per CU): shader cycles of SIMD time per\n"
           "wave-instruction by the launch's wall time, the clock (s_memtime / s_memrealtime), and the cycles one wave took per instruction.\n",
           prop.name, cus);
#define RUN(K, N) run(#K, K, N, cus, d_stamps, d_sink)
    RUN(k_fma, 1); RUN(k_add, 1); RUN(k_mul, 1); RUN(k_max, 1); RUN(k_max3, 1); RUN(k_min3, 1);
    RUN(k_pk_mul, 1); RUN(k_pk_add, 1); RUN(k_pk_fma, 1);
    RUN(k_cmp, 1); RUN(k_cmp_sgpr, 1); RUN(k_min_u32, 1); RUN(k_add_u32, 1); RUN(k_and, 1); RUN(k_xor, 1);
    RUN(k_lshl_or, 1); RUN(k_lshl_add, 1); RUN(k_bfe, 1); RUN(k_mov, 1); RUN(k_mul_lo, 1); RUN(k_mad_u64, 1); RUN(k_rcp, 1); RUN(k_sqrt, 1);
    RUN(k_cvt_ubyte0, 1); RUN(k_cvt_ubyte3, 1); RUN(k_cvt_f16, 1); RUN(k_cvt_f16_hi, 1); RUN(k_cvt_u32, 1); RUN(k_cndmask_vcc, 1);
    RUN(k_perm, 1); RUN(k_mov_dpp, 1); RUN(k_fma_mix, 1);
    RUN(k_cndmask_sgpr, 1); RUN(k_cndmask_nodep, 1); RUN(k_cndmask_imm, 1); RUN(k_sub, 1); RUN(k_ashr, 1); RUN(k_or, 1); RUN(k_and_or, 1);
    RUN(k_min_f32, 1); RUN(k_fmac, 1); RUN(k_mad_u32_u24, 1); RUN(k_sub_u32, 1); RUN(k_lshrrev, 1);
    RUN(k_cndmask_e64_vcc, 1); RUN(k_cmp_cndmask_vcc, 2); RUN(k_cmp_cndmask_sgpr, 2); RUN(k_cmp_cndmask_e64_vcc, 2);
    RUN(k_cmp_2cndmask_vcc, 3); RUN(k_cmp_2cndmask_sgpr, 3); RUN(k_sort_unit_vcc, 5); RUN(k_sort_unit_sgpr, 5);
    RUN(k_div_scale, 1); RUN(k_div_fmas, 1); RUN(k_div_fixup, 1); RUN(k_addc_vcc, 1); RUN(k_addc_sgpr, 1); RUN(k_add_co, 1); RUN(k_mad_u64_u32, 1);
    RUN(k_cmp_class, 1); RUN(k_readfirstlane, 1); RUN(k_log, 1); RUN(k_exp, 1); RUN(k_rsq, 1); RUN(k_mul_f64, 1); RUN(k_lshl_b64, 1);
    RUN(k_bfi, 1); RUN(k_alignbit, 1); RUN(k_med3, 1);
    RUN(k_s_mov, 1); RUN(k_s_and_b64, 1); RUN(k_s_add, 1); RUN(k_s_bcnt, 1); RUN(k_s_nop, 1); RUN(k_saveexec, 2); RUN(k_valu_salu, 2);
    RUN(k_mix, 3);
    return 0;
}
