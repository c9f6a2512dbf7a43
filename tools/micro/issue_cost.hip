// Issue cost of single gfx950 vector instructions: four independent dependency chains per lane, so that what is measured is the slot an instruction takes
// on its SIMD, not its latency -- for a lone wavefront and for four wavefronts per SIMD on every CU.  Round 6: what a gene draw may cost (32-bit multiplies,
// byte sums, conversions) before the counter hash and the Gaussian were redesigned.   hipcc --offload-arch=gfx950 -O3 issue_cost.hip -o issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 1024;

// 32-bit result, operands (x, a): x <- op(x, a)
#define KERNEL32(NAME, ASM)                                                                                          \
    __global__ void k_##NAME(unsigned* out, unsigned long long* t, unsigned a) {                                     \
        unsigned x0 = out[threadIdx.x] + threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3;             \
        unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();                                   \
        for (int i = 0; i < N; i++) {                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                          \
                asm volatile(ASM : "+v"(x0) : "v"(a));                                                               \
                asm volatile(ASM : "+v"(x1) : "v"(a));                                                               \
                asm volatile(ASM : "+v"(x2) : "v"(a));                                                               \
                asm volatile(ASM : "+v"(x3) : "v"(a));                                                               \
            }                                                                                                        \
        }                                                                                                            \
        unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();                                   \
        out[threadIdx.x + blockIdx.x * blockDim.x] = x0 + x1 + x2 + x3;                                              \
        if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;                                     \
    }
KERNEL32(xor, "v_xor_b32 %0, %0, %1")
KERNEL32(lshr, "v_lshrrev_b32 %0, 13, %0")
KERNEL32(add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL32(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(alignbit, "v_alignbit_b32 %0, %0, %0, 13")
KERNEL32(bfe, "v_bfe_u32 %0, %0, 3, 29")
KERNEL32(mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
KERNEL32(sad_u8, "v_sad_u8 %0, %0, %1, %1")
KERNEL32(dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %1")
KERNEL32(bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL32(mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
// 64-bit result in a register pair
#define KERNEL64(NAME, ASM)                                                                                          \
    __global__ void k_##NAME(unsigned* out, unsigned long long* t, unsigned a) {                                     \
        unsigned long long x0 = out[threadIdx.x] + threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3;   \
        double d = (double)a;                                                                                        \
        unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();                                   \
        for (int i = 0; i < N; i++) {                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                          \
                asm volatile(ASM : "+v"(x0) : "v"(a), "v"(d));                                                       \
                asm volatile(ASM : "+v"(x1) : "v"(a), "v"(d));                                                       \
                asm volatile(ASM : "+v"(x2) : "v"(a), "v"(d));                                                       \
                asm volatile(ASM : "+v"(x3) : "v"(a), "v"(d));                                                       \
            }                                                                                                        \
        }                                                                                                            \
        unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();                                   \
        out[threadIdx.x + blockIdx.x * blockDim.x] = (unsigned)(x0 + x1 + x2 + x3);                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;                                     \
    }
KERNEL64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
KERNEL64(cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
KERNEL64(mul_f64, "v_mul_f64 %0, %0, %2")
KERNEL64(fma_f64, "v_fma_f64 %0, %0, %2, %2")
KERNEL64(add_f64, "v_add_f64 %0, %0, %2")
KERNEL64(max_f64, "v_max_f64 %0, %0, %2")
KERNEL64(lshr_b64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(mov_b64, "v_mov_b64 %0, %2")
KERNEL64(pk_mov, "v_pk_mov_b32 %0, %0, %2")

int main() {
    unsigned* out;
    unsigned long long* t;
    CHECK(hipMalloc(&out, 1 << 26));
    CHECK(hipMemset(out, 0, 1 << 26));
    CHECK(hipMalloc(&t, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double fma_ms[3] = {0, 0, 0};
    // whole-kernel time (events) of a grid that puts 1 / 4 / 8 wavefronts on every SIMD of 256 CUs: what a SIMD retires per clock when several wavefronts offer it
    // independent instructions -- the issue slot an instruction really takes (the per-wavefront cycle counter of one workgroup does not see its neighbours)
    const int blocks[3] = {256, 1024, 2048};
    printf("%-14s %28s %28s %28s\n", "instruction", "1 wavefront / SIMD", "4 wavefronts / SIMD", "8 wavefronts / SIMD");
#define RUN(NAME)                                                                                   \
    {                                                                                               \
        printf("%-14s", #NAME);                                                                     \
        for (int s = 0; s < 3; s++) {                                                               \
            float best = 1e30f;                                                                     \
            for (int rep = 0; rep < 4; rep++) {                                                     \
                CHECK(hipEventRecord(e0, 0));                                                       \
                hipLaunchKernelGGL(k_##NAME, dim3(blocks[s]), dim3(256), 0, 0, out, t, 0xD256D193u); \
                CHECK(hipEventRecord(e1, 0));                                                       \
                CHECK(hipEventSynchronize(e1));                                                     \
                float ms;                                                                           \
                CHECK(hipEventElapsedTime(&ms, e0, e1));                                            \
                if (rep && ms < best) best = ms;                                                    \
            }                                                                                       \
            if (fma_ms[s] == 0) fma_ms[s] = best;                                                   \
            const double waves = blocks[s] / 256.0; /* per SIMD */                                  \
            printf("   %8.1f us  %5.2f x fma_f64", best * 1e3, best / fma_ms[s]);                   \
            (void)waves;                                                                            \
        }                                                                                           \
        printf("\n");                                                                               \
    }
    RUN(fma_f64) RUN(mul_f64) RUN(add_f64) RUN(max_f64) RUN(cvt_f64_i32) RUN(cvt_f64_u32) RUN(mov_b64) RUN(pk_mov) RUN(lshr_b64)
    RUN(xor) RUN(lshr) RUN(add3) RUN(lshl_add) RUN(alignbit) RUN(bfe) RUN(mul_lo) RUN(mul_hi) RUN(mul_u24) RUN(mad_u24) RUN(sad_u8) RUN(dot4_u8) RUN(bcnt) RUN(perm)
    RUN(mov_dpp) RUN(cndmask) RUN(mad_u64_u32)
    return 0;
}
