// Microbenchmarks of the gfx950 costs that bound k_solve: dependent / independent FP64 VALU issue, 32-bit integer multiply,
// LDS and scalar-load latency, for a lone wavefront and for 1..8 wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 2048;

__global__ void k_fma_dep(double* out, unsigned long long* t, double a, double b) {
    double x = out[threadIdx.x];
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = __builtin_fma(x, a, b);
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_fma_ind4(double* out, unsigned long long* t, double a, double b) {
    double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x0 = __builtin_fma(x0, a, b), x1 = __builtin_fma(x1, a, b), x2 = __builtin_fma(x2, a, b), x3 = __builtin_fma(x3, a, b);
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_muladd_dep(double* out, unsigned long long* t, double a, double b) {
    double x = out[threadIdx.x];
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            double m;
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m) : "v"(x), "v"(a));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(x) : "v"(m), "v"(b));
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_mulhi_dep(double* out, unsigned long long* t, unsigned a) {
    unsigned x = (unsigned)out[threadIdx.x] + threadIdx.x;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = __umulhi(x, a) ^ 0x9E3779B9u;
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_mulhi_ind4(double* out, unsigned long long* t, unsigned a) {
    unsigned x0 = (unsigned)out[threadIdx.x] + threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x0 = __umulhi(x0, a) ^ 0x9E3779B9u, x1 = __umulhi(x1, a) ^ 0x9E3779B9u, x2 = __umulhi(x2, a) ^ 0x9E3779B9u, x3 = __umulhi(x3, a) ^ 0x9E3779B9u;
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
// one Philox round both ways: 64-bit product as v_mad_u64_u32, or as v_mul_hi_u32 + v_mul_lo_u32
__global__ void k_philox_mad64(double* out, unsigned long long* t, unsigned key) {
    unsigned c0 = (unsigned)out[threadIdx.x] + threadIdx.x, c1 = c0 * 7u + 1u;
    unsigned long long t0 = wall_clock64(), cc0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            unsigned long long p = (unsigned long long)0xD256D193u * (unsigned long long)c0;
            unsigned hi = (unsigned)(p >> 32), lo = (unsigned)p;
            c0 = hi ^ key ^ c1;
            c1 = lo;
        }
    }
    unsigned long long cc1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = c0 + c1;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = cc1 - cc0;
}
__global__ void k_philox_hilo(double* out, unsigned long long* t, unsigned key) {
    unsigned c0 = (unsigned)out[threadIdx.x] + threadIdx.x, c1 = c0 * 7u + 1u;
    unsigned long long t0 = wall_clock64(), cc0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            unsigned hi, lo;
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi) : "v"(c0), "v"(0xD256D193u));
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(c0), "v"(0xD256D193u));
            c0 = hi ^ key ^ c1;
            c1 = lo;
        }
    }
    unsigned long long cc1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = c0 + c1;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = cc1 - cc0;
}
__global__ void k_lds_chase(double* out, unsigned long long* t) {
    __shared__ int next[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) next[i] = (i * 17 + 64) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) p = next[p];
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_sload_chase(double* out, unsigned long long* t, const int __attribute__((address_space(4)))* tab) {
    int p = 0;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) p = tab[p];
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_div_dep(double* out, unsigned long long* t, double a) {
    double x = out[threadIdx.x] + 3.0;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = a / x + 1.0;
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}
__global__ void k_sqrt_dep(double* out, unsigned long long* t, double a) {
    double x = out[threadIdx.x] + 3.0;
    unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = sqrt(x) + a;
    }
    unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0, t[1] = c1 - c0;
}

int main() {
    double* out;
    unsigned long long* t;
    int* tab;
    CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipMemset(out, 0, 1 << 24));
    CHECK(hipMalloc(&t, 64));
    std::vector<int> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (i * 17 + 64) & 4095;
    CHECK(hipMalloc(&tab, 4096 * 4));
    CHECK(hipMemcpy(tab, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    auto report = [&](const char* name, int ops, int blocks, int threads) {
        CHECK(hipDeviceSynchronize());
        unsigned long long r[2];
        CHECK(hipMemcpy(r, t, 16, hipMemcpyDeviceToHost));
        double ns = r[0] * 10.0;
        printf("%-14s blocks %5d x %4d thr: %7.2f ns/op  %7.2f shader-clk/op (memtime ratio %.3f clk/ns)\n", name, blocks, threads, ns / ops, (double)r[1] / ops, r[1] / ns);
    };
    // lone wave, then 1..8 waves per SIMD on every CU (256 CUs x 4 SIMDs; a block of 256 threads = one wave per SIMD)
    int shapes[][2] = {{1, 64}, {256, 256}, {512, 256}, {768, 256}, {1024, 256}, {2048, 256}};
    for (auto& s : shapes) {
        int b = s[0], th = s[1];
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_fma_dep, dim3(b), dim3(th), 0, 0, out, t, 1.0000001, 1e-9);
            if (rep) report("fma_dep", N * 16, b, th);
            hipLaunchKernelGGL(k_fma_ind4, dim3(b), dim3(th), 0, 0, out, t, 1.0000001, 1e-9);
            if (rep) report("fma_ind4", N * 16, b, th);
            hipLaunchKernelGGL(k_muladd_dep, dim3(b), dim3(th), 0, 0, out, t, 1.0000001, 1e-9);
            if (rep) report("mul+add_dep", N * 16, b, th);
            hipLaunchKernelGGL(k_mulhi_dep, dim3(b), dim3(th), 0, 0, out, t, 0xD256D193u);
            if (rep) report("mulhi_dep", N * 16, b, th);
            hipLaunchKernelGGL(k_mulhi_ind4, dim3(b), dim3(th), 0, 0, out, t, 0xD256D193u);
            if (rep) report("mulhi_ind4", N * 16, b, th);
            hipLaunchKernelGGL(k_philox_mad64, dim3(b), dim3(th), 0, 0, out, t, 0x9E3779B9u);
            if (rep) report("philox_mad64", N * 16, b, th);
            hipLaunchKernelGGL(k_philox_hilo, dim3(b), dim3(th), 0, 0, out, t, 0x9E3779B9u);
            if (rep) report("philox_hilo", N * 16, b, th);
            hipLaunchKernelGGL(k_div_dep, dim3(b), dim3(th), 0, 0, out, t, 1.7);
            if (rep) report("div_dep", N * 16, b, th);
            hipLaunchKernelGGL(k_sqrt_dep, dim3(b), dim3(th), 0, 0, out, t, 1.7);
            if (rep) report("sqrt_dep", N * 16, b, th);
            if (b <= 256) {
                hipLaunchKernelGGL(k_lds_chase, dim3(b), dim3(th), 0, 0, out, t);
                if (rep) report("lds_chase", N * 16, b, th);
                hipLaunchKernelGGL(k_sload_chase, dim3(b), dim3(th), 0, 0, out, t, (const int __attribute__((address_space(4)))*)tab);
                if (rep) report("sload_chase", N * 16, b, th);
            }
        }
    }
    return 0;
}
