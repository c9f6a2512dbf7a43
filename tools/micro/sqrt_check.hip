// Is the device's double-precision sqrt (and division) the correctly rounded one?  Ten million random arguments (all exponents, and mantissas next to perfect
// squares) against the host's IEEE sqrt / division, bit for bit.  Round 6: decides whether bioik_acos.h may use sqrt() as a primitive.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(const double* a, const double* b, double* s, double* d, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) s[i] = sqrt(a[i]), d[i] = a[i] / b[i];
}
int main() {
    const size_t n = 10000000;
    std::vector<double> a(n), b(n), s(n), d(n);
    unsigned long long x = 88172645463325252ull;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < n; i++) {
        unsigned long long m = next(), e = next();
        unsigned long long bits;
        if (i % 4 == 0) { double r = (double)(m >> 12) * 0x1p-20; double sq = r * r; memcpy(&bits, &sq, 8); bits += (long long)(e % 5) - 2; }  // next to perfect squares
        else bits = (m & 0x000fffffffffffffull) | ((1ull + e % 2045ull) << 52);  // every normal exponent
        memcpy(&a[i], &bits, 8);
        bits = (next() & 0x000fffffffffffffull) | ((900ull + next() % 250ull) << 52);
        memcpy(&b[i], &bits, 8);
    }
    double *da, *db, *ds, *dd;
    CHECK(hipMalloc(&da, n * 8)); CHECK(hipMalloc(&db, n * 8)); CHECK(hipMalloc(&ds, n * 8)); CHECK(hipMalloc(&dd, n * 8));
    CHECK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, ds, dd, n);
    CHECK(hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(d.data(), dd, n * 8, hipMemcpyDeviceToHost));
    size_t bad_s = 0, bad_d = 0;
    for (size_t i = 0; i < n; i++) {
        const double hs = std::sqrt(a[i]), hd = a[i] / b[i];
        if (memcmp(&hs, &s[i], 8)) { if (bad_s++ < 5) printf("sqrt(%a): host %a device %a\n", a[i], hs, s[i]); }
        if (memcmp(&hd, &d[i], 8)) { if (bad_d++ < 5) printf("%a / %a: host %a device %a\n", a[i], b[i], hd, d[i]); }
    }
    printf("%zu arguments: sqrt differs in %zu, division in %zu\n", n, bad_s, bad_d);
    return 0;
}
