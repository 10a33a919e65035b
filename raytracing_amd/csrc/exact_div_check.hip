// Exhaustive comparison, over all 2^32 float bit patterns, of the short quotient sequences of exact_div.h with the
// compiler's correctly rounded fp32 division (same flags as the kernels: denormals flushed, no contraction).
// Built by the Makefile into ../exact_div_check; tests/test_gpu_exact_div.py runs it and reads the report:
// per candidate the number of inputs whose result differs in any bit (NaN results compare equal to NaN), and the
// runs of input exponents (sign, biased exponent) that hold the differences.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "exact_div.h"

using namespace rtd;
#if !RT_SHORT_DIV
#error "the checker compares the short sequences with the plain operators: build it with RT_SHORT_DIV=1"
#endif

// A candidate that is NOT used by the kernels, kept to show why: a short square root is exact only for x >= 2^-102.
__device__ __forceinline__ float sqrt_short(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    float s1 = __builtin_fmaf(e, h, s);
    return (x == 0.0f || x == INFINITY) ? x : s1;
}

__device__ __forceinline__ bool same(float a, float b) {
    return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
}

struct Report { unsigned long long bad; uint32_t first[8]; unsigned int byExp[512]; };

__device__ void note(Report* rep, uint32_t bits) {
    unsigned long long k = atomicAdd(&rep->bad, 1ull);
    if (k < 8) rep->first[k] = bits;
    atomicAdd(&rep->byExp[bits >> 23], 1u);       // sign and exponent of the input
}

// mode 0: 1/x   1, 2: the short square root without its fallback (not used)   3: x / c   4: 0.5/x
// 5: sqrt(x)   6: 1/sqrt(x) as two roundings
__global__ void check(int mode, float c, float rc, Report* rep) {
    const uint64_t n = 1ull << 32, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x = __uint_as_float((uint32_t)i);
        float ref, got;
        if (mode == 0) { ref = 1.0f / x; got = rcp_exact(x); }
        else if (mode == 1) { ref = sqrtf(x); got = sqrt_short(x); }
        else if (mode == 2) { ref = 1.0f / sqrtf(x); got = rcp_exact(sqrt_short(x)); }
        else if (mode == 4) { ref = 0.5f / x; got = 0.5f * rcp_exact(x); }
        else if (mode == 5) { ref = sqrtf(x); got = sqrt_exact(x); }
        else if (mode == 6) { ref = 1.0f / sqrtf(x); got = rcp_exact(sqrt_exact(x)); }
        else { ref = x / c; got = div_by_uniform(x, c, rc); }
        if (!same(ref, got)) note(rep, (uint32_t)i);
    }
}

// prints the runs of input exponents (sign, biased exponent) that hold mismatches
static void ranges(const Report& h) {
    for (int sgn = 0; sgn < 2; ++sgn) {
        int e = 0;
        while (e < 256) {
            if (!h.byExp[sgn * 256 + e]) { ++e; continue; }
            int e0 = e; unsigned long long n = 0;
            while (e < 256 && h.byExp[sgn * 256 + e]) { n += h.byExp[sgn * 256 + e]; ++e; }
            printf("      %c exponent %d..%d: %llu\n", sgn ? '-' : '+', e0, e - 1, n);
        }
    }
}

int main() {
    Report* d; Report h;
    if (hipMalloc(&d, sizeof(Report)) != hipSuccess) { printf("no device\n"); return 1; }
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const char* names[7] = {"1/x", "sqrt_short(x)[unused]", "1/sqrt_short(x)[unused]", "", "0.5/x", "sqrt(x)", "1/sqrt(x)"};
    for (int mode = 0; mode < 7; ++mode) {
        if (mode == 3) continue;
        hipMemset(d, 0, sizeof(Report));
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, mode, 1.0f, 1.0f, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(&h, d, sizeof(Report), hipMemcpyDeviceToHost);
        printf("%-18s differing inputs: %llu", names[mode], h.bad);
        for (unsigned k = 0; k < 8 && k < h.bad; ++k) printf(" %08x", h.first[k]);
        printf("\n");
        ranges(h);
    }
    const float cs[] = {600.f, 400.f, 3840.f, 2160.f, 1920.f, 1080.f, 640.f, 360.f, 1280.f, 720.f, 64.f, 48.f, 37.f, 23.f, 1.f, 3.f, 7.f, 1000.f, 4096.f, 32768.f, 12345.f, 999.f};
    for (float c : cs) {
        hipMemset(d, 0, sizeof(Report));
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, 3, c, 1.0f / c, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(&h, d, sizeof(Report), hipMemcpyDeviceToHost);
        printf("x/%-8g differing inputs: %llu", c, h.bad);
        for (unsigned k = 0; k < 8 && k < h.bad; ++k) printf(" %08x", h.first[k]);
        printf("\n");
        ranges(h);
    }
    hipFree(d);
    return 0;
}
