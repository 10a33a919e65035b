// Measures vector-ALU issue throughput per SIMD on the current GPU for a few instruction kinds, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void spin(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x + 1, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) {        // v_mul_f32 / v_add_f32, 8 independent chains
                a0 = a0 * seed; a1 = a1 * seed; a2 = a2 * seed; a3 = a3 * seed; a4 = a4 + seed; a5 = a5 + seed; a6 = a6 + seed; a7 = a7 + seed;
            } else if (KIND == 1) { // one dependent chain
                a0 = a0 * seed; a0 = a0 + seed; a0 = a0 * seed; a0 = a0 + seed; a0 = a0 * seed; a0 = a0 + seed; a0 = a0 * seed; a0 = a0 + seed;
            } else if (KIND == 2) { // v_mul_lo_u32
                u0 *= u1; u1 *= u2; u2 *= u3; u3 *= u0; u0 *= u2; u1 *= u3; u2 *= u0; u3 *= u1;
            } else if (KIND == 3) { // v_rcp_f32
                a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
            } else if (KIND == 4) { // v_cndmask / v_min / v_max mix
                a0 = fminf(a0, a1); a1 = fmaxf(a1, a2); a2 = fminf(a2, a3); a3 = fmaxf(a3, a4); a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = fminf(a6, a7); a7 = fmaxf(a7, a0);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3);
}

template <int KIND>
double run(int wavesPerSimd, int cus, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = cus * 4 * wavesPerSimd;
    hipLaunchKernelGGL(spin<KIND>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<KIND>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    int cus = 0, clk = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    float* d; hipMalloc(&d, sizeof(float) * 64 * cus * 4 * 8);
    const int iters = 20000;
    const char* names[] = {"mul/add independent", "mul/add dependent", "v_mul_lo_u32", "v_rcp_f32", "min/max chain"};
    printf("CUs %d, clock %d kHz; cycles per wave64 instruction per SIMD (assuming the nominal clock)\n", cus, clk);
    for (int kind = 0; kind < 5; ++kind) {
        printf("%-22s", names[kind]);
        for (int w : {1, 2, 4, 5, 8}) {
            double ms = kind == 0 ? run<0>(w, cus, d, iters) : kind == 1 ? run<1>(w, cus, d, iters) : kind == 2 ? run<2>(w, cus, d, iters)
                        : kind == 3 ? run<3>(w, cus, d, iters) : run<4>(w, cus, d, iters);
            double instr = (double)iters * 64.0 * w;          // per SIMD
            printf("  w%d: %5.2f", w, ms * 1e-3 * clk * 1e3 / instr);
        }
        printf("\n");
    }
    return 0;
}
