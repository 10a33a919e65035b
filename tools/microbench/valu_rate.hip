// Measures vector-ALU issue throughput per SIMD on the current GPU for a few instruction kinds, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void spin(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x + 1, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned su0 = (unsigned)iters, su1 = blockIdx.x;
    unsigned long long sm = 0;
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
            } else if (KIND == 5) { // 8 vector + 8 scalar ALU instructions (the scalar ones on wave-uniform values)
                a0 = a0 * seed; a1 = a1 * seed; a2 = a2 * seed; a3 = a3 * seed; a4 = a4 + seed; a5 = a5 + seed; a6 = a6 + seed; a7 = a7 + seed;
                asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, 3\n s_xor_b32 %1, %1, %0\n"
                             "s_add_u32 %0, %0, 5\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, 7\n s_xor_b32 %1, %1, %0" : "+s"(su0), "+s"(su1) : : "scc");
            } else if (KIND == 6) { // 8 vector instructions + 4 exec-mask save/restore pairs as around divergent ifs
                a0 = a0 * seed; a1 = a1 * seed; a2 = a2 * seed; a3 = a3 * seed; a4 = a4 + seed; a5 = a5 + seed; a6 = a6 + seed; a7 = a7 + seed;
                asm volatile("s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0\n"
                             "s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0" : "+s"(sm) : : "scc");
            } else if (KIND == 4) { // v_cndmask / v_min / v_max mix
                a0 = fminf(a0, a1); a1 = fmaxf(a1, a2); a2 = fminf(a2, a3); a3 = fmaxf(a3, a4); a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = fminf(a6, a7); a7 = fmaxf(a7, a0);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3) + (float)(su0 ^ su1) + (float)(sm & 1ull);
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
    setvbuf(stdout, nullptr, _IOLBF, 0);      // a row per line as it is measured
    int cus = 0, clk = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    float* d; hipMalloc(&d, sizeof(float) * 64 * cus * 4 * 8);
    const int iters = 20000;
    const char* names[] = {"mul/add independent", "mul/add dependent", "v_mul_lo_u32", "v_rcp_f32", "min/max chain", "8 VALU + 8 SALU", "8 VALU + 8 exec ops"};
    printf("CUs %d, clock %d kHz; cycles per wave64 VECTOR instruction per SIMD (assuming the nominal clock; the last two rows carry as many scalar instructions again)\n", cus, clk);
    for (int kind = 0; kind < 7; ++kind) {
        printf("%-22s", names[kind]);
        for (int w : {1, 2, 4, 5, 8}) {
            double ms = kind == 0 ? run<0>(w, cus, d, iters) : kind == 1 ? run<1>(w, cus, d, iters) : kind == 2 ? run<2>(w, cus, d, iters)
                        : kind == 3 ? run<3>(w, cus, d, iters) : kind == 4 ? run<4>(w, cus, d, iters) : kind == 5 ? run<5>(w, cus, d, iters) : run<6>(w, cus, d, iters);
            double instr = (double)iters * 64.0 * w;          // per SIMD
            printf("  w%d: %5.2f", w, ms * 1e-3 * clk * 1e3 / instr); fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
