// How the vector L1 prices scattered 64-byte record fetches on the current GPU.
//   own:   every lane reads its own 64-byte record with 4 x dwordx4            (4 accesses per lane and record)
//   quad:  the 4 lanes of a quad read one record together, 16 bytes each,
//          4 instructions cover the quad's 4 records                             (1 access per quad and record, if merged)
// Records are chosen pseudo-randomly from a table that fits in L2 (18 MB), like BVH nodes.
// Build: hipcc --offload-arch=gfx950 -O3 -o l1_gather l1_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(64) void gather(const float4* __restrict__ table, unsigned nRec, int iters, float* out) {
    unsigned s = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    const unsigned lane = threadIdx.x, q = lane & 3u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        const unsigned rec = rnd(s) % nRec;              // this lane's record
        if (MODE == 0) {
            const float4* p = table + (size_t)rec * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else {
            // record of quad member r, broadcast inside the quad; lane q reads chunk q of it
            float sum = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned owner = __shfl(rec, (int)((lane & ~3u) + r), 64);
                float4 v = table[(size_t)owner * 4 + q];
                sum += v.x + v.y + v.z + v.w;
            }
            acc += sum;
        }
        s ^= __float_as_uint(acc) & 1u;                  // keep the loads dependent on their results
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const unsigned maxRec = 18u * 1024 * 1024 / 64;
    float4* table; hipMalloc(&table, (size_t)maxRec * 64); hipMemset(table, 0, (size_t)maxRec * 64);
    float* out; hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
    const int iters = 4000, waves = 5;
    // table sizes: inside one CU's 32 KB L1, inside one XCD's 4 MB L2, larger than that
    for (unsigned kb : {8u, 16u, 1024u, 18u * 1024u}) {
        const unsigned nRec = kb * 1024u / 64u;
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            dim3 grid(cus * 4 * waves), block(64);
            if (mode == 0) hipLaunchKernelGGL(gather<0>, grid, block, 0, 0, table, nRec, 50, out); else hipLaunchKernelGGL(gather<1>, grid, block, 0, 0, table, nRec, 50, out);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(gather<0>, grid, block, 0, 0, table, nRec, iters, out); else hipLaunchKernelGGL(gather<1>, grid, block, 0, 0, table, nRec, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double recs = (double)iters * 64.0 * grid.x;
            printf("table %6u KB  %-5s  %8.3f ms   %7.1f G records/s   %5.2f records per cycle per CU (2.4 GHz)\n", kb, mode == 0 ? "own" : "quad", ms,
                   recs / ms / 1e6, recs / ms / 1e6 / (cus * 2.4));
        }
    }
    return 0;
}
