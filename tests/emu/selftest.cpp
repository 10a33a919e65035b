// Checks of the host emulation itself (tests/test_emulated.py runs this program): the wave- and block-wide operations
// return what the hardware returns, and the two mistakes the emulation exists to catch end the process with a message.
//   selftest              values of ballot / shuffle / readfirstlane / mbcnt / barrier / early exit, two waves per work-group
//   selftest divergent    the lanes of a wave reach different wave-wide operations          -> abort
//   selftest deadlock     half a wave waits at a barrier, the other half at a ballot        -> abort
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void values_kernel(uint32_t* out, unsigned long long* masks) {
    __shared__ uint32_t shared[128];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    shared[t] = t * 3u;
    __syncthreads();
    const uint32_t neighbour = shared[(t + 64u) & 127u];                    // written by the other wave before the barrier
    const unsigned long long odd = __ballot(lane & 1u);
    const unsigned long long mine = __ballot(lane < 10u + wave);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(odd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)odd, 0u));
    const uint32_t first = __builtin_amdgcn_readfirstlane(t + 100u);
    uint32_t sum = lane;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if (lane >= 32u) { out[(blockIdx.x * 128u + t) * 4u] = 0xdeadu; return; }   // half of every wave leaves early
    const unsigned long long rest = __ballot(true);                           // only the lanes still running take part
    out[(blockIdx.x * 128u + t) * 4u + 0] = neighbour;
    out[(blockIdx.x * 128u + t) * 4u + 1] = rank;
    out[(blockIdx.x * 128u + t) * 4u + 2] = first;
    out[(blockIdx.x * 128u + t) * 4u + 3] = sum;
    if (lane == 0) { masks[(blockIdx.x * 2u + wave) * 3u] = odd; masks[(blockIdx.x * 2u + wave) * 3u + 1] = mine; masks[(blockIdx.x * 2u + wave) * 3u + 2] = rest; }
}

__global__ void divergent_kernel(uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t v;
    if (lane < 32u) v = (uint32_t)__ballot(true); else v = __shfl_down(lane, 1, 64);
    out[threadIdx.x] = v;
}

__global__ void deadlock_kernel(uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < 32u) __syncthreads();
    out[threadIdx.x] = (uint32_t)__ballot(true);
    if (lane >= 32u) __syncthreads();
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "values";
    uint32_t* out = nullptr;
    unsigned long long* masks = nullptr;
    if (hipMalloc(&out, 3 * 128 * 4 * sizeof(uint32_t)) != hipSuccess || hipMalloc(&masks, 3 * 2 * 3 * sizeof(unsigned long long)) != hipSuccess) return 2;
    if (!strcmp(mode, "divergent")) { hipLaunchKernelGGL(divergent_kernel, dim3(1), dim3(64), 0, nullptr, out); return 0; }
    if (!strcmp(mode, "deadlock")) { hipLaunchKernelGGL(deadlock_kernel, dim3(1), dim3(64), 0, nullptr, out); return 0; }
    hipLaunchKernelGGL(values_kernel, dim3(3), dim3(128), 0, nullptr, out, masks);
    int bad = 0;
    for (uint32_t b = 0; b < 3; ++b)
        for (uint32_t t = 0; t < 128; ++t) {
            const uint32_t lane = t & 63u, wave = t >> 6;
            const uint32_t* o = out + (b * 128u + t) * 4u;
            if (lane >= 32u) { bad += o[0] != 0xdeadu; continue; }
            bad += o[0] != ((t + 64u) & 127u) * 3u;              // the barrier ordered the two waves' writes and reads
            bad += o[1] != lane / 2u;                            // odd lanes below this one
            bad += o[2] != wave * 64u + 100u;                    // lane 0 of the wave
            bad += lane == 0 && o[3] != 2016u;                   // 0 + 1 + ... + 63
        }
    for (uint32_t w = 0; w < 6; ++w) {
        bad += masks[w * 3] != 0xaaaaaaaaaaaaaaaaull;
        bad += masks[w * 3 + 1] != (1ull << (10 + (w & 1))) - 1ull;
        bad += masks[w * 3 + 2] != 0x00000000ffffffffull;
    }
    std::printf("selftest: %d wrong values\n", bad);
    return bad ? 1 : 0;
}
