// Post-processing kernels: bloom chain and final composite (placeholder build: raw path only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_mi355x.h"

namespace rtd {

namespace {
// glGetTexImage(GL_RGB, GL_UNSIGNED_BYTE) of a float texture: clamp to [0,1], scale by 255, round to nearest.
__global__ void to_rgb8_kernel(const float4* __restrict__ img, uint8_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = img[i];
    float v[3] = {p.x, p.y, p.z};
    for (int k = 0; k < 3; ++k) {
        float f = v[k];
        f = !(f > 0.0f) ? 0.0f : (f > 1.0f ? 1.0f : f);
        out[(size_t)i * 3 + k] = (uint8_t)(int)rintf(f * 255.0f);
    }
}
struct PostScratch { uint8_t* rgb8 = nullptr; size_t n = 0; };
}  // namespace

void FreePostprocess(void** scratch) {
    if (!scratch || !*scratch) return;
    PostScratch* s = static_cast<PostScratch*>(*scratch);
    if (s->rgb8) (void)hipFree(s->rgb8);
    delete s;
    *scratch = nullptr;
}

int RunPostprocess(hipStream_t st, const void* accumLinear, uint32_t width, uint32_t height, int bloomOn, void** scratch,
                   float* rgbaOutHost, uint8_t* rgb8OutHost, std::string& err) {
    if (bloomOn) { err = "bloom chain not built yet"; return RT_ERR_UNSUPPORTED; }
    const size_t n = (size_t)width * height;
    if (!*scratch) *scratch = new PostScratch();
    PostScratch* s = static_cast<PostScratch*>(*scratch);
    if (s->n != n) {
        if (s->rgb8) (void)hipFree(s->rgb8);
        s->rgb8 = nullptr; s->n = 0;
        if (hipMalloc((void**)&s->rgb8, n * 3) != hipSuccess) { err = "out of device memory"; return RT_ERR_DEVICE; }
        s->n = n;
    }
    if (rgbaOutHost && hipMemcpyAsync(rgbaOutHost, accumLinear, n * 16, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "copy failed"; return RT_ERR_DEVICE; }
    if (rgb8OutHost) {
        hipLaunchKernelGGL(to_rgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const float4*>(accumLinear), s->rgb8, (uint32_t)n);
        if (hipMemcpyAsync(rgb8OutHost, s->rgb8, n * 3, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "copy failed"; return RT_ERR_DEVICE; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { err = "sync failed"; return RT_ERR_DEVICE; }
    return RT_OK;
}

}  // namespace rtd
