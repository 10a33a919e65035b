// Post-processing kernels for gfx950: the reference's bloom chain and final composite
// (src/main.cpp:242-275 with extractBrightFragment.glsl = E:, bloomFragment.glsl = B:,
// finalFragment.glsl = Fi:), plus the 8-bit read-back of Include/Recorder.h:93-100.
//
// The reference runs every pass over the whole window (it never calls glViewport), although only
// a small corner of each pass is used afterwards, and its `tmp` texture carries pixels from one
// level to the next and from one frame to the next. Here each pass is evaluated only on the
// region later passes actually read (derived in Regions() below), which gives the same images:
//   per level ds in {5,10,20,40}:   (levels 80 and 160 are computed by the reference but never read: Fi:10,26)
//     down   tmp[0:w,0:h] <- nearest sample of the accumulation               main.cpp:244-248
//     pre    threshold E:9-16 + per-texel half of the blur tap B:9-18,35      (whole needed region)
//     blur   49-tap disc, B:20-46, LDS-tiled: a 40x16 texel tile feeds 32x8 outputs
//     up     levels[i] <- bilinear up-blit of tmp[0:w,0:h]                    main.cpp:261-265
//   final  tmp <- pow(accum, 1/1.61803) + levels[0..3]                        Fi:17-30
// The two blits are evaluated as Mesa llvmpipe (the reference's parity target) evaluates glBlitFramebuffer: see
// BlitMap below. That matters for the down-blit: the sample point of an even factor lies exactly on a texel
// boundary, and the rounding of the interpolated coordinate decides which of the two texels is read.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <atomic>
#include <string>

#include "rt_device.h"
#include "rt_mi355x.h"

namespace rtd {

namespace {

constexpr int kLevels = 4;

// The accumulation image as the tracer keeps it: 8x8-pixel tiles, tile-major (or row-major when tilesX == 0).
struct AccumView {
    const float4* p;
    int W, tilesX;
    __device__ __forceinline__ float4 at(int x, int y) const {
        if (tilesX == 0) return p[(size_t)y * W + x];
        return p[((size_t)(y >> 3) * tilesX + (x >> 3)) * 64 + ((y & 7) << 3) + (x & 7)];
    }
};

// ---- glBlitFramebuffer (main.cpp:244-248, 261-265) with llvmpipe's arithmetic ----
// llvmpipe draws a blit as two triangles that share the diagonal of the destination rectangle and carry normalised
// texture coordinates. Per triangle and coordinate it fits a plane c0 + cx * x + cy * y in fp32 (its coefficients are
// off by an ulp or so, differently for the two triangles), evaluates the plane at the pixel's integer position with
// two fused multiply-adds, and multiplies by the texture size. The planes are fitted on the host (MakeBlitMap); the
// kernels repeat the per-pixel part operation by operation, so that NEAREST reads the texel llvmpipe reads (with a
// down-blit factor of 20 at 1280 x 720 that is texel 20x + 9 in every column, never the 20x + 10 of exact arithmetic)
// and LINEAR gets its weights bit for bit.
struct BlitPlane { float c0, cx, cy; };
struct BlitMap {
    BlitPlane u[2], v[2];     // horizontal / vertical coordinate; [0]: triangle below the diagonal, [1]: above it
    int dw, dh;               // destination rectangle [0,dw) x [0,dh)
    float texW, texH;         // source texture size in texels
};

__device__ __forceinline__ void blit_source_coords(const BlitMap& m, int x, int y, float& su, float& sv) {
    // which side of the diagonal from (0,0) to (dw,dh) the pixel centre lies on; centres on it go to triangle 0
    const bool below = (long long)(2 * x + 1) * m.dh >= (long long)(2 * y + 1) * m.dw;
    const BlitPlane pu = below ? m.u[0] : m.u[1], pv = below ? m.v[0] : m.v[1];
    const float fx = (float)x, fy = (float)y;
    su = __fmaf_rn(pu.cy, fy, __fmaf_rn(pu.cx, fx, pu.c0)) * m.texW;
    sv = __fmaf_rn(pv.cy, fy, __fmaf_rn(pv.cx, fx, pv.c0)) * m.texH;
}

// (S = row stride of tmp and pre, in texels: the window width when they are whole-window images)
__global__ void down_kernel(AccumView accum, float4* __restrict__ tmp, int S, int W, int H, int w, int h, BlitMap map) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float su, sv;
    blit_source_coords(map, x, y, su, sv);
    const int sx = min(max((int)su, 0), W - 1), sy = min(max((int)sv, 0), H - 1);      // truncate, clamp to the texture
    float4 s = accum.at(sx, sy);
    tmp[(size_t)y * S + x] = make_float4(s.x, s.y, s.z, 0.0f);
}

// E:9-16 then B:9-18 and the texel-only factors of B:35: out = (cc(rgb) * level) * 0.5, w = level >= 1.01
__global__ void pre_kernel(const float4* __restrict__ tmp, float4* __restrict__ pre, int S, int rw, int rh) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= rw || y >= rh) return;
    float4 s = tmp[(size_t)y * S + x];
    float len = sqrtf((s.z * s.z + s.y * s.y) + s.x * s.x);
    float c[3] = {0.0f, 0.0f, 0.0f};
    if (len >= 1.5f) { c[0] = s.x; c[1] = s.y; c[2] = s.z; }
    float level = fmaxf(c[0], fmaxf(c[1], c[2]));
    if (level > 1.0f) {
        float e = 1.0f / level;
        for (int k = 0; k < 3; ++k) c[k] = powf(c[k] / level, e);
    }
    pre[(size_t)y * S + x] = make_float4((c[0] * level) * 0.5f, (c[1] * level) * 0.5f, (c[2] * level) * 0.5f, level >= 1.01f ? 1.0f : 0.0f);
}

constexpr int kBlurTX = 32, kBlurTY = 8, kApron = 4;
constexpr int kTileW = kBlurTX + 2 * kApron, kTileH = kBlurTY + 2 * kApron;

// B:20-46. One block = 32x8 outputs; the 40x16 texel neighbourhood is staged in LDS once
// instead of being fetched 49 times per output.
__global__ __launch_bounds__(kBlurTX* kBlurTY) void blur_kernel(const float4* __restrict__ pre, float4* __restrict__ tmp, int S, int W, int H,
                                                                  int preW, int preH, int rw, int rh) {
    __shared__ float4 tile[kTileH][kTileW];
    const int bx0 = blockIdx.x * kBlurTX, by0 = blockIdx.y * kBlurTY;
    for (int i = threadIdx.y * kBlurTX + threadIdx.x; i < kTileW * kTileH; i += kBlurTX * kBlurTY) {
        int lx = i % kTileW, ly = i / kTileW;
        int gx = bx0 - kApron + lx, gy = by0 - kApron + ly;
        // texelFetch outside the texture returns 0; `pre` is only valid on [0,preW) x [0,preH)
        bool ok = gx >= 0 && gy >= 0 && gx < W && gy < H && gx < preW && gy < preH;
        tile[ly][lx] = ok ? pre[(size_t)gy * S + gx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int x = bx0 + threadIdx.x, y = by0 + threadIdx.y;
    if (x >= rw || y >= rh) return;
    float col[3] = {0.0f, 0.0f, 0.0f};
    int n = 1;
    for (int dx = -4; dx <= 4; ++dx) {
        for (int dy = -4; dy <= 4; ++dy) {
            float fx = (float)dx, fy = (float)dy;
            if (!(sqrtf(fy * fy + fx * fx) <= 4.0f)) continue;                 // B:31
            // ivec2(pixel + coord) truncates toward zero: -0.5 -> 0, -1.5 -> -1 (B:33)
            int sx = x + dx, sy = y + dy;
            int tx = sx >= 0 ? sx : sx + 1, ty = sy >= 0 ? sy : sy + 1;
            float4 p = tile[ty - by0 + kApron][tx - bx0 + kApron];
            if (p.w != 0.0f) {
                float cx = fx / 4.0f, cy = fy / 4.0f;
                float len = sqrtf(cy * cy + cx * cx);
                float v0 = p.x / len, v1 = p.y / len, v2 = p.z / len;          // centre tap divides by 0
                col[0] += fminf(fmaxf(v0, 0.0f), 2.0f);                        // clamp; NaN (0/0) -> 0
                col[1] += fminf(fmaxf(v1, 0.0f), 2.0f);
                col[2] += fminf(fmaxf(v2, 0.0f), 2.0f);
            }
            n++;
        }
    }
    float d = (float)(n > 1 ? n : 1);
    tmp[(size_t)y * S + x] = make_float4(col[0] / d, col[1] / d, col[2] / d, 0.0f);
}

// Keeps what the GL_LINEAR up-blit of level i would read: tmp[0:w+1, 0:h+1], the corner plus the one texel of
// bleed beyond it (CLAMP_TO_EDGE acts at the texture edge, not at the source rectangle, so that texel is whatever
// earlier passes left in tmp). The next level overwrites the corner, hence the copy.
__global__ void snap_kernel(const float4* __restrict__ tmp, float4* __restrict__ snap, int S, int sw, int sh) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= sw || y >= sh) return;
    snap[(size_t)y * sw + x] = tmp[(size_t)y * S + x];
}

struct LevelView { const float4* snap; int w, h, sw, sh; BlitMap map; int texW, texH; };

// One pixel of the GL_LINEAR up-blit of a level's corner [0:w, 0:h] to the whole image, read from its snapshot.
__device__ __forceinline__ float4 up_sample(const LevelView& L, int x, int y) {
    float u, v;
    blit_source_coords(L.map, x, y, u, v);
    // CLAMP_TO_EDGE bilinear footprint: texel centres at +0.5, coordinate held inside the texture
    u = fmaxf(fminf(u, L.map.texW) - 0.5f, 0.0f);
    v = fmaxf(fminf(v, L.map.texH) - 0.5f, 0.0f);
    const float fu = floorf(u), fv = floorf(v);
    const float wx = u - fu, wy = v - fv;
    int x0 = (int)fu, y0 = (int)fv, x1 = min(x0 + 1, L.texW - 1), y1 = min(y0 + 1, L.texH - 1);
    // the footprint never leaves [0, w] x [0, h], which is what the snapshot holds (or the whole texture, if smaller)
    x0 = min(x0, L.sw - 1); x1 = min(x1, L.sw - 1); y0 = min(y0, L.sh - 1); y1 = min(y1, L.sh - 1);
    float4 a = L.snap[(size_t)y0 * L.sw + x0], b = L.snap[(size_t)y0 * L.sw + x1], c = L.snap[(size_t)y1 * L.sw + x0], d = L.snap[(size_t)y1 * L.sw + x1];
    float4 o;      // llvmpipe's lerps are fused: v0 + w * (v1 - v0) rounded once
    { float top = __fmaf_rn(wx, b.x - a.x, a.x), bot = __fmaf_rn(wx, d.x - c.x, c.x); o.x = __fmaf_rn(wy, bot - top, top); }
    { float top = __fmaf_rn(wx, b.y - a.y, a.y), bot = __fmaf_rn(wx, d.y - c.y, c.y); o.y = __fmaf_rn(wy, bot - top, top); }
    { float top = __fmaf_rn(wx, b.z - a.z, a.z), bot = __fmaf_rn(wx, d.z - c.z, c.z); o.z = __fmaf_rn(wy, bot - top, top); }
    o.w = 0.0f;
    return o;
}

// glGetTexImage(GL_RGB, GL_UNSIGNED_BYTE) of a float texture: clamp to [0,1], scale by 255, round to nearest.
__device__ __forceinline__ uint8_t to_unorm8(float f) {
    f = !(f > 0.0f) ? 0.0f : (f > 1.0f ? 1.0f : f);
    return (uint8_t)(int)rintf(f * 255.0f);
}

// pow(x, g) for the gamma step, 0 < g < 1. Up to 64 (an accumulated radiance is rarely above a few units) it is
// exp2(g * log2(x)) on the hardware's two transcendental instructions: their errors of about one ulp each leave the
// result within 1e-6 of powf there, a twentieth of the tolerance the chain is held to against the reference (whose own
// pow is an approximation). Zero, negative values, infinities and NaNs come out as powf returns them. Larger values take
// powf, in a branch a wave skips when none of its lanes needs it.
__device__ __forceinline__ float gamma_pow(float x, float g) {
    float y = __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(x));
    if (__builtin_expect(x > 64.0f, 0)) y = powf(x, g);
    return y;
}

// Fi:17-30, with the four up-blits of main.cpp:262-268 evaluated in place: the reference first writes each level,
// enlarged to the whole window, into a texture of its own and reads the four back here; the values added are the same.
// The result goes to a row-major float image and/or to its 8-bit read-back (either may be NULL).
__global__ void final_kernel(AccumView accum, LevelView l0, LevelView l1, LevelView l2, LevelView l3,
                             float4* __restrict__ outRgba, uint8_t* __restrict__ outRgb8, int W, int H) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    float4 s = accum.at(x, y);
    const float g = 1.0f / 1.61803f;
    float4 a = up_sample(l0, x, y), b = up_sample(l1, x, y), c = up_sample(l2, x, y), d = up_sample(l3, x, y);
    float r = gamma_pow(s.x, g), gr = gamma_pow(s.y, g), bl = gamma_pow(s.z, g);
    r += a.x; r += b.x; r += c.x; r += d.x;
    gr += a.y; gr += b.y; gr += c.y; gr += d.y;
    bl += a.z; bl += b.z; bl += c.z; bl += d.z;
    if (outRgba) outRgba[i] = make_float4(r, gr, bl, 0.0f);
    if (outRgb8) { outRgb8[i * 3 + 0] = to_unorm8(r); outRgb8[i * 3 + 1] = to_unorm8(gr); outRgb8[i * 3 + 2] = to_unorm8(bl); }
}

__global__ void to_rgb8_kernel(const float4* __restrict__ img, uint8_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = img[i];
    out[i * 3 + 0] = to_unorm8(p.x); out[i * 3 + 1] = to_unorm8(p.y); out[i * 3 + 2] = to_unorm8(p.z);
}

struct PostScratch {
    uint32_t W = 0, H = 0;
    float4* tmp = nullptr;            // persists between calls, like the reference's tmp texture
    float4* pre = nullptr;
    float4* snap[kLevels] = {nullptr, nullptr, nullptr, nullptr};   // per level: blurred corner + one texel of bleed
    int sw[kLevels] = {0, 0, 0, 0}, sh[kLevels] = {0, 0, 0, 0};
    uint8_t* rgb8 = nullptr;
};

void Release(PostScratch* s) {
    if (s->tmp) (void)hipFree(s->tmp);
    if (s->pre) (void)hipFree(s->pre);
    for (auto& l : s->snap) if (l) (void)hipFree(l);
    if (s->rgb8) (void)hipFree(s->rgb8);
    *s = PostScratch();
}

// The planes of a blit from the source rectangle [0,sw) x [0,sh) of a texW x texH texture onto [0,dw) x [0,dh) of a
// surfW x surfH render target, fitted the way llvmpipe's triangle setup fits them (fp32 throughout, one fused
// multiply-add where the vertices are mapped to the window, none in the fit itself).
BlitPlane FitPlane(float ax, float ay, float bx, float by, float cx, float cy, float va, float vb, float vc) {
    // vertices a, b, c with attribute values va, vb, vc; a is the vertex the constant term is anchored at
    const float abx = ax - bx, aby = ay - by, cax = cx - ax, cay = cy - ay;
    const float rArea = 1.0f / (abx * cay - aby * cax);
    const float cayN = cay * rArea, abyN = aby * rArea, caxN = cax * rArea, abxN = abx * rArea;
    const float dab = va - vb, dca = vc - va;
    BlitPlane p;
    p.cx = dab * cayN - dca * abyN;
    p.cy = dca * abxN - dab * caxN;
    p.c0 = va - (p.cx * (ax - 0.5f) + p.cy * (ay - 0.5f));
    return p;
}

BlitMap MakeBlitMap(int sw, int sh, int texW, int texH, int dw, int dh, int surfW, int surfH) {
    auto toWindow = [](int px, int size) {      // pixel -> clip space -> window, as the blit's vertices travel
        const float clip = (float)px / (float)size * 2.0f - 1.0f, half = 0.5f * (float)size;
        return fmaf(clip, half, half);
    };
    const float xl = toWindow(0, surfW), xr = toWindow(dw, surfW), yb = toWindow(0, surfH), yt = toWindow(dh, surfH);
    const float uMax = (float)sw / (float)texW, vMax = (float)sh / (float)texH;
    BlitMap m;
    // triangle below the diagonal: corners (xr,yb), (xl,yb), (xr,yt) in the order the setup sees them; above: (xr,yt), (xl,yb), (xl,yt)
    m.u[0] = FitPlane(xr, yb, xl, yb, xr, yt, uMax, 0.0f, uMax);
    m.v[0] = FitPlane(xr, yb, xl, yb, xr, yt, 0.0f, 0.0f, vMax);
    m.u[1] = FitPlane(xr, yt, xl, yb, xl, yt, uMax, 0.0f, 0.0f);
    m.v[1] = FitPlane(xr, yt, xl, yb, xl, yt, vMax, 0.0f, vMax);
    m.dw = dw; m.dh = dh; m.texW = (float)texW; m.texH = (float)texH;
    return m;
}

// Region [0,R[i]) x [0,R'[i]) on which level i's blur output is needed:
//   the up-blit reads the corner plus one texel of bleed; the next level's blur (over its own
//   region) reads this level's output, as left-over tmp content, up to 4 texels further out.
void Regions(int W, int H, int rw[kLevels], int rh[kLevels]) {
    const int ds[kLevels] = {5, 10, 20, 40};
    int needW = 0, needH = 0;
    for (int i = kLevels - 1; i >= 0; --i) {
        rw[i] = W / ds[i] + 1 > needW ? W / ds[i] + 1 : needW;
        rh[i] = H / ds[i] + 1 > needH ? H / ds[i] + 1 : needH;
        if (rw[i] > W) rw[i] = W;
        if (rh[i] > H) rh[i] = H;
        needW = rw[i] + kApron; needH = rh[i] + kApron;
    }
}

// The buffers one run of the chain works in. tmp and pre have `stride` texels per row and at least
// ChainRows() rows of ChainCols() texels; snap[i] holds sw[i] x sh[i] texels.
struct ChainBuffers {
    float4* tmp; float4* pre; int stride;
    float4* snap[kLevels]; int sw[kLevels], sh[kLevels];
};

void SnapSizes(int W, int H, int sw[kLevels], int sh[kLevels]) {
    const int ds[kLevels] = {5, 10, 20, 40};
    for (int i = 0; i < kLevels; ++i) { sw[i] = min(W / ds[i] + 1, W); sh[i] = min(H / ds[i] + 1, H); }
}
// The part of tmp / pre the level passes touch: everything lies in [0, ChainCols) x [0, ChainRows).
int ChainCols(int W, int H) { int rw[kLevels], rh[kLevels]; Regions(W, H, rw, rh); return min(W, rw[0] + kApron); }
int ChainRows(int W, int H) { int rw[kLevels], rh[kLevels]; Regions(W, H, rw, rh); return min(H, rh[0] + kApron); }

// Queues the level passes and the final composite on `st`.
void LaunchChain(hipStream_t st, const AccumView& view, int W, int H, const ChainBuffers& cb, float4* outRgba, uint8_t* outRgb8) {
    const dim3 b2(32, 8);
    auto grid2 = [&](int w, int h) { return dim3((unsigned)((w + 31) / 32), (unsigned)((h + 7) / 8)); };
    const int ds[kLevels] = {5, 10, 20, 40};
    const int S = cb.stride;
    int rw[kLevels], rh[kLevels];
    Regions(W, H, rw, rh);
    LevelView lv[kLevels];
    for (int i = 0; i < kLevels; ++i) {
        const int w = W / ds[i], h = H / ds[i];
        if (w > 0 && h > 0) hipLaunchKernelGGL(down_kernel, grid2(w, h), b2, 0, st, view, cb.tmp, S, W, H, w, h, MakeBlitMap(W, H, W, H, w, h, W, H));
        const int pw = min(W, rw[i] + kApron), ph = min(H, rh[i] + kApron);
        hipLaunchKernelGGL(pre_kernel, grid2(pw, ph), b2, 0, st, cb.tmp, cb.pre, S, pw, ph);
        hipLaunchKernelGGL(blur_kernel, grid2(rw[i], rh[i]), dim3(kBlurTX, kBlurTY), 0, st, cb.pre, cb.tmp, S, W, H, pw, ph, rw[i], rh[i]);
        hipLaunchKernelGGL(snap_kernel, grid2(cb.sw[i], cb.sh[i]), b2, 0, st, cb.tmp, cb.snap[i], S, cb.sw[i], cb.sh[i]);
        lv[i] = LevelView{cb.snap[i], w, h, cb.sw[i], cb.sh[i], MakeBlitMap(w, h, W, H, W, H, W, H), W, H};
    }
    hipLaunchKernelGGL(final_kernel, grid2(W, H), b2, 0, st, view, lv[0], lv[1], lv[2], lv[3], outRgba, outRgb8, W, H);
}

}  // namespace

void FreePostprocess(void** scratch) {
    if (!scratch || !*scratch) return;
    PostScratch* s = static_cast<PostScratch*>(*scratch);
    Release(s);
    delete s;
    *scratch = nullptr;
}

// Back to the state of a freshly started process: the tmp image is cleared (the buffers stay).
int ResetPostprocess(hipStream_t st, void** scratch) {
    PostScratch* s = scratch ? static_cast<PostScratch*>(*scratch) : nullptr;
    if (!s || !s->tmp) return RT_OK;
    return hipMemsetAsync(s->tmp, 0, (size_t)s->W * s->H * 16, st) == hipSuccess ? RT_OK : RT_ERR_DEVICE;
}

int ReadDisplay(hipStream_t st, void** scratch, uint32_t width, uint32_t height, float* rgbaOutHost, uint8_t* rgb8OutHost, std::string& err) {
    PostScratch* s = scratch ? static_cast<PostScratch*>(*scratch) : nullptr;
    if (!s || s->W != width || s->H != height) { err = "nothing on display yet (run rt_postprocess with bloom_on = 1 first)"; return RT_ERR_STATE; }
    const size_t n = (size_t)width * height;
    if (rgbaOutHost && hipMemcpyAsync(rgbaOutHost, s->tmp, n * 16, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "copy failed"; return RT_ERR_DEVICE; }
    if (rgb8OutHost) {
        hipLaunchKernelGGL(to_rgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->tmp, s->rgb8, n);
        if (hipMemcpyAsync(rgb8OutHost, s->rgb8, n * 3, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "copy failed"; return RT_ERR_DEVICE; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { err = "stream synchronisation failed"; return RT_ERR_DEVICE; }
    return RT_OK;
}

// `accumImage` is the tile-major accumulation image when tilesX > 0 (bloom chain: the kernels index it directly) and a
// row-major copy of it when tilesX == 0 (needed for the raw path, whose image is handed out as it is).
int RunPostprocess(hipStream_t st, const void* accumImage, uint32_t tilesX, uint32_t width, uint32_t height, int bloomOn, void** scratch,
                   float* rgbaOutHost, uint8_t* rgb8OutHost, std::string& err) {
    const size_t n = (size_t)width * height;
    if (!*scratch) *scratch = new PostScratch();
    PostScratch* s = static_cast<PostScratch*>(*scratch);
    if (s->W != width || s->H != height) {
        Release(s);
        hipError_t e = hipMalloc((void**)&s->tmp, n * 16);
        if (e == hipSuccess) e = hipMemsetAsync(s->tmp, 0, n * 16, st);
        if (e == hipSuccess) e = hipMalloc((void**)&s->pre, n * 16);
        SnapSizes((int)width, (int)height, s->sw, s->sh);
        for (int i = 0; i < kLevels && e == hipSuccess; ++i) e = hipMalloc((void**)&s->snap[i], (size_t)s->sw[i] * s->sh[i] * 16);
        if (e == hipSuccess) e = hipMalloc((void**)&s->rgb8, n * 3);
        if (e != hipSuccess) { Release(s); err = hipGetErrorString(e); return RT_ERR_DEVICE; }
        s->W = width; s->H = height;
    }
    const int W = (int)width, H = (int)height;
    if (!bloomOn && tilesX != 0) { err = "the raw path needs the row-major image"; return RT_ERR_INVALID; }
    const float4* accum = static_cast<const float4*>(accumImage);
    const AccumView view{accum, (int)width, (int)tilesX};
    // bloom off = the raw blit of main.cpp:235-240: the accumulation is shown as it is and tmp is left alone
    const float4* shown = accum;
    if (bloomOn) {
        ChainBuffers cb;
        cb.tmp = s->tmp; cb.pre = s->pre; cb.stride = W;
        for (int i = 0; i < kLevels; ++i) { cb.snap[i] = s->snap[i]; cb.sw[i] = s->sw[i]; cb.sh[i] = s->sh[i]; }
        LaunchChain(st, view, W, H, cb, s->tmp, nullptr);       // the composite replaces tmp, as in the reference
        shown = s->tmp;
    }
    if (hipGetLastError() != hipSuccess) { err = "kernel launch failed"; return RT_ERR_DEVICE; }
    if (rgbaOutHost && hipMemcpyAsync(rgbaOutHost, shown, n * 16, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "copy failed"; return RT_ERR_DEVICE; }
    if (rgb8OutHost) {
        hipLaunchKernelGGL(to_rgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, shown, s->rgb8, n);
        if (hipMemcpyAsync(rgb8OutHost, s->rgb8, n * 3, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "copy failed"; return RT_ERR_DEVICE; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { err = "stream synchronisation failed"; return RT_ERR_DEVICE; }
    return RT_OK;
}

}  // namespace rtd

// ---------------------------------------------------------------------------------------------------------------
// rt_postprocess_static: the bloom chain for callers that must not allocate (code under HIP graph capture, tight
// serving loops). Same kernels as rt_postprocess; what that entry point keeps in buffers it allocates per context
// (the working images tmp and pre, cut down to the corner the level passes touch, and the four level snapshots)
// lives here in one fixed array in device memory, one copy per GPU the library is loaded on.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kStaticScratchBytes = 20u << 20;      // enough for a 3840 x 2160 image (17.9 MB)
constexpr int kMaxStaticDevices = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
// The array's address on each device, looked up once per device with that device current. In host code the name
// `scratch` is only a handle for this lookup; kernels and memsets get the pointer kept here.
// (one atomic slot per device, nullptr = not looked up yet: two host threads making their first call on a device at
// the same time both look the address up and store the same value)
std::atomic<void*> g_staticScratch[kMaxStaticDevices];

hipError_t StaticScratchOfCurrentDevice(void** out, int* deviceOut) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxStaticDevices) return hipErrorInvalidDevice;
    void* p = g_staticScratch[dev].load(std::memory_order_acquire);
    if (!p) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_staticScratch[dev].store(p, std::memory_order_release);
    }
    *out = p;
    if (deviceOut) *deviceOut = dev;
    return hipSuccess;
}
}  // namespace

namespace rtd {
void SetGlobalError(const char* fmt, ...);
// Called by rt_create with the context's device current, so that the address is known before a caller that cannot
// afford the lookup (graph capture) makes its first rt_postprocess_static call on that device.
int PrepareStaticScratch() {
    void* p = nullptr;
    return StaticScratchOfCurrentDevice(&p, nullptr) == hipSuccess ? RT_OK : RT_ERR_DEVICE;
}
}  // namespace rtd

extern "C" int rt_probe_blit_planes(uint32_t srcW, uint32_t srcH, uint32_t texW, uint32_t texH, uint32_t dstW, uint32_t dstH, uint32_t surfW, uint32_t surfH,
                                    float* planes12) {
    using namespace rtd;
    if (!planes12 || !srcW || !srcH || !texW || !texH || !dstW || !dstH || !surfW || !surfH ||
        (srcW | srcH | texW | texH | dstW | dstH | surfW | surfH) > 32768u) { SetGlobalError("rt_probe_blit_planes: bad argument"); return RT_ERR_INVALID; }
    const BlitMap m = MakeBlitMap((int)srcW, (int)srcH, (int)texW, (int)texH, (int)dstW, (int)dstH, (int)surfW, (int)surfH);
    const BlitPlane* p[4] = {&m.u[0], &m.u[1], &m.v[0], &m.v[1]};
    for (int k = 0; k < 4; ++k) { planes12[3 * k] = p[k]->c0; planes12[3 * k + 1] = p[k]->cx; planes12[3 * k + 2] = p[k]->cy; }
    return RT_OK;
}

extern "C" int rt_postprocess_static(const float* accumDevice, uint32_t width, uint32_t height, float* rgbaOutDevice, uint8_t* rgb8OutDevice,
                                     void* hipStream) {
    using namespace rtd;
    if (!accumDevice || (!rgbaOutDevice && !rgb8OutDevice)) { SetGlobalError("rt_postprocess_static: NULL image"); return RT_ERR_INVALID; }
    if (width == 0 || height == 0 || width > 32768 || height > 32768) { SetGlobalError("rt_postprocess_static: bad image size %ux%u", width, height); return RT_ERR_INVALID; }
    const int W = (int)width, H = (int)height;
    // carve the fixed array: tmp, pre (both cols x rows texels), then the four snapshots; every piece 256-byte aligned
    const int cols = ChainCols(W, H), rows = ChainRows(W, H);
    ChainBuffers cb;
    SnapSizes(W, H, cb.sw, cb.sh);
    auto align = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    const size_t regionBytes = align((size_t)cols * rows * sizeof(float4));
    size_t need = 2 * regionBytes;
    size_t snapAt[kLevels];
    for (int i = 0; i < kLevels; ++i) { snapAt[i] = need; need += align((size_t)cb.sw[i] * cb.sh[i] * sizeof(float4)); }
    if (need > kStaticScratchBytes) {
        SetGlobalError("rt_postprocess_static: a %ux%u image needs %zu bytes of working memory, the fixed array holds %zu (up to 3840x2160); "
                       "use rt_postprocess", width, height, need, (size_t)kStaticScratchBytes);
        return RT_ERR_UNSUPPORTED;
    }
    void* base = nullptr;
    hipError_t e = StaticScratchOfCurrentDevice(&base, nullptr);
    if (e != hipSuccess) { SetGlobalError("rt_postprocess_static: hipGetSymbolAddress: %s", hipGetErrorString(e)); return RT_ERR_DEVICE; }
    unsigned char* b = static_cast<unsigned char*>(base);
    cb.tmp = reinterpret_cast<float4*>(b); cb.pre = reinterpret_cast<float4*>(b + regionBytes); cb.stride = cols;
    for (int i = 0; i < kLevels; ++i) cb.snap[i] = reinterpret_cast<float4*>(b + snapAt[i]);
    hipStream_t st = (hipStream_t)hipStream;
    // The array keeps what the last call left in it. tmp must start as a freshly created texture does (zero: the level
    // passes read texels of it that nothing has written yet); pre and the snapshots are written before they are read.
    e = hipMemsetAsync(cb.tmp, 0, (size_t)cols * rows * sizeof(float4), st);
    if (e != hipSuccess) { SetGlobalError("rt_postprocess_static: hipMemsetAsync: %s", hipGetErrorString(e)); return RT_ERR_DEVICE; }
    const AccumView view{reinterpret_cast<const float4*>(accumDevice), W, 0};
    LaunchChain(st, view, W, H, cb, reinterpret_cast<float4*>(rgbaOutDevice), rgb8OutDevice);
    e = hipGetLastError();
    if (e != hipSuccess) { SetGlobalError("rt_postprocess_static: kernel launch: %s", hipGetErrorString(e)); return RT_ERR_DEVICE; }
    return RT_OK;
}
