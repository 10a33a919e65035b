// post_oracle — scalar CPU restatement of the reference's post-processing chain.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.cpp).
//
// Follows src/main.cpp:242-275 of the reference pass by pass, with the three fragment shaders
// src/shaders/extractBrightFragment.glsl (cited E:), bloomFragment.glsl (B:) and
// finalFragment.glsl (Fi:). All passes run over the whole image, as in the reference, where no
// glViewport call ever shrinks the "down-sampled" passes; the `tmp` image therefore carries
// content from one level to the next and from one frame to the next, and that is reproduced.
//
//   for ds in {5,10,20,40,80,160}:
//     tmp[0:W/ds, 0:H/ds]  <- nearest down-blit of the accumulation      (main.cpp:244-248)
//     level                <- length(tmp.rgb) >= 1.5 ? tmp.rgb : 0       (E:9-16, whole image)
//     tmp                  <- 49-tap disc blur of level                  (B:20-46, whole image)
//     levels[i]            <- bilinear up-blit of tmp[0:W/ds, 0:H/ds]    (main.cpp:261-265)
//   tmp <- pow(accum, 1/1.61803) + levels[0..3]                          (Fi:17-30)
//
// Pinned by tests/golden/post_*.npz (llvmpipe output of the reference's shaders) to a stated
// tolerance: llvmpipe evaluates pow() with polynomial approximations, this file with libm.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Img {
    int W = 0, H = 0;
    std::vector<float> px;   // RGB, row 0 = bottom
    void init(int w, int h) { W = w; H = h; px.assign((size_t)w * h * 3, 0.0f); }
    float* at(int x, int y) { return &px[((size_t)y * W + x) * 3]; }
    const float* at(int x, int y) const { return &px[((size_t)y * W + x) * 3]; }
};

struct Post {
    int W, H;
    Img tmp;
    Img levels[6];
};

const int kDown[6] = {5, 10, 20, 40, 80, 160};   // main.cpp:183

// B:9-18 followed by the per-tap prefactor of B:35: rgb * level * intensity.
inline void colour_correct_premul(const float* rgb, float out[3], float& level) {
    float c[3] = {rgb[0], rgb[1], rgb[2]};
    level = fmaxf(c[0], fmaxf(c[1], c[2]));
    if (level > 1.0f) {
        for (int k = 0; k < 3; ++k) c[k] = c[k] / level;
        float e = 1.0f / level;
        for (int k = 0; k < 3; ++k) c[k] = powf(c[k], e);
    }
    for (int k = 0; k < 3; ++k) out[k] = (c[k] * level) * 0.5f;
}

// texelFetch index of `pixel + coord` (B:33): pixel is the fragment centre (x + 0.5), the sum is
// converted with ivec2(), which truncates toward zero, so -0.5 -> 0, -1.5 -> -1, ...
inline int tap_index(int p, int d) { int s = p + d; return s >= 0 ? s : s + 1; }

void bloom_pass(const Img& level, Img& out) {
    const int W = level.W, H = level.H;
    // per-texel part of the tap (identical for every fragment that fetches the texel)
    std::vector<float> pre((size_t)W * H * 3);
    std::vector<uint8_t> bright((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float lv;
            colour_correct_premul(level.at(x, y), &pre[((size_t)y * W + x) * 3], lv);
            bright[(size_t)y * W + x] = lv >= 1.01f;     // threshold argument of B:45
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float col[3] = {0, 0, 0};
            int i = 1;
            for (int dx = -4; dx <= 4; ++dx)
                for (int dy = -4; dy <= 4; ++dy) {
                    float fx = (float)dx, fy = (float)dy;
                    if (!(sqrtf(fy * fy + fx * fx) <= 4.0f)) continue;            // B:31
                    int tx = tap_index(x, dx), ty = tap_index(y, dy);
                    bool inside = tx >= 0 && tx < W && ty >= 0 && ty < H;          // out-of-range texelFetch reads 0
                    if (inside && bright[(size_t)ty * W + tx]) {
                        float cx = fx / 4.0f, cy = fy / 4.0f;
                        float len = sqrtf(cy * cy + cx * cx);
                        const float* p = &pre[((size_t)ty * W + tx) * 3];
                        for (int k = 0; k < 3; ++k) {
                            float v = p[k] / len;                                  // centre tap: x/0
                            v = fminf(fmaxf(v, 0.0f), 2.0f);                       // clamp; a NaN (0/0) becomes 0
                            col[k] += v;
                        }
                    }
                    i++;
                }
            float* o = out.at(x, y);
            for (int k = 0; k < 3; ++k) o[k] = col[k] / (float)(i > 1 ? i : 1);     // B:41
        }
}

// ---- glBlitFramebuffer as Mesa 23.2.1 llvmpipe executes it ------------------------------------------------
// The reference leaves its blits (main.cpp:244-248 NEAREST, :261-265 LINEAR) to the GL implementation; its parity
// target is llvmpipe, where a blit is a textured quad: src/gallium/auxiliary/util/u_blitter.c draws the destination
// rectangle as a fan of two triangles with normalised texture coordinates, the draw module maps the vertices to the
// window with one fused multiply-add, lp_state_setup.c fits a plane a0 + dadx*x + dady*y through each triangle's
// texture coordinates, lp_bld_interp.c evaluates it per pixel with two fused multiply-adds, and lp_bld_sample_soa.c
// turns the coordinate into texels. The sample points of the even down-blit factors lie exactly on texel boundaries,
// so the last bit of that arithmetic decides which texel is read (for factor 20 at 1280x720: 20x+9, not 20x+10).
// Mesa is not part of the reference tree (system library, 23.2.1-1ubuntu3.1~22.04.3); the steps below restate its
// arithmetic and are pinned by measurement: oracle/glsl_ref/probe_blit.py blits index ramps and random images
// through llvmpipe and finds every texel choice and every filtered value reproduced exactly.
struct Plane { float a0, dadx, dady; };
struct BlitPlan {
    Plane s[2], t[2];        // per triangle of the fan: [0] holds the destination's (x2, y1) corner, [1] its (x1, y2) corner
    long long dw, dh;        // destination rectangle [0,dw) x [0,dh): its diagonal separates the two triangles
    float texW, texH;        // size of the source texture
};

// u_blitter.c blitter_set_rectangle (pixel -> clip space) followed by the viewport transform of the draw module.
inline float window_coord(int x, int surfaceSize) {
    const float ndc = (float)x / (float)surfaceSize * 2.0f - 1.0f;
    const float half = 0.5f * (float)surfaceSize;
    return fmaf(ndc, half, half);
}

// lp_state_setup.c: the plane through attribute values a0v, a1v, a2v at window positions p0, p1, p2.
inline Plane fit_plane(const float p0[2], const float p1[2], const float p2[2], float a0v, float a1v, float a2v) {
    const float x0c = p0[0] - 0.5f, y0c = p0[1] - 0.5f;
    const float dx01 = p0[0] - p1[0], dy01 = p0[1] - p1[1], dx20 = p2[0] - p0[0], dy20 = p2[1] - p0[1];
    const float ooa = 1.0f / (dx01 * dy20 - dy01 * dx20);
    const float dy20o = dy20 * ooa, dy01o = dy01 * ooa, dx20o = dx20 * ooa, dx01o = dx01 * ooa;
    const float da01 = a0v - a1v, da20 = a2v - a0v;
    Plane pl;
    pl.dadx = da01 * dy20o - da20 * dy01o;
    pl.dady = da20 * dx01o - da01 * dx20o;
    pl.a0 = a0v - (pl.dadx * x0c + pl.dady * y0c);
    return pl;
}

// A blit of the source rectangle [0,sw) x [0,sh) of a texW x texH texture onto [0,dw) x [0,dh) of a surfW x surfH surface.
BlitPlan plan_blit(int sw, int sh, int texW, int texH, int dw, int dh, int surfW, int surfH) {
    const float x1 = window_coord(0, surfW), x2 = window_coord(dw, surfW), y1 = window_coord(0, surfH), y2 = window_coord(dh, surfH);
    const float s2 = (float)sw / (float)texW, t2 = (float)sh / (float)texH;
    const float v0[2] = {x1, y1}, v1[2] = {x2, y1}, v2[2] = {x2, y2}, v3[2] = {x1, y2};
    BlitPlan b;
    // the fan's triangles (v0,v1,v2) and (v0,v2,v3) reach the setup code with their first two vertices exchanged
    b.s[0] = fit_plane(v1, v0, v2, s2, 0.0f, s2);
    b.t[0] = fit_plane(v1, v0, v2, 0.0f, 0.0f, t2);
    b.s[1] = fit_plane(v2, v0, v3, s2, 0.0f, 0.0f);
    b.t[1] = fit_plane(v2, v0, v3, t2, 0.0f, t2);
    b.dw = dw; b.dh = dh; b.texW = (float)texW; b.texH = (float)texH;
    return b;
}

// Texture coordinates of destination pixel (x, y), scaled to texels (lp_bld_interp.c, then the sampler's multiply).
inline void blit_coords(const BlitPlan& b, int x, int y, float& u, float& v) {
    // pixel centres on the shared diagonal belong to the first triangle
    const int tri = ((2LL * x + 1) * b.dh - (2LL * y + 1) * b.dw >= 0) ? 0 : 1;
    const float fx = (float)x, fy = (float)y;
    u = fmaf(b.s[tri].dady, fy, fmaf(b.s[tri].dadx, fx, b.s[tri].a0)) * b.texW;
    v = fmaf(b.t[tri].dady, fy, fmaf(b.t[tri].dadx, fx, b.t[tri].a0)) * b.texH;
}

// glBlitFramebuffer(GL_NEAREST) of the whole source onto the corner [0,w) x [0,h).
void blit_down_nearest(const float* accumRGBA, int W, int H, Img& dst, int w, int h) {
    const BlitPlan b = plan_blit(W, H, W, H, w, h, dst.W, dst.H);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float u, v;
            blit_coords(b, x, y, u, v);
            int sx = (int)u, sy = (int)v;                       // truncation, then clamp to the texture
            sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
            sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
            const float* s = accumRGBA + ((size_t)sy * W + sx) * 4;
            float* d = dst.at(x, y);
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        }
}

// glBlitFramebuffer(GL_LINEAR) of the corner [0,w) x [0,h) of src onto the whole destination:
// bilinear on the whole source texture (CLAMP_TO_EDGE), so texels just outside the corner bleed in.
void blit_up_linear(const Img& src, int w, int h, Img& dst) {
    const int W = dst.W, H = dst.H;
    const BlitPlan b = plan_blit(w, h, src.W, src.H, W, H, W, H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float u, v;
            blit_coords(b, x, y, u, v);
            u = fmaxf(fminf(u, b.texW) - 0.5f, 0.0f);
            v = fmaxf(fminf(v, b.texH) - 0.5f, 0.0f);
            const float fu = floorf(u), fv = floorf(v);
            const float wx = u - fu, wy = v - fv;
            const int x0 = (int)fu, y0 = (int)fv;
            const int x1 = x0 + 1 > src.W - 1 ? src.W - 1 : x0 + 1, y1 = y0 + 1 > src.H - 1 ? src.H - 1 : y0 + 1;
            const float *a = src.at(x0, y0), *bb = src.at(x1, y0), *c = src.at(x0, y1), *d = src.at(x1, y1);
            float* o = dst.at(x, y);
            for (int k = 0; k < 3; ++k) {
                const float top = fmaf(wx, bb[k] - a[k], a[k]);
                const float bot = fmaf(wx, d[k] - c[k], c[k]);
                o[k] = fmaf(wy, bot - top, top);
            }
        }
}

}  // namespace

extern "C" {

void* post_create(int W, int H) {
    Post* p = new Post();
    p->W = W; p->H = H;
    p->tmp.init(W, H);
    for (auto& l : p->levels) l.init(W, H);
    return p;
}
void post_destroy(void* h) { delete (Post*)h; }

// accumRGBA: W*H*4 floats. Runs the chain once (one frame). Outputs may be NULL.
//   rgba_out: the tmp image after the final composite (alpha 0); rgb8_out: its 8-bit read-back.
void post_run(void* h, const float* accumRGBA, float* rgba_out, uint8_t* rgb8_out) {
    Post* p = (Post*)h;
    const int W = p->W, H = p->H;
    Img level; level.init(W, H);
    Img blurred; blurred.init(W, H);
    for (int i = 0; i < 6; ++i) {
        const int w = W / kDown[i], hh = H / kDown[i];
        blit_down_nearest(accumRGBA, W, H, p->tmp, w, hh);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float* s = p->tmp.at(x, y);
                float len = sqrtf((s[2] * s[2] + s[1] * s[1]) + s[0] * s[0]);      // E:11
                float* d = level.at(x, y);
                bool keep = len >= 1.5f;
                d[0] = keep ? s[0] : 0.0f; d[1] = keep ? s[1] : 0.0f; d[2] = keep ? s[2] : 0.0f;
            }
        bloom_pass(level, blurred);
        p->tmp.px = blurred.px;
        blit_up_linear(p->tmp, w, hh, p->levels[i]);
    }
    const float g = 1.0f / 1.61803f;                                                // Fi:13-15,21
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float* s = accumRGBA + ((size_t)y * W + x) * 4;
            float* d = p->tmp.at(x, y);
            for (int k = 0; k < 3; ++k) {
                float v = powf(s[k], g);
                for (int l = 0; l < 4; ++l) v += p->levels[l].at(x, y)[k];          // Fi:22-25
                d[k] = v;
            }
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float* s = p->tmp.at(x, y);
            if (rgba_out) { float* o = rgba_out + ((size_t)y * W + x) * 4; o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 0.0f; }
            if (rgb8_out) {
                for (int k = 0; k < 3; ++k) {
                    float f = s[k];
                    f = !(f > 0.0f) ? 0.0f : (f > 1.0f ? 1.0f : f);
                    rgb8_out[((size_t)y * W + x) * 3 + k] = (uint8_t)lrintf(f * 255.0f);
                }
            }
        }
}

// The blit arithmetic on its own (for tests): for every pixel of the destination rectangle [0,dw) x [0,dh) of a blit from
// the source rectangle [0,sw) x [0,sh) of a texW x texH texture, rendered into a surfW x surfH surface, the texel a NEAREST
// blit reads (ix, iy; wx, wy untouched) or the first texel and the weights of a LINEAR blit's footprint.
void post_blit_map(int sw, int sh, int texW, int texH, int dw, int dh, int surfW, int surfH, int linear, int32_t* ix, int32_t* iy, float* wx, float* wy) {
    const BlitPlan b = plan_blit(sw, sh, texW, texH, dw, dh, surfW, surfH);
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const size_t i = (size_t)y * dw + x;
            float u, v;
            blit_coords(b, x, y, u, v);
            if (!linear) {
                int sx = (int)u, sy = (int)v;
                ix[i] = sx < 0 ? 0 : (sx > texW - 1 ? texW - 1 : sx);
                iy[i] = sy < 0 ? 0 : (sy > texH - 1 ? texH - 1 : sy);
            } else {
                u = fmaxf(fminf(u, b.texW) - 0.5f, 0.0f);
                v = fmaxf(fminf(v, b.texH) - 0.5f, 0.0f);
                const float fu = floorf(u), fv = floorf(v);
                ix[i] = (int)fu; iy[i] = (int)fv; wx[i] = u - fu; wy[i] = v - fv;
            }
        }
}

// The fitted planes themselves: c0, cx, cy of the horizontal coordinate for the two triangles, then of the vertical one.
void post_blit_planes(int sw, int sh, int texW, int texH, int dw, int dh, int surfW, int surfH, float* planes12) {
    const BlitPlan b = plan_blit(sw, sh, texW, texH, dw, dh, surfW, surfH);
    const Plane* p[4] = {&b.s[0], &b.s[1], &b.t[0], &b.t[1]};
    for (int k = 0; k < 4; ++k) { planes12[3 * k] = p[k]->a0; planes12[3 * k + 1] = p[k]->dadx; planes12[3 * k + 2] = p[k]->dady; }
}

// One bloom level image after the up-blit (for tests).
void post_level(void* h, int i, float* rgb_out) {
    Post* p = (Post*)h;
    memcpy(rgb_out, p->levels[i].px.data(), p->levels[i].px.size() * sizeof(float));
}

}  // extern "C"
