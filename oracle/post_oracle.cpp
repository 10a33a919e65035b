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

// glBlitFramebuffer(GL_NEAREST) of the whole source onto the corner [0,w) x [0,h).
void blit_down_nearest(const float* accumRGBA, int W, int H, Img& dst, int w, int h) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int sx = (int)floorf(((float)x + 0.5f) * ((float)W / (float)w));
            int sy = (int)floorf(((float)y + 0.5f) * ((float)H / (float)h));
            if (sx > W - 1) sx = W - 1;
            if (sy > H - 1) sy = H - 1;
            const float* s = accumRGBA + ((size_t)sy * W + sx) * 4;
            float* d = dst.at(x, y);
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        }
}

// glBlitFramebuffer(GL_LINEAR) of the corner [0,w) x [0,h) of src onto the whole destination:
// bilinear on the whole source texture (CLAMP_TO_EDGE), so texels just outside the corner bleed in.
void blit_up_linear(const Img& src, int w, int h, Img& dst) {
    const int W = dst.W, H = dst.H;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float u = ((float)x + 0.5f) * ((float)w / (float)W) - 0.5f;
            float v = ((float)y + 0.5f) * ((float)h / (float)H) - 0.5f;
            float fu = floorf(u), fv = floorf(v);
            float wx = u - fu, wy = v - fv;
            int x0 = (int)fu, y0 = (int)fv, x1 = x0 + 1, y1 = y0 + 1;
            auto cl = [](int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); };
            x0 = cl(x0, src.W); x1 = cl(x1, src.W); y0 = cl(y0, src.H); y1 = cl(y1, src.H);
            const float *a = src.at(x0, y0), *b = src.at(x1, y0), *c = src.at(x0, y1), *d = src.at(x1, y1);
            float* o = dst.at(x, y);
            for (int k = 0; k < 3; ++k) {
                float top = a[k] + wx * (b[k] - a[k]);
                float bot = c[k] + wx * (d[k] - c[k]);
                o[k] = top + wy * (bot - top);
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

// One bloom level image after the up-blit (for tests).
void post_level(void* h, int i, float* rgb_out) {
    Post* p = (Post*)h;
    memcpy(rgb_out, p->levels[i].px.data(), p->levels[i].px.size() * sizeof(float));
}

}  // extern "C"
