// Path-tracing kernels for gfx950 (MI355X).
//
// WHAT is computed is the reference's tracer pass (src/shaders/Fragment.glsl of the reference,
// cited below as F:<line>), with every floating-point operation in the same order and rounding
// as that shader evaluates it under Mesa llvmpipe: reversed three-term dot products, normalize as
// multiply by 1/sqrt, mix as a + t*(b-a), correctly rounded division (exact_div.h holds shorter sequences for
// 1/x and a/uniform that return the same bits) and square root, denormals flushed, and
// llvmpipe's 8-bit fixed-point bilinear texture filter. This file must be compiled with
// -ffp-contract=off -fgpu-flush-denormals-to-zero (see Makefile; -fno-slp-vectorize is for speed
// only); results are then bit-identical to the CPU parity oracle, and per-pixel results do not
// depend on how pixels are scheduled.
//
// HOW it is computed is specific to this hardware:
//   * BVH: 64-byte records hold both children of an interior node, leaves are encoded in the
//     child reference, so a visit is one aligned 64 B fetch instead of three 32 B fetches and the
//     second AABB test the shader performs when it pops a node is replaced by the entry distance
//     remembered from when it was pushed (same value, same comparison).
//   * The near child is descended into directly; only the far child goes to a stack whose first
//     kLdsStack entries live in LDS as [entry][lane] (conflict-free: each lane owns a column,
//     8-byte entries) and whose rare deeper entries spill to a per-wave HBM area. Entry 0 of every column
//     is a stopper that ends the traversal when popped, so the pop loop has no empty-stack case.
//   * Work-groups of four waves that run independently and share one thing: an LDS copy of the first
//     kTopLds node records (the top of the tree in breadth-first order, where a third of all visits go),
//     so that those visits cost LDS reads instead of trips through the vector L1.
//   * Persistent waves: the grid is sized to the machine, each wave pulls one 8x8-pixel tile at a
//     time from per-XCD queues (one returning atomic per tile; neighbouring tiles stay on one
//     XCD's L2; the queues hand out the tiles that were most expensive in earlier launches first),
//     and a lane that finishes a pixel takes the next pixel of the wave's tile at once.
//   * Wave-level scheduling by ballot: a lane is descending (interior node), at a leaf, or waiting
//     for the "feed" body (shade a finished traversal, hand out pixels, start paths); each round the
//     wave runs the body of the most populated group, so the long interior-node loop, the triangle
//     tests and the material code each run with many lanes instead of every round executing all
//     three for a few lanes each.
//   * One memory round trip per body: a leaf's triangles are requested together, and the feed body
//     requests the hit triangle, the first noise footprint and the sky footprint before it uses any.
//   * Per-frame work is hoisted: the pixel jitter (a noise lookup that depends on the frame's seed
//     only) is computed once per frame by prepare_frames_kernel; the material table sits in LDS.
//   * Lanes are persistent over frames too: the running mean of a pixel stays in registers for all
//     frames of a launch; the accumulation image is read and written once per launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "exact_div.h"
#include "rt_device.h"

using namespace rtd;

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
// GLSL built-ins as llvmpipe lowers them
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

__device__ __forceinline__ V3 normalize3(V3 v) { float r = rcp_exact(sqrt_exact(dot3(v, v))); return v * r; }
__device__ __forceinline__ V3 mix3(V3 a, V3 b, float t) { return a + t * (b - a); }
__device__ __forceinline__ V3 reflect3(V3 I, V3 N) { float d = dot3(N, I); return I - (2.0f * d) * N; }
__device__ __forceinline__ V3 refract3(V3 I, V3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * (eta * (1.0f - d * d));
    if (k < 0.0f) return mk(0.0f, 0.0f, 0.0f);
    return eta * I - (eta * d + sqrt_exact(k)) * N;
}
__device__ __forceinline__ V3 abs3(V3 a) { return mk(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }

__device__ __forceinline__ uint32_t Hash(uint32_t x) {      // F:96-105
    x ^= x >> 17; x *= 0xed5ad4bbU; x ^= x >> 11; x *= 0xac4c1b51U; x ^= x >> 15; x *= 0x31848babU; x ^= x >> 14;
    return x;
}
__device__ __forceinline__ uint32_t Hashf(float f) { return Hash(__float_as_uint(f)); }

// Hash(i) and Hash(i + 69) for the bounce counters a path can reach in practice (F:392-393): two full-rate-quarter
// multiplies and eight other instructions each, replaced by one 8-byte load from a table that stays in the vector L1.
constexpr uint32_t HashConst(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbU; x ^= x >> 11; x *= 0xac4c1b51U; x ^= x >> 15; x *= 0x31848babU; x ^= x >> 14;
    return x;
}
constexpr uint32_t kHashedBounces = 64;
struct BounceHashes {
    uint32_t v[2 * kHashedBounces];
    constexpr BounceHashes() : v() {
        for (uint32_t i = 0; i < kHashedBounces; ++i) { v[2 * i] = HashConst(i); v[2 * i + 1] = HashConst(i + 69u); }
    }
};
__constant__ BounceHashes kBounceHashes = BounceHashes();

// Products of two numbers below 2^24 (texel rows x texture width, triangle index x record size, tile quotients,
// 8-bit channels x weights): the 24-bit multiplier gives the same 32-bit result as v_mul_lo_u32 at a quarter of its
// issue cost.
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }

// ---- llvmpipe's default LINEAR filter on 8-bit formats: 8.8 fixed-point coordinates, rounded
// 8-bit lerps, x first then y, result scaled by 1/255 ----
// The lerp is a + (((b - a) * w + 128) >> 8) with an arithmetic shift. Adding a * 256 inside the shift gives the
// same value for all a, b, w in 0..255 without a signed intermediate: (a * (256 - w) + b * w + 128) >> 8, at most
// 255. In that form two channels that sit 16 bits apart in a register are filtered by the same two multiply-adds:
// every field stays below 2^16, nothing carries into its neighbour.
__device__ __forceinline__ uint32_t lerp8(uint32_t a, uint32_t b, uint32_t w, uint32_t iw /* 256 - w */) {
    return (mul24(b, w) + (mul24(a, iw) + 128u)) >> 8;
}
__device__ __forceinline__ uint32_t lerp8x2(uint32_t a, uint32_t b, uint32_t w, uint32_t iw) {      // fields at bits 0..7 and 16..23
    return ((mul24(b, w) + (mul24(a, iw) + 0x00800080u)) >> 8) & 0x00ff00ffu;
}

__device__ __forceinline__ uint32_t bilerp_rgbx(uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, int wx, int wy, int ch) {
    const uint32_t sh = (uint32_t)ch * 8u, ux = (uint32_t)wx, uy = (uint32_t)wy, ix = 256u - ux, iy = 256u - uy;
    const uint32_t a = (t00 >> sh) & 255u, b = (t10 >> sh) & 255u, c = (t01 >> sh) & 255u, d = (t11 >> sh) & 255u;
    return lerp8(lerp8(a, b, ux, ix), lerp8(c, d, ux, ix), uy, iy);
}

__device__ __forceinline__ void coord_repeat(float c, int n, int& i0, int& i1, int& w) {
    float f = c - floorf(c);
    f = f * (float)n;
    f = f * 256.0f;
    int fixed = (int)rintf(f) - 128;
    w = fixed & 255;
    i0 = fixed >> 8;
    if (i0 < 0) i0 = n - 1;
    if (i0 > n - 1) i0 = n - 1;
    i1 = (i0 == n - 1) ? 0 : i0 + 1;
}

// A bilinear footprint: the four texels and the two 8-bit weights. Fetching and filtering are
// separate so that a caller can request several footprints before it waits for any of them.
struct Texels { uint32_t t00, t10, t01, t11; int wx, wy; };


// n / d and n % d for n < 2^24 and a divisor d that is the same in every lane, with rd = 1.0f / float(d): the float
// product is within one of the quotient (its error is at most n/d * 2^-23 <= 1 for d >= 2, none for d = 1), and
// one correction step makes it exact. About 12 instructions against the 25 of the compiler's general sequence, four
// of which are full 32-bit multiplies.
__device__ __forceinline__ void divmod_uniform(uint32_t n, uint32_t d, float rd, uint32_t& q, uint32_t& r) {
    uint32_t q0 = (uint32_t)((float)n * rd);
    int32_t rem = (int32_t)(n - mul24(q0, d));
    if (rem < 0) { q0 -= 1u; rem += (int32_t)d; }
    else if (rem >= (int32_t)d) { q0 += 1u; rem -= (int32_t)d; }
    q = q0; r = (uint32_t)rem;
}

__device__ __forceinline__ float resolve(const Texels& q, int ch) {
    return (float)bilerp_rgbx(q.t00, q.t10, q.t01, q.t11, q.wx, q.wy, ch) * (float)(1.0 / 255.0);
}
// All three channels of a footprint: red and blue together (bits 0..7 and 16..23 of a texel), green on its own.
__device__ __forceinline__ V3 resolve3(const Texels& q) {
    const uint32_t ux = (uint32_t)q.wx, uy = (uint32_t)q.wy, ix = 256u - ux, iy = 256u - uy, rb = 0x00ff00ffu;
    const uint32_t prb = lerp8x2(lerp8x2(q.t00 & rb, q.t10 & rb, ux, ix), lerp8x2(q.t01 & rb, q.t11 & rb, ux, ix), uy, iy);
    const uint32_t pg = lerp8(lerp8((q.t00 >> 8) & 255u, (q.t10 >> 8) & 255u, ux, ix), lerp8((q.t01 >> 8) & 255u, (q.t11 >> 8) & 255u, ux, ix), uy, iy);
    const float k = (float)(1.0 / 255.0);
    return mk((float)(prb & 255u) * k, (float)pg * k, (float)(prb >> 16) * k);
}

__device__ __forceinline__ Texels fetch_noise(const SceneParams& sp, float s, float t) {   // texture(u_RgbNoise, ..)
    int x0, x1, y0, y1;
    Texels q;
    coord_repeat(s, sp.noiseW, x0, x1, q.wx);
    coord_repeat(t, sp.noiseH, y0, y1, q.wy);
    const uint32_t* T = sp.noise;
    const uint32_t r0 = mul24((uint32_t)y0, (uint32_t)sp.noiseW), r1 = mul24((uint32_t)y1, (uint32_t)sp.noiseW);
    q.t00 = T[r0 + (uint32_t)x0]; q.t10 = T[r0 + (uint32_t)x1];
    q.t01 = T[r1 + (uint32_t)x0]; q.t11 = T[r1 + (uint32_t)x1];
    return q;
}
__device__ __forceinline__ V3 sample_noise(const SceneParams& sp, float s, float t) { return resolve3(fetch_noise(sp, s, t)); }

__device__ __forceinline__ void coord_clamp(float c, int n, int& i0, int& i1, int& w) {
    float f = c * (float)n;
    f = f * 256.0f;
    int fixed = (int)rintf(f) - 128;
    w = fixed & 255;
    i0 = fixed >> 8;
    i1 = i0 + 1;
    i0 = min(max(i0, 0), n - 1);
    i1 = min(max(i1, 0), n - 1);
}

// texture(u_Skybox, RotateAroundAxis(dir, (1,0,0), -3.1415/2)) : F:158-163,406,411
__device__ __forceinline__ Texels fetch_sky(const SceneParams& sp, V3 dir) {
    // Rodrigues' formula with the axis (1,0,0) written out: v*cos + (k x v)*sin + k*dot(k,v)*(1-cos). The terms that
    // are products with the axis' zeros only add signed zeros, which can change nothing but the sign of a zero
    // component, and the cube lookup does not distinguish those. With the axis a constant, the shader compiler in
    // front of llvmpipe factors the x component, dir.x * cos + dir.x * (1 - cos), into dir.x * (cos + (1 - cos)):
    // the two differ in the last bit now and then, and the factor comes ready from the host.
    V3 d = mk(dir.x * sp.skyXScale, dir.y * sp.skyCos + (-dir.z) * sp.skySin, dir.z * sp.skyCos + dir.y * sp.skySin);
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int axis = (az >= fmaxf(ax, ay)) ? 2 : (ax > ay ? 0 : 1);
    float ma = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
    bool neg = (__float_as_uint(ma) >> 31) != 0;
    float sc, tc;
    if (axis == 0) { sc = neg ? d.z : -d.z; tc = -d.y; }
    else if (axis == 1) { sc = d.x; tc = neg ? -d.z : d.z; }
    else { sc = neg ? -d.x : d.x; tc = -d.y; }
    float ima = 0.5f * rcp_exact(fabsf(ma));      // = 0.5f / |ma| bit for bit: the halving is exact
    float s = sc * ima + 0.5f, t = tc * ima + 0.5f;
    int x0, x1, y0, y1;
    Texels q;
    const int S = sp.skySize;
    coord_clamp(s, S, x0, x1, q.wx);
    coord_clamp(t, S, y0, y1, q.wy);
    const uint32_t* T = sp.sky + (size_t)(axis * 2 + (neg ? 1 : 0)) * S * S;
    const uint32_t r0 = mul24((uint32_t)y0, (uint32_t)S), r1 = mul24((uint32_t)y1, (uint32_t)S);
    q.t00 = T[r0 + (uint32_t)x0]; q.t10 = T[r0 + (uint32_t)x1]; q.t01 = T[r1 + (uint32_t)x0]; q.t11 = T[r1 + (uint32_t)x1];
    return q;
}

// Whether `p` holds in any of the lanes that execute the call; used only to skip memory requests no lane needs, so an
// answer of "yes" is safe whenever some lane hit something. (The tests' host build of this file runs lanes one by one and
// cannot tell which lanes share a divergent branch: there every lane answers for itself.)
__device__ __forceinline__ bool any_lane(bool p) {
#ifdef RT_HOST_EMULATION
    return p;
#else
    return __ballot(p) != 0ull;
#endif
}

// Moves a value that is equal in all lanes into a scalar register.
__device__ __forceinline__ float uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ V3 uni3(V3 v) { return mk(uni(v.x), uni(v.y), uni(v.z)); }

struct Ray { V3 o, d, inv; };
__device__ __forceinline__ Ray MakeRay(V3 o, V3 d) {            // F:135-142 (magnitude is always 1)
    Ray r; r.o = o; r.d = d; r.inv = mk(rcp_exact(d.x), rcp_exact(d.y), rcp_exact(d.z)); return r;
}

// F:242-263. Returns hit flag, entry distance in tmin.
__device__ __forceinline__ bool hit_box(const Ray& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float& tmin) {
    float t0x = (mnx - r.o.x) * r.inv.x, t0y = (mny - r.o.y) * r.inv.y, t0z = (mnz - r.o.z) * r.inv.z;
    float t1x = (mxx - r.o.x) * r.inv.x, t1y = (mxy - r.o.y) * r.inv.y, t1z = (mxz - r.o.z) * r.inv.z;
    float lo = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float hi = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    tmin = lo;
    // max(lo, 0) as the one v_max_f32 it is: written as fmaxf the compiler first re-quiets `lo` with a second
    // v_max_f32, which the three-operand maximum that produced it has already done.
    float lo0;
#ifdef RT_HOST_EMULATION      // (the tests' host build of this file: same value, lo being a maximum of minima, NaN only if all are)
    lo0 = fmaxf(lo, 0.0f);
#else
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(lo0) : "v"(lo));
#endif
    return hi >= lo0;
}


struct Tally { uint32_t rays, pops, boxes, tris, noise, sky, paths, hits; uint32_t rounds[4]; uint32_t lanes[4]; };
// Wave time per scheduler state (counting builds only); lane 0 carries the sums.
struct RoundClock {
    unsigned long long t0, cycles[4];
    __device__ __forceinline__ void start() { t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void stop(int state) { cycles[state] += __builtin_readcyclecounter() - t0; }
};

enum LaneState : int { kIdle = 0, kNewPath = 1, kTraverse = 2 };
// What a traversing lane stands on is read from its reference, Lane::cur: an interior record's byte offset (bit 31
// clear, below kNoRef), a leaf (bit 31 set), or kNoRef once the traversal is over and the lane waits to be shaded.
// Lanes in the other two states carry kNoRef as well. The traversal code therefore never writes Lane::state.
__device__ __forceinline__ bool at_interior(uint32_t cur) { return cur < kNoRef; }
__device__ __forceinline__ bool at_leaf(uint32_t cur) { return (int32_t)cur < 0; }
__device__ __forceinline__ bool awaits_shading(int state, uint32_t cur) { return state == kTraverse && cur == kNoRef; }

#ifndef RT_LDS_STACK
#define RT_LDS_STACK 12
#endif
#ifndef RT_INT_STEPS
#define RT_INT_STEPS 8
#endif
#ifndef RT_KEEP_NUM
#define RT_KEEP_NUM 3
#endif
#ifndef RT_KEEP_DEN
#define RT_KEEP_DEN 4
#endif
#ifndef RT_LEAF_AT
#define RT_LEAF_AT 64      // run the leaf code as soon as this many lanes wait for it (64 = only when it is the biggest group)
#endif
#ifndef RT_SHADE_AT
#define RT_SHADE_AT 24
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 5
#endif
#ifndef RT_DOUBLE_STEP
#define RT_DOUBLE_STEP 1
#endif
#ifndef RT_WG_WAVES
#define RT_WG_WAVES 4      // waves per work-group: they share one LDS copy of the top of the tree, nothing else
#endif
#ifndef RT_TOP_LDS
#define RT_TOP_LDS 104     // interior records (the first ones: top of the tree, breadth-first) each work-group keeps in LDS
#endif
constexpr int kLdsStack = RT_LDS_STACK;   // stack entries per lane kept in LDS (512 B per entry per wave)
constexpr int kWgWaves = RT_WG_WAVES;
constexpr uint32_t kTopLds = RT_TOP_LDS;
// 20 waves per CU (5 on each SIMD) must fit the 160 KiB of LDS, which is handed out in granules of 1 280 bytes:
// 128 granules per CU, 25 per work-group of four waves = 32 000 bytes. One granule more and only four work-groups
// fit (measured: 7.64 instead of 6.97 ms per frame).
constexpr size_t kWavesPerCu = 4 * RT_MIN_WAVES;
constexpr size_t kLdsGranule = 1280, kLdsPerGroup = (163840 / kLdsGranule) / (kWavesPerCu / kWgWaves) * kLdsGranule;
static_assert(kWavesPerCu % kWgWaves == 0, "work-groups must divide the waves of a CU");
static_assert((size_t)kWgWaves * kLdsStack * 512 + sizeof(MaterialRec) * kMaxMaterials + (size_t)kTopLds * sizeof(NodeRec) <= kLdsPerGroup,
              "LDS per work-group too large for the intended number of waves per CU");
static_assert(kTopLds <= kTopRecords, "the LDS copy must lie within the breadth-first part of the node array");

// Everything a lane carries between scheduler rounds.
struct Lane {
    int state;
    uint32_t slot;          // tile-major pixel slot of this shard (tile * 64 + lane-in-tile)
    uint32_t k;             // frame counter within the launch
    V3 previous;            // running mean of the pixel (F:455,487)
    V3 throughput;          // rayColour (F:382)
    Ray ray;
    int bounce;             // i of the bounce loop (F:384)
    uint32_t segments;      // traversals of the current path
    // traversal (F:265-326)
    uint32_t cur;           // node / leaf reference being visited
    int top;                // deferred stack entries
    float bestT; int bestIdx;
    int iters;              // iterationsCount
};
// The loop over RAY_COUNT rays of one pixel and frame (F:453-485), carried by a lane next to its Lane in kernels built for
// more than one ray.
struct PixelRays {
    uint32_t sample;        // i of that loop
    uint32_t did;           // iterationsDid
    V3 sum;                 // rgb
};

typedef uint32_t Words4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) Words4 LdsWords4;

struct StackRef {
    uint2* lds;                 // this lane's LDS column, stride 64 entries
    unsigned long long* spill;  // this wave's HBM area (wave-uniform): entry e of lane l at [(e - kLdsStack) * 64 + l]
    uint32_t lane;
    const LdsWords4* top;       // the work-group's LDS copy of the first records of `nodes`, in 16-byte units
    uint32_t topBytes;          // how many bytes of `nodes` that copy holds (wave-uniform)
};

// The two halves of the stack (LDS entries as uint2, spilled entries as one 64-bit word) are accessed with
// different value types on purpose: with equal types the compiler merges the LDS and the HBM access into one FLAT
// instruction on a selected pointer, which is slower than ds_read/ds_write and drags the LDS traffic through the
// vector memory path.
//
// Pops deferred nodes until one is still worth visiting (F:277-283 with the entry distance kept
// from push time). Leaves the next reference, or kNoRef when the traversal is over, in L.cur.
//
// The bottom entry of every lane's stack is a stopper, written once per launch: reference kNoRef with entry
// distance -inf, which no `>= bestT` test skips. Popping it ends the traversal, so the loop needs no test for an
// empty stack; a traversal starts with top = 1.
template <bool COUNT, bool ITERS>
__device__ __forceinline__ void pop_next(const StackRef& st, Lane& L) {
    for (;;) {
        --L.top;
        uint2 e = st.lds[min(L.top, kLdsStack - 1) * 64];
        if (L.top >= kLdsStack) {
            // (the empty asm keeps the address arithmetic of this rare case inside the branch: without it the
            // compiler maintains the spill index in the loop for every pop)
            uint32_t deep = (uint32_t)(L.top - kLdsStack);
            asm volatile("" : "+v"(deep));
            unsigned long long w = st.spill[deep * 64u + st.lane];
            e = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
        }
        if (ITERS) L.iters += (e.x != kNoRef);      // the stopper is not one of the shader's pops
        if (__uint_as_float(e.y) >= L.bestT) continue;
        L.cur = e.x;
        break;
    }
}

// Starts the traversal of L.ray: the root's own box test (the first pop of F:276-283).
template <bool COUNT, bool ITERS>
__device__ __forceinline__ void begin_traversal(const SceneParams& sp, Lane& L, Tally& tl) {
    L.bestT = INFINITY; L.bestIdx = -1; L.iters = 0; L.top = 1; L.cur = kNoRef;
    L.state = kTraverse;
    tl.rays++;
    if (sp.rootRef != kNoRef) {
        if (ITERS) L.iters = 1;
        float tr;
        if (COUNT) tl.boxes++;
        if (hit_box(L.ray, sp.rootMin[0], sp.rootMin[1], sp.rootMin[2], sp.rootMax[0], sp.rootMax[1], sp.rootMax[2], tr) && !(tr >= L.bestT))
            L.cur = sp.rootRef;
    }
}

// One interior node: both children from one 64-byte record, F:296-311. An interior reference is the
// record's byte offset, so the address is the (scalar) array base plus a 32-bit lane offset.
template <bool COUNT, bool ITERS>
__device__ __forceinline__ void interior_step(const SceneParams& sp, const StackRef& st, Lane& L, Tally& tl) {
    // The first records (the top of the tree, which every ray passes: a third of all visits go to the first 64) are
    // read from the work-group's LDS copy: LDS reads do not go through the texture-address unit and the vector L1
    // (30 % fewer L1 accesses per frame on the benchmark scene).
    float4 n0, n1, n2;
    uint2 refs;                                                       // 56 of the record's 64 bytes are read
#ifdef RT_PROBE_STEP
    unsigned long long probeBegin = 0;
    if (COUNT) probeBegin = __builtin_readcyclecounter();
#endif
    if (kTopLds > 0 && L.cur < st.topBytes) {
        // (an LDS-qualified pointer: these loads can only become ds_read, and the compiler cannot fold them and the
        // loads of the other branch into FLAT loads from a selected pointer)
        const LdsWords4* T = (const LdsWords4*)((const __attribute__((address_space(3))) char*)st.top + L.cur);
        const Words4 a = T[0], b = T[1], c = T[2], d = T[3];
        n0 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        n1 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
        n2 = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
        refs = make_uint2(d.x, d.y);
    } else {
        const float4* N = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sp.nodes) + L.cur);
        n0 = N[0]; n1 = N[1]; n2 = N[2];
        refs = *reinterpret_cast<const uint2*>(N + 3);
    }
    if (COUNT) tl.boxes += 2;
#ifdef RT_PROBE_STEP
    // Measurement build only (tools/step_probe.py): where the time of a node visit goes. One lane of a wave adds, in units
    // of 16 cycles and on top of three of the ordinary tallies, the time from the record request to its data (sky), the
    // whole visit (noise) and the number of visits (hits).
    unsigned long long probeData = 0;
    if (COUNT) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        probeData = __builtin_readcyclecounter();
    }
#endif
#ifdef RT_PROBE_TOP
    // Measurement build only (tools/top_probe.py): how many visits go to the first 32 / 64 / 128 / 256 / 1024 / 4096
    // records, added on top of six of the ordinary tallies (subtract the tallies of an ordinary build).
    if (COUNT) { tl.sky += (L.cur < 32u * 64u); tl.noise += (L.cur < 64u * 64u); tl.hits += (L.cur < 128u * 64u);
                 tl.paths += (L.cur < 256u * 64u); tl.tris += (L.cur < 1024u * 64u); tl.pops += (L.cur < 4096u * 64u); }
#endif
    float tL, tR;
    bool hL = hit_box(L.ray, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tL);
    bool hR = hit_box(L.ray, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tR);
    const uint32_t refL = refs.x, refR = refs.y;
    const bool okL = hL && tL < L.bestT, okR = hR && tR < L.bestT;
    // The reference pushes both children and pops the one pushed last: the left one when tL < tR.
    // Here the child visited next is taken directly and only the other one is deferred.
    const bool leftFirst = tL < tR;
    if (okL || okR) {
        // The far child's entry is written whether or not both children were hit: it only counts once `top` has
        // moved past it, which happens in the both-hit case alone. This spares a divergent branch around the write.
        const bool both = okL && okR;
        const uint32_t farRef = leftFirst ? refR : refL;
        const float farT = leftFirst ? tR : tL;
        if (L.top < kLdsStack) st.lds[L.top * 64] = make_uint2(farRef, __float_as_uint(farT));
        else if (both) st.spill[(uint32_t)(L.top - kLdsStack) * 64u + st.lane] = ((unsigned long long)__float_as_uint(farT) << 32) | farRef;
        L.top += both ? 1 : 0;
        if (ITERS) L.iters++;    // the immediate pop of the chosen child (its test passes: bestT is unchanged)
        const bool takeLeft = okL && (leftFirst || !okR);
        L.cur = takeLeft ? refL : refR;
    } else {
        pop_next<COUNT, ITERS>(st, L);
    }
#ifdef RT_PROBE_STEP
    if (COUNT) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long probeEnd = __builtin_readcyclecounter();
        if (st.lane == (uint32_t)__builtin_ctzll(__ballot(true))) {      // one lane of those taking part
            tl.sky += (uint32_t)((probeData - probeBegin) >> 4); tl.noise += (uint32_t)((probeEnd - probeBegin) >> 4); tl.hits += 1u;
        }
    }
#endif
}

// hitTriangle, F:207-234: the distance at which L.ray meets triangle (v0, e1, e2), accepted only if
// it passes the shader's four early returns (with its comparison operators: NaNs fall through them).
__device__ __forceinline__ bool hit_triangle(const Ray& r, float4 a, float4 b, float e2z, float& t) {
    V3 v0 = mk(a.x, a.y, a.z), e1 = mk(a.w, b.x, b.y), e2 = mk(b.z, b.w, e2z);
    V3 h = cross3(r.d, e2);
    float det = dot3(e1, h);
    float invDet = rcp_exact(det);
    V3 s = r.o - v0;
    const float dsh = dot3(s, h);
    float u = invDet * dsh;
    V3 q = cross3(s, e1);
    const float ddq = dot3(r.d, q);
    float v = invDet * ddq;
    t = invDet * dot3(e2, q);
    // The shader's `u + v` adds two products that share invDet; the shader compiler in front of llvmpipe factors it
    // into invDet * (dot(s,h) + dot(d,q)), which now and then differs in the last bit: that decides hits lying exactly
    // on an edge two triangles share. u and v on their own are compared as written.
    bool reject = (fabsf(det) < 1e-8f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (invDet * (dsh + ddq) > 1.0f) | (t < 0.0001f);
    return !reject;
}

// One leaf: F:285-295; triangles in order, strictly closer wins. The records of the first three
// triangles (the reference builds leaves of at most three) are requested together so that the
// leaf costs one memory round trip; indices past the leaf's end are clamped to its last triangle
// and their result ignored.
template <bool COUNT, bool ITERS>
__device__ __forceinline__ void leaf_step(const SceneParams& sp, const StackRef& st, Lane& L, Tally& tl) {
    const uint32_t first = L.cur & 0x07ffffffu, count = ((L.cur >> 27) & 15u) + 1u;
    const uint32_t last = first + count - 1u;
    // 36 of a triangle record's 48 bytes are read here (the material slot only matters for the accepted hit)
    const float4* T0 = reinterpret_cast<const float4*>(sp.tris + first);
    const float4* T1 = reinterpret_cast<const float4*>(sp.tris + min(first + 1u, last));
    const float4* T2 = reinterpret_cast<const float4*>(sp.tris + min(first + 2u, last));
    float4 a0 = T0[0], b0 = T0[1]; float c0 = T0[2].x;
    float4 a1 = T1[0], b1 = T1[1]; float c1 = T1[2].x;
    float4 a2 = T2[0], b2 = T2[1]; float c2 = T2[2].x;
    if (COUNT) tl.tris += count;
    float t;
    if (hit_triangle(L.ray, a0, b0, c0, t) && L.bestT > t) { L.bestT = t; L.bestIdx = (int)first; }
    if (hit_triangle(L.ray, a1, b1, c1, t) && count > 1u && L.bestT > t) { L.bestT = t; L.bestIdx = (int)first + 1; }
    if (hit_triangle(L.ray, a2, b2, c2, t) && count > 2u && L.bestT > t) { L.bestT = t; L.bestIdx = (int)first + 2; }
    for (uint32_t i = first + 3u; i <= last; ++i) {
        const float4* T = reinterpret_cast<const float4*>(sp.tris + i);
        if (hit_triangle(L.ray, T[0], T[1], T[2].x, t) && L.bestT > t) { L.bestT = t; L.bestIdx = (int)i; }
    }
    pop_next<COUNT, ITERS>(st, L);
}

// Runs before every launch of the tracer, on the same stream: fills in the per-frame pixel jitter (F:461: the noise
// lookup there depends on the frame's seed only, so it is done once per frame here instead of once per pixel) and
// clears the tile queue heads. A one-frame launch hands its record over as a kernel argument (`single`), so that
// the host neither copies nor waits; the records of a multi-frame launch are already in `frames`.
__global__ void prepare_frames_kernel(SceneParams sp, FrameParams* __restrict__ frames, uint32_t nFrames, FrameParams single, int useSingle,
                                      uint32_t* __restrict__ queueHeads /* [64] */) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 64u) queueHeads[k] = 0u;
    if (k >= nFrames) return;
    FrameParams f = useSingle ? single : frames[k];
    V3 ro = sample_noise(sp, f.seedY + 0.007f * 0.0f, f.seedZ + 0.007f * 0.0f);
    f.jitterX = ro.x;
    f.jitterY = ro.y;
    frames[k] = f;
}

// More than one ray per pixel and frame (the shader's compile-time RAY_COUNT, F:5,459-483): behind the nFrames frame
// records lies one record whose first word is the ray count R, and behind that the pixel jitter of ray i of frame k as
// float2 [k * R + i]: F:461 looks it up at u_RandSeed.yz + 0.007 * i, which depends on the frame and on i only. Runs after
// prepare_frames_kernel, on the same stream.
__global__ void prepare_rays_kernel(SceneParams sp, FrameParams* __restrict__ frames, uint32_t nFrames, uint32_t rayCount) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0u) reinterpret_cast<uint32_t*>(frames + nFrames)[0] = rayCount;
    if (j >= nFrames * rayCount) return;
    const uint32_t k = j / rayCount, i = j - k * rayCount;
    const FrameParams f = frames[k];
    const float off = 0.007f * (float)i;
    V3 ro = sample_noise(sp, f.seedY + off, f.seedZ + off);
    reinterpret_cast<float2*>(frames + nFrames + 1)[j] = make_float2(ro.x, ro.y);
}

// COUNT: full tallies; IDS: primary-hit ids instead of radiance (one frame); SINGLE: the launch
// renders one frame, whose parameters then live in scalar registers; ITERS: the shader's iterationsCount is
// kept (needed by the tallies, the id pass and the heat-map view of bounce limit 0 only); MULTI: several rays per pixel
// and frame (the ray count and the jitter table lie behind the frame records, see prepare_rays_kernel).
template <bool COUNT, bool IDS, bool SINGLE, bool ITERS, bool MULTI>
__global__ __launch_bounds__(64 * RT_WG_WAVES, RT_MIN_WAVES) void trace_kernel(SceneParams sp, const FrameParams* __restrict__ frames, uint32_t nFrames,
                                                   float4* __restrict__ accum, uint32_t* __restrict__ tally /* [slots][tallyStride] */,
                                                   int32_t* __restrict__ ids /* IDS only: [slot][3] */,
                                                   uint2* __restrict__ spillArea /* [gridDim.x][kMaxStack-kLdsStack][64] */,
                                                   uint32_t* __restrict__ queueHeads /* [8] next tile per band, zeroed before the launch */,
                                                   uint32_t nGroups /* tiles of this shard */, uint32_t tallyStride,
                                                   const uint32_t* __restrict__ order /* [nGroups] tile hand-out order, or NULL for 0,1,2.. */,
                                                   uint32_t* __restrict__ tileCost /* [nGroups] rays traced per tile are added here, or NULL */) {
    // The waves of a work-group run independently (each is its own persistent worker); they only share the LDS
    // copies of the material table and of the top of the tree, filled here before the one barrier of the kernel.
    __shared__ uint2 lds_stack[kWgWaves * kLdsStack * 64];
    __shared__ MaterialRec lds_mat[kMaxMaterials];
    __shared__ Words4 lds_top[kTopLds > 0 ? kTopLds * 4 : 1];
    const uint32_t lane = threadIdx.x & 63u, waveInGroup = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * (uint32_t)kWgWaves + waveInGroup;      // index into the per-wave arrays
    const uint32_t nTop = min(kTopLds, sp.nInterior);
    StackRef st;
    st.lds = lds_stack + waveInGroup * (uint32_t)(kLdsStack * 64) + lane;
    st.spill = reinterpret_cast<unsigned long long*>(spillArea) + (size_t)wave * (kMaxStack - kLdsStack) * 64;
    st.lane = lane;
    st.top = (const LdsWords4*)lds_top;
    st.topBytes = nTop * (uint32_t)sizeof(NodeRec);
    st.lds[0] = make_uint2(kNoRef, __float_as_uint(-INFINITY));      // the stopper below every traversal's entries

    // the material table is small and read at every hit: keep a copy in LDS
    for (uint32_t j = threadIdx.x; j < (uint32_t)(sizeof(MaterialRec) / 4 * kMaxMaterials); j += blockDim.x)
        reinterpret_cast<uint32_t*>(lds_mat)[j] = reinterpret_cast<const uint32_t*>(sp.materials)[j];
    for (uint32_t j = threadIdx.x; j < nTop * 4u; j += blockDim.x)
        lds_top[j] = reinterpret_cast<const Words4*>(sp.nodes)[j];
    __syncthreads();

    // Per-launch constants are the same in every lane: computed with vector instructions (there
    // is no scalar float unit), then moved to scalar registers so they cost no vector registers.
    const uint32_t BL = sp.bounceLimit;
    const V3 camPos = mk(sp.camPos[0], sp.camPos[1], sp.camPos[2]);
    const V3 facing = mk(sp.camFacing[0], sp.camFacing[1], sp.camFacing[2]);
    const float resx = uni((float)sp.width), resy = uni((float)sp.height);
    const float nresx = uni((float)sp.noiseW), nresy = uni((float)sp.noiseH);
    const float rTilesX = uni(1.0f / (float)sp.tilesX);
    const float rresx = uni(1.0f / resx), rresy = uni(1.0f / resy), rnresx = uni(1.0f / nresx), rnresy = uni(1.0f / nresy);
    const float vpH = uni(sp.vpW * (resy / resx));                              // F:145
    const V3 U = uni3(normalize3(cross3(facing, mk(0.0f, 0.0f, 1.0f))));        // F:147
    const V3 Vd = uni3(normalize3(cross3(U, facing)));                          // F:148
    const uint32_t nTiles = sp.tilesX * sp.tilesY;
    FrameParams f0 = {};
    if (SINGLE) f0 = frames[0];
    const uint32_t rayCount = MULTI ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t*>(frames + nFrames)[0]) : 1u;
    const float2* rayJitter = reinterpret_cast<const float2*>(frames + nFrames + 1);      // MULTI only

    // Tiles are queued in 8 bands of consecutive positions of the hand-out order (with the order the host
    // builds, a band is a contiguous eighth of the shard, most expensive tiles first); a wave drains the band
    // with its blockIdx % 8 label first (waves with equal labels share an XCD, hence an L2), then helps the
    // other bands. Which tile a wave gets when only affects speed: a pixel's result does not depend on it.
    const uint32_t bandSize = (nGroups + 7u) / 8u;
    uint32_t band = blockIdx.x % 8u, bandsTried = 0;
    uint32_t poolNext = 0, poolEnd = 0;       // wave-uniform: slots [poolNext, poolEnd) wait for a lane
    bool queueEmpty = false;

    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
    RoundClock rc = {0ull, {0ull, 0ull, 0ull, 0ull}};
    Lane L;
    L.state = kIdle; L.slot = 0; L.k = 0; L.bounce = 0; L.segments = 0; L.cur = kNoRef; L.top = 0;
    L.bestT = INFINITY; L.bestIdx = -1; L.iters = 0;
    L.previous = mk(0.f, 0.f, 0.f); L.throughput = mk(1.f, 1.f, 1.f);
    L.ray.o = L.ray.d = L.ray.inv = mk(0.f, 0.f, 0.f);
    V3 hitNormal = mk(0.0f, 0.0f, 0.0f);      // hitRecord.normal persists across bounces of a path (F:381)
    PixelRays P = {0u, 0u, mk(0.f, 0.f, 0.f)};

    for (;;) {
        // ---- pick what to run: the most populated of interior / leaf / feed (ballot + popcount, wave-uniform).
        // "Feed" is everything the shading round serves: lanes with a finished traversal, lanes about to start
        // a path, and idle lanes while pixels are left to hand out.
        const int nInt = __popcll(__ballot(at_interior(L.cur)));
        const int nLeaf = __popcll(__ballot(at_leaf(L.cur)));
        const bool pixelsLeft = (poolNext != poolEnd) || !queueEmpty;
        const int nFeed = __popcll(__ballot(L.cur == kNoRef && (pixelsLeft || L.state != kIdle)));
        if ((nInt | nLeaf | nFeed) == 0) break;      // every wave reaches this: the queues only shrink
        const int most = max(nInt, max(nLeaf, nFeed));
        const bool runFeed = nFeed >= RT_SHADE_AT, runLeaf = !runFeed && nLeaf >= RT_LEAF_AT;

        if (COUNT) rc.start();
#ifdef RT_PROBE_IDLE
        // Measurement build only: what the lanes that sit out an interior round are waiting for. Added on top of four
        // of the ordinary tallies by lane 0 (subtract the tallies of an ordinary build).
        if (COUNT && nInt == most && !runFeed && !runLeaf && lane == 0) {
            tl.sky += 1u; tl.noise += (uint32_t)nInt; tl.hits += (uint32_t)nLeaf; tl.paths += (uint32_t)nFeed;
        }
#endif
        if (nInt == most && !runFeed && !runLeaf) {
            // the hot loop: stay in it while most of the lanes that entered are still descending
            int steps = 0;
            bool descending = at_interior(L.cur);
            do {
                if (COUNT) { tl.rounds[1] += (lane == 0); tl.lanes[1] += descending; }
                if (descending) interior_step<COUNT, ITERS>(sp, st, L, tl);
                descending = at_interior(L.cur);
                ++steps;
#if RT_DOUBLE_STEP
                // a second node visit before the wave re-counts its lanes: half the loop control per visit
                if (COUNT) { tl.rounds[1] += (lane == 0); tl.lanes[1] += descending; }
                if (descending) interior_step<COUNT, ITERS>(sp, st, L, tl);
                descending = at_interior(L.cur);
                ++steps;
#endif
            } while (steps < RT_INT_STEPS && __popcll(__ballot(descending)) * RT_KEEP_DEN >= nInt * RT_KEEP_NUM);
            if (COUNT) rc.stop(1);
        } else if (runLeaf || (nLeaf == most && !runFeed)) {
            if (COUNT) { tl.rounds[2] += (lane == 0); tl.lanes[2] += at_leaf(L.cur); }
            if (at_leaf(L.cur)) leaf_step<COUNT, ITERS>(sp, st, L, tl);
            if (COUNT) rc.stop(2);
        } else {
            // ================= shading round: shade, hand out pixels, start paths =================
            if (COUNT) { tl.rounds[3] += (lane == 0); tl.lanes[3] += awaits_shading(L.state, L.cur); }
            if (awaits_shading(L.state, L.cur)) {
                if (COUNT) tl.pops += (uint32_t)L.iters;
                const bool hitAnything = L.bestIdx >= 0;
                bool done = false;          // path finished with colour `result`
                V3 result = mk(0.0f, 0.0f, 0.0f);
                FrameParams fp = {};
                if (IDS) {
                    ids[(size_t)L.slot * 3 + 0] = L.bestIdx; ids[(size_t)L.slot * 3 + 1] = (int32_t)__float_as_uint(L.bestT);
                    ids[(size_t)L.slot * 3 + 2] = L.iters;
                    L.state = kIdle;
                } else if (BL < 2u) {
                    // the two debug views
                    if (SINGLE) fp = f0; else fp = frames[L.k];
                    if (COUNT && hitAnything) tl.hits++;
                    if (hitAnything) {
                        const float4* T = reinterpret_cast<const float4*>(sp.tris + L.bestIdx);
                        float4 a = T[0], b4 = T[1], c = T[2];
                        V3 e1 = mk(a.w, b4.x, b4.y), e2 = mk(b4.z, b4.w, c.x);
                        V3 n = normalize3(cross3(e1, e2));
                        hitNormal = (dot3(L.ray.d, n) < 0.0f) ? n : -n;
                    }
                    if (BL == 0u) {
                        // BVH heat map, F:313-324,386-387
                        const int iters = L.iters;
                        const float itf = (float)iters;
                        if (iters < 64) { float g = (1.0f / 64.0f) * itf; result = mk(g, g, g); }
                        else if (iters < 128) { float kk = (float)(iters - 64); result = mk(1.0f - (0.0f / 64.0f) * kk, 1.0f - (1.0f / 64.0f) * kk, 1.0f - (1.0f / 64.0f) * kk); }
                        else { float kk = (float)(iters - 128); result = mk(1.0f - (1.0f / 64.0f) * kk, 0.0f - (0.0f / 64.0f) * kk, 0.0f - (0.0f / 64.0f) * kk); }
                    } else {
                        result = abs3(hitNormal);                       // F:388-389
                    }
                    done = true;
                } else {
                    if (SINGLE) fp = f0; else fp = frames[L.k];
                    if (COUNT && hitAnything) tl.hits++;
                    const int i = L.bounce;
                    // F:391-394
                    uint32_t hi, hj;
                    if (__builtin_expect((uint32_t)i < kHashedBounces, 1)) {
                        const uint2 h = *reinterpret_cast<const uint2*>(&kBounceHashes.v[2 * i]);
                        hi = h.x; hj = h.y;
                    } else {
                        hi = Hash((uint32_t)i); hj = Hash((uint32_t)(i + 69));
                    }
                    uint32_t sx = Hashf(L.ray.o.x + L.ray.d.y) + hi + fp.hashX;
                    uint32_t sy = Hashf(L.ray.o.y + L.ray.d.z) + hj + fp.hashY;
                    if (MULTI) {                 // Hash(iRayCountSeed): the frame records hold the term of ray 0
                        const uint32_t d = Hash(P.sample) - Hash(0u);
                        sx += d; sy += d;
                    }
                    float fsx = (float)sx, fsy = (float)sy;
                    float seedx = div_by_uniform(fsx - nresx * floorf(div_by_uniform(fsx, nresx, rnresx)), nresx, rnresx);
                    float seedy = div_by_uniform(fsy - nresy * floorf(div_by_uniform(fsy, nresy, rnresy)), nresy, rnresy);
                    // What every lane of the round reads first is requested here, before any of it is used:
                    // the hit triangle and the first noise footprint, or the sky footprint, share one memory
                    // round trip. All lanes take part in both groups of requests (a lane that missed reads
                    // triangle 0 and ignores it): no divergent branch separates request from use.
                    float ta_w = 0.0f, tc_x = 0.0f, tc_y = 0.0f;      // (initialised so that nothing is carried between rounds)
                    float4 tb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    Texels nA = {0u, 0u, 0u, 0u, 0, 0}, sk = {0u, 0u, 0u, 0u, 0, 0};
                    if (any_lane(hitAnything)) {
                        const TriRec* T = reinterpret_cast<const TriRec*>(reinterpret_cast<const char*>(sp.tris) +
                                                                          mul24(hitAnything ? (uint32_t)L.bestIdx : 0u, (uint32_t)sizeof(TriRec)));
                        ta_w = T->e1x; tb = *reinterpret_cast<const float4*>(&T->e1y); tc_x = T->e2z; tc_y = __int_as_float(T->material);
                        nA = fetch_noise(sp, seedx + 0.01534f, seedy + 0.183f);      // F:420
                    }
                    if (any_lane(!hitAnything)) sk = fetch_sky(sp, L.ray.d);  // F:410-411
                    if (!hitAnything) {
                        if (COUNT) tl.sky += (i == 0) ? 2u : 1u;
                        result = L.throughput * resolve3(sk);                        // F:410-413
                        done = true;
                    } else {
                        // the part of hitTriangle that only the accepted hit needs (F:235-238) + material (F:291)
                        V3 e1 = mk(ta_w, tb.x, tb.y), e2 = mk(tb.z, tb.w, tc_x);
                        V3 hitPoint = L.ray.o + (1.0f * L.bestT) * L.ray.d;
                        V3 n = normalize3(cross3(e1, e2));
                        const bool frontFace = dot3(L.ray.d, n) < 0.0f;
                        hitNormal = frontFace ? n : -n;
                        const MaterialRec m = *reinterpret_cast<const MaterialRec*>(reinterpret_cast<const char*>(lds_mat) +
                                                                                    mul24((uint32_t)__float_as_int(tc_y), (uint32_t)sizeof(MaterialRec)));
                        V3 hitColour = mk(m.colour[0], m.colour[1], m.colour[2]);
                        if (m.isLight) {
                            result = (L.throughput * hitColour) * m.transparency;   // F:415-418
                            done = true;
                        } else {
                            V3 next;
                            int ii = i;
                            if (COUNT) tl.noise++;
                            float trRng = resolve(nA, 0);
                            if (trRng < m.transparency) {
                                // TransparentScatter, F:328-336
                                const float AIR = 1.0003f;
                                float ri = frontFace ? m.airOverIndex : m.indexOverAir;      // AIR / n : n / AIR
                                float cos_theta = fminf(dot3(-L.ray.d, hitNormal), 1.0f);
                                float sin_theta = sqrt_exact(1.0f - cos_theta * cos_theta);
                                next = (ri * sin_theta <= 1.0f) ? refract3(L.ray.d, hitNormal, ri) : reflect3(L.ray.d, hitNormal);
                                if (trRng < 0.5f) ii--;
                            } else {
                                if (COUNT) tl.noise += 2;
                                const Texels nB = fetch_noise(sp, seedx - 0.062f, seedy - 0.0345f);       // F:426
                                const Texels nC = fetch_noise(sp, seedx, seedy);                          // F:428
                                V3 n3 = resolve3(nB);
                                V3 diffuseDir = normalize3(hitNormal + ((2.0f * n3) - mk(1.0f, 1.0f, 1.0f)) * m.roughness);
                                V3 specularDir = reflect3(L.ray.d, hitNormal);
                                bool isSpecular = resolve(nC, 1) < (1.0f - m.roughness);
                                specularDir = mix3(specularDir, diffuseDir, m.roughness);
                                next = isSpecular ? specularDir : diffuseDir;
                                V3 spec = mk(m.specular[0], m.specular[1], m.specular[2]);
                                L.throughput = L.throughput * (isSpecular ? mix3(spec, hitColour, m.metallic) : hitColour);
                            }
                            L.ray = MakeRay(hitPoint + 1e-4f * next, next);         // F:433
                            // adaptive early-out against the accumulated pixel, F:435-446
                            V3 diff = abs3(L.throughput - L.previous);
                            float distance = (diff.x + diff.y) + diff.z;
                            int inc = (int)(BL / 4u);
                            if (distance < 1.0f) ii += inc;
                            if (distance < 0.4f) ii += inc;
                            if (distance < 0.2f) ii += inc;
                            ii++;                                                   // ++i of the for statement
                            L.bounce = ii;
                            ++L.segments;
                            // F:384,448; plus a bound on free glass bounces so that every wave is sure to finish
                            if (!((uint32_t)ii <= BL) || L.segments >= kMaxSegments) { result = mk(0.0f, 0.0f, 0.0f); done = true; }
                            else begin_traversal<COUNT, ITERS>(sp, L, tl);
                        }
                    }
                }
                if (done) {
                    bool frameDone = true;
                    V3 rgb;
                    if (MULTI) {
                        // F:467-483: the loop over the pixel's rays skips up to three of them when a ray's colour is close
                        // to the sum of those before it
                        const V3 diff = abs3(P.sum - result);
                        const float distance = (diff.x + diff.y) + diff.z;
                        uint32_t i = P.sample;
                        if (i > 0u) {
                            if (distance < 0.5f) i += 1u;
                            if (distance < 0.2f) i += 1u;
                            if (distance < 0.1f) i += 1u;
                        }
                        P.sum = P.sum + result;
                        P.did += 1u;
                        i += 1u;
                        P.sample = i;
                        frameDone = !(i < rayCount);
                        rgb = P.sum * (1.0f / (float)P.did);                         // F:484-485
                    } else {
                        // F:480-487
                        rgb = mk(0.0f, 0.0f, 0.0f) + result;
                        rgb = rgb * (1.0f / 1.0f);
                    }
                    if (frameDone) L.previous = mix3(L.previous, rgb, fp.weight);
                    if (COUNT || MULTI) tl.paths++;
                    // what this path cost, for the next launches' hand-out order (no value returned: fire and forget)
                    if (tileCost) __hip_atomic_fetch_add(&tileCost[L.slot >> 6], L.segments + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (frameDone) {
                        if (MULTI) { P.sample = 0u; P.did = 0u; P.sum = mk(0.0f, 0.0f, 0.0f); }
                        ++L.k;
                    }
                    if (L.k >= nFrames) {
                        accum[L.slot] = make_float4(L.previous.x, L.previous.y, L.previous.z, 0.0f);
                        L.state = kIdle;
                    } else {
                        L.state = kNewPath;
                    }
                }
            }

            // ---- hand waiting pixels to idle lanes ----
            unsigned long long idleMask = __ballot(L.state == kIdle);
            if (idleMask != 0ull) {
                if (poolNext == poolEnd && !queueEmpty) {
                    // pull the next group of tiles: one returning atomic by lane 0, broadcast to the wave
                    for (;;) {
                        uint32_t g = 0;
                        if (lane == 0) g = atomicAdd(&queueHeads[band], 1u);
                        g = __builtin_amdgcn_readfirstlane(g);
                        const uint32_t first = band * bandSize;
                        const uint32_t last = min(first + bandSize, nGroups);
                        if (first + g < last) {
                            uint32_t t = first + g;
                            if (order) t = order[t];
                            poolNext = t * 64u;
                            poolEnd = poolNext + 64u;
                            break;
                        }
                        band = (band + 1u) % 8u;
                        if (++bandsTried >= 8u) { queueEmpty = true; break; }
                    }
                }
                if (poolNext != poolEnd && L.state == kIdle) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
                    const uint32_t s = poolNext + rank;
                    if (s < poolEnd) {
                        const uint32_t tile = (s >> 6) * sp.shardWorld + sp.shardRank;
                        const uint32_t inTile = s & 63u;
                        uint32_t tileX, tileY;
                        divmod_uniform(tile, sp.tilesX, rTilesX, tileY, tileX);
                        const uint32_t px = tileX * kTile + (inTile & 7u), py = tileY * kTile + (inTile >> 3);
                        if (tile < nTiles && px < sp.width && py < sp.height) {
                            L.slot = s; L.k = 0;
                            if (MULTI) { P.sample = 0u; P.did = 0u; P.sum = mk(0.0f, 0.0f, 0.0f); }
                            float4 acc = IDS ? make_float4(0.f, 0.f, 0.f, 0.f) : accum[s];
                            L.previous = mk(acc.x, acc.y, acc.z);
                            L.state = kNewPath;
                        }
                    }
                }
                if (poolNext != poolEnd) poolNext = min(poolEnd, poolNext + (uint32_t)__popcll(idleMask));
            }

            // ---- start paths: F:461-462 (pixel jitter precomputed per frame), primary ray ----
            if (COUNT) { tl.rounds[0] += (lane == 0); tl.lanes[0] += (L.state == kNewPath); }
            if (L.state == kNewPath) {
                float jx, jy;
                if (MULTI) { const float2 j = rayJitter[mul24(L.k, rayCount) + P.sample]; jx = j.x; jy = j.y; }
                else if (SINGLE) { jx = f0.jitterX; jy = f0.jitterY; } else { jx = frames[L.k].jitterX; jy = frames[L.k].jitterY; }
                const uint32_t tile = (L.slot >> 6) * sp.shardWorld + sp.shardRank, inTile = L.slot & 63u;
                uint32_t tileX, tileY;
                divmod_uniform(tile, sp.tilesX, rTilesX, tileY, tileX);
                const float fx = (float)(tileX * kTile + (inTile & 7u)) + 0.5f;
                const float fy = (float)(tileY * kTile + (inTile >> 3)) + 0.5f;
                if (COUNT) tl.noise++;
                V3 pt = ((camPos + sp.vpD * facing) + (sp.vpW * (-0.5f + div_by_uniform((fx - 0.5f) + jx, resx, rresx))) * U) +
                        (vpH * (-0.5f + div_by_uniform((fy - 0.5f) + jy, resy, rresy))) * Vd;
                L.ray = MakeRay(camPos, normalize3(pt - camPos));
                L.throughput = mk(1.0f, 1.0f, 1.0f);
                L.bounce = 0; L.segments = 0;
                // A new invocation starts with a blank hit record. (The shader never initialises it, F:381, and its normal
                // view shows the record's normal even after a miss, F:389: zeros on the parity target for a pixel's first ray,
                // and for its later rays what the previous ray left there.)
                if (!MULTI || P.sample == 0u) hitNormal = mk(0.0f, 0.0f, 0.0f);
                begin_traversal<COUNT, ITERS>(sp, L, tl);
            }
            if (COUNT) rc.stop(3);
        }
    }

    // per-wave tallies: each wave owns one column (no atomics)
    uint32_t vals[kCounterSlots] = {tl.rays, tl.pops, tl.boxes, tl.tris, tl.noise, tl.sky, tl.paths, tl.hits,
                                    tl.rounds[0], tl.rounds[1], tl.rounds[2], tl.rounds[3], tl.lanes[0], tl.lanes[1], tl.lanes[2], tl.lanes[3],
                                    lane == 0 ? (uint32_t)(rc.cycles[0] >> 6) : 0u, lane == 0 ? (uint32_t)(rc.cycles[1] >> 6) : 0u,
                                    lane == 0 ? (uint32_t)(rc.cycles[2] >> 6) : 0u, lane == 0 ? (uint32_t)(rc.cycles[3] >> 6) : 0u};
    for (int s = 0; s < (int)kCounterSlots; ++s) {
        if (!COUNT && s != (int)kRays && !(MULTI && s == (int)kPaths)) continue;      // (with several rays per pixel the host cannot count paths itself)
        uint32_t v = vals[s];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) tally[(size_t)s * tallyStride + wave] += v;
    }
}

// Sort keys for the tile hand-out order: band first (a contiguous eighth of the shard, so that neighbouring
// tiles stay on one XCD), then descending cost. Clears the cost array for the next period.
__global__ void order_keys_kernel(uint32_t* __restrict__ tileCost, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                  uint32_t nTiles, uint32_t bandSize) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    const uint32_t cost = min(tileCost[i], 0x3ffffffu);
    keys[i] = ((i / bandSize) << 26) | (0x3ffffffu - cost);
    vals[i] = i;
    tileCost[i] = 0u;
}

// Sums the per-tile tallies into 64-bit totals: out[slot] += sum(tally[slot][*]); clears tally.
__global__ void reduce_tally_kernel(uint32_t* __restrict__ tally, uint32_t nTiles, unsigned long long* __restrict__ out) {
    const int slot = blockIdx.x;
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < nTiles; i += blockDim.x) {
        s += tally[(size_t)slot * nTiles + i];
        tally[(size_t)slot * nTiles + i] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ unsigned long long part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) t += part[w];
        out[slot] += t;
    }
}

// Tile-major <-> row-major conversion of the accumulation image.
__global__ void untile_kernel(const float4* __restrict__ tiled, float4* __restrict__ linear, uint32_t width, uint32_t height,
                              uint32_t tilesX, uint32_t world, uint32_t shardTiles, int toTiled) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    uint32_t tile = (y / kTile) * tilesX + x / kTile, lane = (y % kTile) * kTile + x % kTile;
    // gathered layout: rank-major, then local tile
    size_t slot = ((size_t)(tile % world) * shardTiles + tile / world) * 64 + lane;
    if (toTiled) const_cast<float4*>(tiled)[slot] = linear[(size_t)y * width + x];
    else linear[(size_t)y * width + x] = tiled[slot];
}

// Gathered shards (rank-major: shard r holds tiles r, r + world, ...) -> the tile-major image of an unsharded context:
// whole 1 KiB tiles change place, nothing inside a tile moves.
__global__ void retile_kernel(const float4* __restrict__ gathered, float4* __restrict__ tiled, uint32_t nTiles, uint32_t world, uint32_t shardTiles) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = i >> 6, lane = i & 63u;
    if (tile >= nTiles) return;
    tiled[i] = gathered[((size_t)(tile % world) * shardTiles + tile / world) * 64u + lane];
}

// The tracer's ray/triangle test on independent (triangle, ray) pairs, one pair per lane: the same hit_triangle, fed
// with the same 48-byte records, that the leaf body runs. hitT[i] = bit pattern of t on a hit, 0xffffffff on a miss.
__global__ void probe_triangles_kernel(const TriRec* __restrict__ tris, const float* __restrict__ o3, const float* __restrict__ d3, uint32_t n,
                                       uint32_t* __restrict__ hitT) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* T = reinterpret_cast<const float4*>(tris + i);
    const Ray r = MakeRay(mk(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), mk(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]));
    float t = 0.0f;
    const bool hit = hit_triangle(r, T[0], T[1], T[2].x, t);
    hitT[i] = hit ? __float_as_uint(t) : 0xffffffffu;
}

}  // namespace


// ---- launchers used by rt_api.cpp ----
namespace rtd {

uint32_t SpillEntriesPerWave() { return (uint32_t)(kMaxStack - kLdsStack) * 64u; }

// Number of waves that can be resident at once on the current device (the persistent grid size).
uint32_t ResidentWaves() {
    int dev = 0, cus = 0, perCu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, trace_kernel<false, false, true, false, false>, 64 * kWgWaves, 0) != hipSuccess) return 0;
    if (perCu < 1) perCu = 1;
    return (uint32_t)cus * (uint32_t)perCu * (uint32_t)kWgWaves;
}

void LaunchPrepareFrames(hipStream_t st, const SceneParams& sp, FrameParams* frames, uint32_t nFrames, const FrameParams* single, uint32_t* queueHeads) {
    hipLaunchKernelGGL(prepare_frames_kernel, dim3((nFrames + 63u) / 64u), dim3(64), 0, st, sp, frames, nFrames, single ? *single : FrameParams{},
                       single ? 1 : 0, queueHeads);
}

void LaunchPrepareRays(hipStream_t st, const SceneParams& sp, FrameParams* frames, uint32_t nFrames, uint32_t rayCount) {
    const uint32_t n = nFrames * rayCount;
    hipLaunchKernelGGL(prepare_rays_kernel, dim3((n + 63u) / 64u), dim3(64), 0, st, sp, frames, nFrames, rayCount);
}

// multi: more than one ray per pixel and frame; `frames` then carries the ray count and the jitter table behind its
// nFrames records (LaunchPrepareRays).
void LaunchTrace(bool count, bool ids, bool multi, hipStream_t st, uint32_t nWaves, const SceneParams& sp, const FrameParams* frames, uint32_t nFrames,
                 void* accum, uint32_t* tally, uint32_t tallyStride, int32_t* idsOut, void* spillArea, uint32_t* queueHeads, uint32_t nGroups,
                 const uint32_t* order, uint32_t* tileCost) {
    float4* a = static_cast<float4*>(accum);
    uint2* sa = static_cast<uint2*>(spillArea);
    dim3 grid((nWaves + (uint32_t)kWgWaves - 1u) / (uint32_t)kWgWaves), block(64 * kWgWaves);     // the per-wave arrays hold a multiple of kWgWaves
    const bool single = nFrames == 1u;
    const bool heat = sp.bounceLimit == 0u;        // the only view that shows iterationsCount
#define RT_LAUNCH(C, I, S, T, M) hipLaunchKernelGGL((trace_kernel<C, I, S, T, M>), grid, block, 0, st, sp, frames, nFrames, a, tally, idsOut, sa, queueHeads, nGroups, tallyStride, order, tileCost)
    if (ids) RT_LAUNCH(true, true, true, true, false);
    else if (multi) {
        if (count && single) RT_LAUNCH(true, false, true, true, true);
        else if (count) RT_LAUNCH(true, false, false, true, true);
        else if (single) RT_LAUNCH(false, false, true, true, true);       // (the views that show iterationsCount included)
        else RT_LAUNCH(false, false, false, true, true);
    }
    else if (count && single) RT_LAUNCH(true, false, true, true, false);
    else if (count) RT_LAUNCH(true, false, false, true, false);
    else if (single && heat) RT_LAUNCH(false, false, true, true, false);
    else if (single) RT_LAUNCH(false, false, true, false, false);
    else if (heat) RT_LAUNCH(false, false, false, true, false);
    else RT_LAUNCH(false, false, false, false, false);
#undef RT_LAUNCH
}

void LaunchOrderKeys(hipStream_t st, uint32_t* tileCost, uint32_t* keys, uint32_t* vals, uint32_t nTiles) {
    const uint32_t bandSize = (nTiles + 7u) / 8u;
    hipLaunchKernelGGL(order_keys_kernel, dim3((nTiles + 255u) / 256u), dim3(256), 0, st, tileCost, keys, vals, nTiles, bandSize);
}

void LaunchReduceTally(hipStream_t st, uint32_t* tally, uint32_t nTiles, unsigned long long* out, int slots) {
    hipLaunchKernelGGL(reduce_tally_kernel, dim3(slots), dim3(256), 0, st, tally, nTiles, out);
}

void LaunchUntile(hipStream_t st, const void* tiled, void* linear, uint32_t width, uint32_t height, uint32_t tilesX, uint32_t world,
                  uint32_t shardTiles, bool toTiled) {
    dim3 block(32, 8), grid((width + 31) / 32, (height + 7) / 8);
    hipLaunchKernelGGL(untile_kernel, grid, block, 0, st, static_cast<const float4*>(tiled), static_cast<float4*>(linear), width, height,
                       tilesX, world, shardTiles, toTiled ? 1 : 0);
}

void LaunchRetile(hipStream_t st, const void* gathered, void* tiled, uint32_t nTiles, uint32_t world, uint32_t shardTiles) {
    hipLaunchKernelGGL(retile_kernel, dim3((nTiles * 64u + 255u) / 256u), dim3(256), 0, st, static_cast<const float4*>(gathered), static_cast<float4*>(tiled),
                       nTiles, world, shardTiles);
}

void LaunchProbeTriangles(hipStream_t st, const void* tris, const float* o3, const float* d3, uint32_t n, uint32_t* hitT) {
    hipLaunchKernelGGL(probe_triangles_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, static_cast<const TriRec*>(tris), o3, d3, n, hitT);
}

}  // namespace rtd
