// Device-side data layout shared by the HIP kernels and the host side of the C ABI.
//
// HBM layout (all arrays 256-byte aligned by hipMalloc):
//   nodes   NodeRec[nInterior] 64 B per INTERIOR node, in the reference's pre-order with the leaves
//                            left out (so a node's left child, when interior, is the next record).
//                            A record carries the boxes of BOTH children plus what each child is,
//                            so one 64-byte fetch (4 x dwordx4, one cache line half) replaces the
//                            reference's three 32-byte fetches per visited node (node + two
//                            children, Fragment.glsl:278,298-299) and a leaf is never fetched at
//                            all: its triangle range travels in the child ref.
//   tris    TriRec[nTri]     48 B per triangle in leaf order: v0, e1 = v1-v0, e2 = v2-v0 (the
//                            same fp32 subtractions the shader does per test, done once), and
//                            the material slot.
//   noise   uint32[w*h]      RGBX8 texels (RGB8 source widened so one dword load = one texel)
//   sky     uint32[6*s*s]    RGBA8 texels, faces +X,-X,+Y,-Y,+Z,-Z
//   accum   float4[tiles*64] accumulation image in 8x8-pixel tile-major order: pixel (x,y) lives
//                            at tile (y/8)*tilesX + x/8, lane (y%8)*8 + x%8, so a wave reads and
//                            writes one contiguous 1 KiB block, and a rank's shard is contiguous.
#pragma once

#include <stdint.h>

namespace rtd {

constexpr int kTile = 8;                 // 8x8 pixels = one wave64
constexpr int kMaxMaterials = 16;        // Fragment.glsl:12
constexpr int kMaxStack = 64;            // Fragment.glsl:9
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoRef = 0x7fffffffu; // "no node"
// A transparent hit may leave the bounce counter unchanged (Fragment.glsl:421-424), so a path has no
// hard length limit in the reference. A path is cut (contributing black) after this many ray
// segments; the parity oracle applies the same bound.
constexpr uint32_t kMaxSegments = 4096;
// Interior records that come first in `nodes`, in breadth-first order from the root.
constexpr uint32_t kTopRecords = 4096;

// child reference: interior -> byte offset of the child's record in `nodes` (< 2^31);
//                  leaf -> kLeafBit | (count-1) << 27 | triStart
__host__ __device__ inline uint32_t MakeLeafRef(uint32_t triStart, uint32_t count) {
    return kLeafBit | ((count - 1u) << 27) | triStart;
}

struct alignas(16) NodeRec {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t leftRef, rightRef;
    uint32_t pad[2];
};
static_assert(sizeof(NodeRec) == 64, "node record must be 64 bytes");

struct alignas(16) TriRec {
    float v0[3]; float e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; int32_t material; uint32_t pad[2];
};
static_assert(sizeof(TriRec) == 48, "triangle record must be 48 bytes");

struct alignas(16) MaterialRec {   // Fragment.glsl:15-23
    float colour[3];
    float specular[3];
    float roughness, metallic, transparency;
    float airOverIndex;          // 1.0003 / refractionIndex: what Fragment.glsl:330 divides per hit, divided once at upload
    int32_t isLight;
    float indexOverAir;          // refractionIndex / 1.0003
};
static_assert(sizeof(MaterialRec) == 48, "material record must be 48 bytes");

// Everything the tracer reads besides the big arrays; passed by value as a kernel argument
// (lands in SGPRs / the kernarg segment, i.e. the scalar cache, like GL uniforms).
struct SceneParams {
    const NodeRec* nodes;
    const TriRec* tris;
    const uint32_t* noise;
    const uint32_t* sky;
    const MaterialRec* materials;   // kMaxMaterials entries
    uint32_t nNode;
    uint32_t nInterior;          // records in `nodes`
    uint32_t rootRef;            // kNoRef for an empty scene
    float rootMin[3], rootMax[3];
    int32_t noiseW, noiseH, skySize;
    uint32_t bounceLimit;
    float camPos[3], camFacing[3];
    float vpW, vpD;
    float skyCos, skySin;        // cos/sin of the constant sky rotation angle, Fragment.glsl:406,411
    float skyXScale;             // skyCos + (1 - skyCos): what the rotation multiplies the x component by
    uint32_t width, height, tilesX, tilesY;
    uint32_t shardRank, shardWorld, shardTiles;   // this context renders global tile j*world+rank
};

// Per-frame uniforms, precomputed on the host where that is exact integer / single-op work.
struct alignas(16) FrameParams {
    uint32_t hashX, hashY;       // frame-constant parts of the per-bounce seed, Fragment.glsl:392-393
    float weight;                // 1.0 / float(u_FrameIndex + 1), Fragment.glsl:487
    uint32_t frameIndex;
    float jitterX, jitterY;      // texture(u_RgbNoise, u_RandSeed.yz).xy: the pixel jitter of Fragment.glsl:461,
                                 // filled in on the device (prepare_frames_kernel) from the two seeds below
    float seedY, seedZ;          // u_RandSeed.yz
};
static_assert(sizeof(FrameParams) == 32, "frame record must be 32 bytes");

enum CounterSlot { kRays = 0, kPops, kBoxes, kTris, kNoise, kSky, kPaths, kHits,
                   kRoundsNew, kRoundsInterior, kRoundsLeaf, kRoundsShade,   // scheduler rounds run per state (per wave)
                   kLanesNew, kLanesInterior, kLanesLeaf, kLanesShade,       // lanes that took part in those rounds
                   kCyclesNew, kCyclesInterior, kCyclesLeaf, kCyclesShade,   // wave time spent in those rounds, 64-cycle units
                   kCounterSlots };

}  // namespace rtd
