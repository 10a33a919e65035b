// Host side of the C ABI declared in include/rt_mi355x.h: owns the HIP resources of one
// renderer, validates and re-lays-out the caller's flat scene arrays, and launches the kernels.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_mi355x.h"

namespace rtd {
void LaunchPrepareFrames(hipStream_t st, const SceneParams& sp, FrameParams* frames, uint32_t nFrames, const FrameParams* single, uint32_t* queueHeads);
void LaunchPrepareRays(hipStream_t st, const SceneParams& sp, FrameParams* frames, uint32_t nFrames, uint32_t rayCount);
void LaunchTrace(bool count, bool ids, bool multi, hipStream_t st, uint32_t nWaves, const SceneParams& sp, const FrameParams* frames, uint32_t nFrames,
                 void* accum, uint32_t* tally, uint32_t tallyStride, int32_t* idsOut, void* spillArea, uint32_t* queueHeads, uint32_t nGroups,
                 const uint32_t* order, uint32_t* tileCost);
void LaunchOrderKeys(hipStream_t st, uint32_t* tileCost, uint32_t* keys, uint32_t* vals, uint32_t nTiles);
uint32_t SpillEntriesPerWave();
uint32_t ResidentWaves();
void LaunchReduceTally(hipStream_t st, uint32_t* tally, uint32_t nTiles, unsigned long long* out, int slots);
void LaunchUntile(hipStream_t st, const void* tiled, void* linear, uint32_t width, uint32_t height, uint32_t tilesX, uint32_t world,
                  uint32_t shardTiles, bool toTiled);
void LaunchRetile(hipStream_t st, const void* gathered, void* tiled, uint32_t nTiles, uint32_t world, uint32_t shardTiles);
void LaunchProbeTriangles(hipStream_t st, const void* tris, const float* o3, const float* d3, uint32_t n, uint32_t* hitT);
int RunPostprocess(hipStream_t st, const void* accumImage, uint32_t tilesX, uint32_t width, uint32_t height, int bloomOn, void** scratch,
                   float* rgbaOutHost, uint8_t* rgb8OutHost, std::string& err);
int ReadDisplay(hipStream_t st, void** scratch, uint32_t width, uint32_t height, float* rgbaOutHost, uint8_t* rgb8OutHost, std::string& err);
void FreePostprocess(void** scratch);
int ResetPostprocess(hipStream_t st, void** scratch);
int PrepareStaticScratch();
void SetGlobalError(const char* fmt, ...);
}  // namespace rtd

using namespace rtd;

static thread_local std::string g_globalError;

namespace rtd {
void SetGlobalError(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_globalError = buf;
}
}  // namespace rtd

struct rt_context {
    int device = 0;
    uint32_t width = 0, height = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;
    SceneParams sp{};
    // device arrays
    NodeRec* dNodes = nullptr; TriRec* dTris = nullptr; uint32_t* dNoise = nullptr; uint32_t* dSky = nullptr;
    MaterialRec* dMaterials = nullptr;
    void* dAccum = nullptr;          // tile-major shard, shardTiles*64 float4
    void* dLinear = nullptr;         // row-major staging image, width*height float4
    FrameParams* dFrames = nullptr; uint32_t framesCap = 0;   // per-frame records of a multi-frame launch
    // pinned staging for the records of multi-frame launches: a slot is reused once the copy out of it has run
    struct FrameStage { FrameParams* host = nullptr; uint32_t cap = 0; hipEvent_t copied = nullptr; bool pending = false; };
    FrameStage stage[4]; uint32_t stageNext = 0;
    uint32_t* dTally = nullptr;      // [kCounterSlots][maxWaves]
    void* dSpill = nullptr;          // stack spill area, maxWaves * SpillEntriesPerWave() * 8 bytes
    uint32_t* dQueue = nullptr;      // 64 words; the first 8 are the per-band tile counters
    // tile hand-out order: rays traced per tile since the last re-ordering, and the order derived from them
    uint32_t* dTileCost = nullptr;   // [shardTiles]
    uint32_t* dOrderKeys = nullptr;  // [2][shardTiles] sort keys in / out
    uint32_t* dOrderVals = nullptr;  // [2][shardTiles] tile ids in / out; the second half is the order
    void* dSortTemp = nullptr; size_t sortTempBytes = 0;
    bool orderValid = false;
    uint64_t launchesSinceAlloc = 0;
    uint32_t launchesSinceReduce = 0; // the 32-bit per-wave tallies are folded into the 64-bit totals every few launches
    uint64_t tileFramesSinceReduce = 0; // frames x tiles per wave traced since the last fold
    uint64_t shardPixels = 0;        // pixels of this shard that lie inside the image
    uint32_t maxWaves = 0;           // persistent grid size limit (resident waves of the device)
    unsigned long long* dTotals = nullptr;   // [kCounterSlots]
    int32_t* dIds = nullptr;
    void* postScratch = nullptr;
    bool haveNoise = false, haveSky = false, haveScene = false;
    bool countAll = false;
    uint64_t hostPaths = 0;          // pixel samples rendered while full counting was off
    uint32_t rayCount = 1;           // RAY_COUNT, Fragment.glsl:5: rays traced and averaged per pixel and frame
    // kernel timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pendingEvents, freeEvents;
    double kernelMs = 0.0; uint64_t kernelLaunches = 0;
    std::string error;
};

namespace {

int Fail(rt_context* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->error = buf; else g_globalError = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do { hipError_t e_ = (expr);                                                                 \
         if (e_ != hipSuccess) return Fail((c), RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

inline float Flush(float v) { return std::fabs(v) < 1.17549435e-38f ? std::copysign(0.0f, v) : v; }   // FTZ/DAZ, as the kernels run

uint32_t ShardTiles(const rt_context* c) {
    uint32_t n = c->sp.tilesX * c->sp.tilesY;
    return (n + c->sp.shardWorld - 1) / c->sp.shardWorld;
}

int AllocShard(rt_context* c) {
    void** old[] = {&c->dAccum, (void**)&c->dIds, (void**)&c->dTileCost, (void**)&c->dOrderKeys, (void**)&c->dOrderVals, &c->dSortTemp};
    for (void** p : old) if (*p) { (void)hipFree(*p); *p = nullptr; }
    c->orderValid = false; c->launchesSinceAlloc = 0; c->sortTempBytes = 0;
    c->sp.shardTiles = ShardTiles(c);
    const uint32_t n = c->sp.shardTiles;
    {   // pixels of this shard: tiles j*world+rank, clipped to the image
        uint64_t px = 0;
        const uint32_t nTiles = c->sp.tilesX * c->sp.tilesY;
        for (uint32_t t = c->sp.shardRank; t < nTiles; t += c->sp.shardWorld) {
            uint32_t tx = t % c->sp.tilesX, ty = t / c->sp.tilesX;
            uint32_t w = std::min<uint32_t>(kTile, c->width - tx * kTile), h = std::min<uint32_t>(kTile, c->height - ty * kTile);
            px += (uint64_t)w * h;
        }
        c->shardPixels = px;
    }
    size_t bytes = (size_t)n * 64 * sizeof(float) * 4;
    HIP_TRY(c, hipMalloc(&c->dAccum, bytes));
    HIP_TRY(c, hipMemsetAsync(c->dAccum, 0, bytes, c->stream));
    HIP_TRY(c, hipMalloc((void**)&c->dTileCost, sizeof(uint32_t) * n));
    HIP_TRY(c, hipMemsetAsync(c->dTileCost, 0, sizeof(uint32_t) * n, c->stream));
    HIP_TRY(c, hipMalloc((void**)&c->dOrderKeys, sizeof(uint32_t) * 2 * n));
    HIP_TRY(c, hipMalloc((void**)&c->dOrderVals, sizeof(uint32_t) * 2 * n));
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, c->sortTempBytes, c->dOrderKeys, c->dOrderKeys + n, c->dOrderVals, c->dOrderVals + n,
                                                  (int)n, 0, 29, c->stream));
    HIP_TRY(c, hipMalloc(&c->dSortTemp, std::max<size_t>(c->sortTempBytes, 16)));
    return RT_OK;
}

// Rebuilds the tile hand-out order from the rays traced per tile since the last call: within each eighth of the
// shard, most expensive tiles first, so that the last tiles a launch hands out are cheap ones and the waves
// finish together. Everything is queued on the context's stream; the host does not wait.
int UpdateTileOrder(rt_context* c) {
    const uint32_t n = c->sp.shardTiles;
    LaunchOrderKeys(c->stream, c->dTileCost, c->dOrderKeys, c->dOrderVals, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->dSortTemp, c->sortTempBytes, c->dOrderKeys, c->dOrderKeys + n, c->dOrderVals, c->dOrderVals + n,
                                                  (int)n, 0, 29, c->stream));
    c->orderValid = true;
    return RT_OK;
}

int ResolveEvents(rt_context* c) {
    for (auto& pr : c->pendingEvents) {
        HIP_TRY(c, hipEventSynchronize(pr.second));
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, pr.first, pr.second));
        c->kernelMs += ms; c->kernelLaunches++;
        c->freeEvents.push_back(pr);
    }
    c->pendingEvents.clear();
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_global_error(void) { return g_globalError.c_str(); }
const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->error.c_str() : g_globalError.c_str(); }

int rt_create(int device, uint32_t width, uint32_t height, rt_context** out) {
    if (!out) return Fail(nullptr, RT_ERR_INVALID, "rt_create: out is NULL");
    *out = nullptr;
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return Fail(nullptr, RT_ERR_INVALID, "rt_create: bad image size %ux%u", width, height);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return Fail(nullptr, RT_ERR_DEVICE, "rt_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return Fail(nullptr, RT_ERR_INVALID, "rt_create: device %d out of range (have %d)", device, count);
    HIP_TRY(nullptr, hipSetDevice(device));
    if (PrepareStaticScratch() != RT_OK) return Fail(nullptr, RT_ERR_DEVICE, "rt_create: cannot locate the library's device-side working array");
    rt_context* c = new rt_context();
    c->device = device; c->width = width; c->height = height;
    SceneParams& sp = c->sp;
    sp.width = width; sp.height = height;
    sp.tilesX = (width + kTile - 1) / kTile; sp.tilesY = (height + kTile - 1) / kTile;
    sp.shardRank = 0; sp.shardWorld = 1;
    sp.rootRef = kNoRef; sp.nNode = 0; sp.nInterior = 0;
    sp.bounceLimit = 4;                                   // src/Scene.cpp:19
    sp.camPos[0] = sp.camPos[1] = sp.camPos[2] = 0.0f;    // Include/Camera.h:7-11
    sp.camFacing[0] = 1.0f; sp.camFacing[1] = sp.camFacing[2] = 0.0f;
    sp.vpD = 1.0f;
    sp.vpW = (float)(2.0 * 1.0 * std::tan((float)(120 * 0.01745329251994329576923690768489) / 2.0));   // src/Camera.cpp:14,19
    const float a = (float)(-3.1415 / 2);                 // Fragment.glsl:406
    sp.skyCos = std::cos(a); sp.skySin = std::sin(a);
    { volatile float rest = 1.0f - sp.skyCos; volatile float sum = sp.skyCos + rest; sp.skyXScale = sum; }   // two fp32 roundings, as on the device
    hipError_t se = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return Fail(nullptr, RT_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(se)); }
    c->stream = c->ownStream;
    int rc = AllocShard(c);
    if (rc == RT_OK) {
        hipError_t me = hipMalloc((void**)&c->dMaterials, sizeof(MaterialRec) * kMaxMaterials);
        if (me == hipSuccess) me = hipMemsetAsync(c->dMaterials, 0, sizeof(MaterialRec) * kMaxMaterials, c->stream);
        if (me == hipSuccess) me = hipMalloc((void**)&c->dTotals, sizeof(unsigned long long) * kCounterSlots);
        if (me == hipSuccess) me = hipMemsetAsync(c->dTotals, 0, sizeof(unsigned long long) * kCounterSlots, c->stream);
        if (me == hipSuccess) me = hipMalloc(&c->dLinear, (size_t)width * height * 16);
        c->maxWaves = ResidentWaves();
        if (me == hipSuccess && c->maxWaves == 0) me = hipErrorUnknown;
        const size_t tb = (size_t)kCounterSlots * c->maxWaves * sizeof(uint32_t);
        if (me == hipSuccess) me = hipMalloc((void**)&c->dTally, tb);
        if (me == hipSuccess) me = hipMemsetAsync(c->dTally, 0, tb, c->stream);
        if (me == hipSuccess) me = hipMalloc(&c->dSpill, (size_t)c->maxWaves * SpillEntriesPerWave() * 8);
        if (me == hipSuccess) me = hipMalloc((void**)&c->dQueue, 256);
        if (me == hipSuccess) me = hipMalloc((void**)&c->dFrames, sizeof(FrameParams) * 64);
        if (me == hipSuccess) c->framesCap = 64;
        if (me != hipSuccess) rc = Fail(c, RT_ERR_DEVICE, "rt_create: %s", hipGetErrorString(me));
    }
    if (rc != RT_OK) { g_globalError = c->error; rt_destroy(c); return rc; }
    sp.materials = c->dMaterials;
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& pr : c->pendingEvents) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : c->freeEvents) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    FreePostprocess(&c->postScratch);
    for (auto& sg : c->stage) { if (sg.host) (void)hipHostFree(sg.host); if (sg.copied) (void)hipEventDestroy(sg.copied); }
    void* ptrs[] = {c->dNodes, c->dTris, c->dNoise, c->dSky, c->dMaterials, c->dAccum, c->dLinear, c->dFrames, c->dTally, c->dTotals, c->dIds, c->dSpill, c->dQueue,
                    c->dTileCost, c->dOrderKeys, c->dOrderVals, c->dSortTemp};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

namespace {
// One triangle of the reference's buffer (three corners, xyz0 each) as the 48-byte device record: first corner and the
// two edges, subtracted here once the way the shader subtracts them per test (Fragment.glsl:212-213).
TriRec MakeTriRec(const float* p, int32_t material) {
    TriRec t;
    float v0[3] = {Flush(p[0]), Flush(p[1]), Flush(p[2])};
    float e1[3], e2[3];
    for (int k = 0; k < 3; ++k) { e1[k] = Flush(Flush(p[4 + k]) - v0[k]); e2[k] = Flush(Flush(p[8 + k]) - v0[k]); }
    t.v0[0] = v0[0]; t.v0[1] = v0[1]; t.v0[2] = v0[2];
    t.e1x = e1[0]; t.e1y = e1[1]; t.e1z = e1[2];
    t.e2x = e2[0]; t.e2y = e2[1]; t.e2z = e2[2];
    t.material = material; t.pad[0] = t.pad[1] = 0;
    return t;
}
}  // namespace

int rt_scene_upload(rt_context* c, const float* tri12, const int32_t* matIdx, uint32_t nTri, const float* node8, uint32_t nNode) try {
    if (!c) return RT_ERR_INVALID;
    if ((nTri && (!tri12 || !matIdx)) || (nNode && !node8)) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: NULL array");
    if (nTri >= (1u << 24)) return Fail(c, RT_ERR_UNSUPPORTED, "rt_scene_upload: %u triangles; indices travel as floats and are exact only below 2^24", nTri);
    if (nNode >= (1u << 24)) return Fail(c, RT_ERR_UNSUPPORTED, "rt_scene_upload: %u nodes; indices travel as floats and are exact only below 2^24", nNode);
    HIP_TRY(c, hipSetDevice(c->device));

    // ---- validate the pre-order tree and measure its depth (iteratively; no recursion) ----
    struct Info { int count; int ref; };
    auto info = [&](uint32_t i) { Info r; r.count = (int)node8[(size_t)i * 8 + 6]; r.ref = (int)node8[(size_t)i * 8 + 7]; return r; };
    std::vector<NodeRec> nodes;
    int maxInteriorDepth = 0;
    if (nNode) {
        std::vector<std::pair<uint32_t, int>> todo;   // (node, depth), root depth 1
        std::vector<uint8_t> seen(nNode, 0);
        todo.emplace_back(0u, 1);
        uint64_t visited = 0;
        while (!todo.empty()) {
            auto [i, depth] = todo.back(); todo.pop_back();
            if (i >= nNode) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: child index %u out of range", i);
            if (seen[i]) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: node %u reachable twice (not a tree)", i);
            seen[i] = 1; ++visited;
            Info in = info(i);
            if (in.count > 0) {
                if (in.count > 16) return Fail(c, RT_ERR_UNSUPPORTED, "rt_scene_upload: leaf %u holds %d triangles (limit 16)", i, in.count);
                if (in.ref < 0 || (uint64_t)in.ref + (uint64_t)in.count > nTri) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: leaf %u triangle range out of bounds", i);
            } else {
                if (in.count < 0) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: node %u has a negative triangle count", i);
                // An interior node at depth d (root = 1) defers its far child into stack entry d when all its ancestors
                // were entered through both children; entries 0 .. kMaxStack-1 exist (the shader's stack[64] holds the
                // same: its interior nodes may be 63 deep).
                if (depth >= kMaxStack) return Fail(c, RT_ERR_UNSUPPORTED, "rt_scene_upload: interior nodes more than %d levels deep", kMaxStack - 1);
                if (in.ref <= (int)i + 1 || (uint32_t)in.ref >= nNode || i + 1 >= nNode) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: node %u has a bad right child %d", i, in.ref);
                if (depth > maxInteriorDepth) maxInteriorDepth = depth;
                todo.emplace_back((uint32_t)in.ref, depth + 1);
                todo.emplace_back(i + 1, depth + 1);
            }
        }
        (void)visited;
        // ---- re-lay-out: one 64-byte record per interior node, interior nodes numbered consecutively in
        // the reference's pre-order (a node's left child, when interior, is the next record); an interior
        // child is referenced by its record's byte offset ----
        // The first kTopRecords records are the top of the tree in breadth-first order (the nodes every ray passes),
        // so that they are contiguous; the others follow in pre-order. (Choosing them by box area instead, largest
        // first, covers the same share of the visits: measured, no gain.)
        constexpr uint32_t kUnset = 0xffffffffu;
        std::vector<uint32_t> recordOf(nNode, kUnset);
        uint32_t nInterior = 0;
        if (info(0).count == 0) {
            std::vector<uint32_t> level{0u};
            for (size_t head = 0; head < level.size() && nInterior < kTopRecords; ++head) {
                const uint32_t i = level[head];
                recordOf[i] = nInterior++;
                const uint32_t kids[2] = {i + 1, (uint32_t)info(i).ref};
                for (uint32_t k : kids) if (info(k).count == 0) level.push_back(k);
            }
        }
        for (uint32_t i = 0; i < nNode; ++i)
            if (seen[i] && info(i).count == 0 && recordOf[i] == kUnset) recordOf[i] = nInterior++;
        nodes.assign(nInterior, NodeRec{});
        auto childRef = [&](uint32_t child) {
            Info in = info(child);
            return in.count > 0 ? MakeLeafRef((uint32_t)in.ref, (uint32_t)in.count) : recordOf[child] * (uint32_t)sizeof(NodeRec);
        };
        auto copyBox = [&](uint32_t child, float* mn, float* mx) {
            const float* p = node8 + (size_t)child * 8;
            for (int k = 0; k < 3; ++k) { mx[k] = p[k]; mn[k] = p[3 + k]; }
        };
        for (uint32_t i = 0; i < nNode; ++i) {
            Info in = info(i);
            if (in.count > 0 || !seen[i]) continue;
            NodeRec& r = nodes[recordOf[i]];
            copyBox(i + 1, r.lmin, r.lmax);
            copyBox((uint32_t)in.ref, r.rmin, r.rmax);
            r.leftRef = childRef(i + 1); r.rightRef = childRef((uint32_t)in.ref);
        }
    }
    for (uint32_t i = 0; i < nTri; ++i)
        if (matIdx[i] < 0 || matIdx[i] >= kMaxMaterials) return Fail(c, RT_ERR_INVALID, "rt_scene_upload: triangle %u uses material slot %d (valid: 0..15)", i, matIdx[i]);

    std::vector<TriRec> tris(nTri);
    for (uint32_t i = 0; i < nTri; ++i) tris[i] = MakeTriRec(tri12 + (size_t)i * 12, matIdx[i]);

    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // from here until the new arrays are in place the context has no scene (a failed allocation must not leave
    // the old, freed arrays reachable)
    c->haveScene = false; c->sp.nodes = nullptr; c->sp.tris = nullptr; c->sp.rootRef = kNoRef; c->sp.nNode = 0; c->sp.nInterior = 0;
    if (c->dNodes) { (void)hipFree(c->dNodes); c->dNodes = nullptr; }
    if (c->dTris) { (void)hipFree(c->dTris); c->dTris = nullptr; }
    if (!nodes.empty()) {
        HIP_TRY(c, hipMalloc((void**)&c->dNodes, sizeof(NodeRec) * nodes.size()));
        HIP_TRY(c, hipMemcpy(c->dNodes, nodes.data(), sizeof(NodeRec) * nodes.size(), hipMemcpyHostToDevice));
    }
    if (nTri) {
        HIP_TRY(c, hipMalloc((void**)&c->dTris, sizeof(TriRec) * nTri));
        HIP_TRY(c, hipMemcpy(c->dTris, tris.data(), sizeof(TriRec) * nTri, hipMemcpyHostToDevice));
    }
    SceneParams& sp = c->sp;
    sp.nodes = c->dNodes; sp.tris = c->dTris; sp.nNode = nNode; sp.nInterior = (uint32_t)nodes.size();
    if (nNode == 0) sp.rootRef = kNoRef;
    else {
        Info in = info(0);
        sp.rootRef = in.count > 0 ? MakeLeafRef((uint32_t)in.ref, (uint32_t)in.count) : 0u;
        for (int k = 0; k < 3; ++k) { sp.rootMax[k] = node8[k]; sp.rootMin[k] = node8[3 + k]; }
    }
    c->haveScene = true;
    return RT_OK;
} catch (const std::exception& e) {
    return Fail(c, RT_ERR_DEVICE, "out of host memory or internal error: %s", e.what());
}

namespace {
// Tile-frames (frames x tiles per wave) after which the 32-bit per-wave tally columns are folded into the 64-bit totals.
constexpr uint64_t kFoldTileFrames = 4096;
constexpr float kAirIndex = 1.0003f;     // Fragment.glsl:329

// a / b as the kernels compute it: correctly rounded fp32 with denormal operands read as zero and a denormal
// result flushed to zero (the host's division rounds the same way; only the flushing has to be added).
float DeviceQuotient(float a, float b) {
    auto flush = [](float v) { return std::fabs(v) < FLT_MIN ? std::copysign(0.0f, v) : v; };
    volatile float q = flush(a) / flush(b);
    return flush(q);
}
}  // namespace

int rt_set_materials(rt_context* c, const rt_material* m, uint32_t count) {
    if (!c) return RT_ERR_INVALID;
    if (count > (uint32_t)kMaxMaterials) return Fail(c, RT_ERR_UNSUPPORTED, "rt_set_materials: %u materials (limit 16)", count);
    if (count && !m) return Fail(c, RT_ERR_INVALID, "rt_set_materials: NULL table");
    HIP_TRY(c, hipSetDevice(c->device));
    MaterialRec recs[kMaxMaterials];
    std::memset(recs, 0, sizeof recs);
    for (uint32_t i = 0; i < count; ++i) {
        for (int k = 0; k < 3; ++k) { recs[i].colour[k] = m[i].colour[k]; recs[i].specular[k] = m[i].specularColour[k]; }
        recs[i].roughness = m[i].roughness; recs[i].metallic = m[i].metallic; recs[i].transparency = m[i].transparency;
        recs[i].airOverIndex = DeviceQuotient(kAirIndex, m[i].refractionIndex);
        recs[i].indexOverAir = DeviceQuotient(m[i].refractionIndex, kAirIndex);
        recs[i].isLight = m[i].isLight ? 1 : 0;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->dMaterials, recs, sizeof recs, hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_set_camera(rt_context* c, const rt_camera* cam) {
    if (!c || !cam) return c ? Fail(c, RT_ERR_INVALID, "rt_set_camera: NULL camera") : RT_ERR_INVALID;
    for (int k = 0; k < 3; ++k) { c->sp.camPos[k] = cam->position[k]; c->sp.camFacing[k] = cam->facing[k]; }
    c->sp.vpW = cam->viewportWidth; c->sp.vpD = cam->viewportDistance;
    return RT_OK;
}

int rt_set_bounce_limit(rt_context* c, uint32_t bl) {
    if (!c) return RT_ERR_INVALID;
    c->sp.bounceLimit = bl;
    return RT_OK;
}

int rt_set_ray_count(rt_context* c, uint32_t rays) {
    if (!c) return RT_ERR_INVALID;
    if (rays < 1u || rays > RT_MAX_RAY_COUNT) return Fail(c, RT_ERR_INVALID, "rt_set_ray_count: %u is outside 1..%u", rays, (unsigned)RT_MAX_RAY_COUNT);
    c->rayCount = rays;
    return RT_OK;
}

int rt_set_noise(rt_context* c, const uint8_t* rgb, uint32_t w, uint32_t h) try {
    if (!c) return RT_ERR_INVALID;
    if (!rgb || w == 0 || h == 0 || w > 16384 || h > 16384) return Fail(c, RT_ERR_INVALID, "rt_set_noise: bad texture");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<uint32_t> px((size_t)w * h);
    for (size_t i = 0; i < px.size(); ++i) px[i] = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->haveNoise = false; c->sp.noise = nullptr;
    if (c->dNoise) { (void)hipFree(c->dNoise); c->dNoise = nullptr; }
    HIP_TRY(c, hipMalloc((void**)&c->dNoise, px.size() * 4));
    HIP_TRY(c, hipMemcpy(c->dNoise, px.data(), px.size() * 4, hipMemcpyHostToDevice));
    c->sp.noise = c->dNoise; c->sp.noiseW = (int32_t)w; c->sp.noiseH = (int32_t)h;
    c->haveNoise = true;
    return RT_OK;
} catch (const std::exception& e) {
    return Fail(c, RT_ERR_DEVICE, "out of host memory or internal error: %s", e.what());
}

int rt_set_skybox(rt_context* c, const uint8_t* const faces[6], uint32_t size) {
    if (!c) return RT_ERR_INVALID;
    if (!faces || size == 0 || size > 8192) return Fail(c, RT_ERR_INVALID, "rt_set_skybox: bad cube map");
    for (int f = 0; f < 6; ++f) if (!faces[f]) return Fail(c, RT_ERR_INVALID, "rt_set_skybox: face %d is NULL", f);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)size * size;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->haveSky = false; c->sp.sky = nullptr;
    if (c->dSky) { (void)hipFree(c->dSky); c->dSky = nullptr; }
    HIP_TRY(c, hipMalloc((void**)&c->dSky, n * 6 * 4));
    for (int f = 0; f < 6; ++f) HIP_TRY(c, hipMemcpy(c->dSky + n * f, faces[f], n * 4, hipMemcpyHostToDevice));
    c->sp.sky = c->dSky; c->sp.skySize = (int32_t)size;
    c->haveSky = true;
    return RT_OK;
}

int rt_accum_clear(rt_context* c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->dAccum) return Fail(c, RT_ERR_STATE, "rt_accum_clear: no accumulation buffer (rt_set_shard failed earlier)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->dAccum, 0, (size_t)c->sp.shardTiles * 64 * 16, c->stream));
    return RT_OK;
}

int rt_accum_write(rt_context* c, const float* rgba) {
    if (!c || !rgba) return c ? Fail(c, RT_ERR_INVALID, "rt_accum_write: NULL image") : RT_ERR_INVALID;
    if (c->sp.shardWorld != 1) return Fail(c, RT_ERR_STATE, "rt_accum_write: only on an unsharded context");
    if (!c->dAccum) return Fail(c, RT_ERR_STATE, "rt_accum_write: no accumulation buffer (rt_set_shard failed earlier)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->dLinear, rgba, (size_t)c->width * c->height * 16, hipMemcpyHostToDevice, c->stream));
    LaunchUntile(c->stream, c->dAccum, c->dLinear, c->width, c->height, c->sp.tilesX, 1, c->sp.shardTiles, true);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_accum_read(rt_context* c, float* rgba) {
    if (!c || !rgba) return c ? Fail(c, RT_ERR_INVALID, "rt_accum_read: NULL image") : RT_ERR_INVALID;
    if (c->sp.shardWorld != 1) return Fail(c, RT_ERR_STATE, "rt_accum_read: sharded context; gather the shards and call rt_accum_assemble on an unsharded one");
    if (!c->dAccum) return Fail(c, RT_ERR_STATE, "rt_accum_read: no accumulation buffer (rt_set_shard failed earlier)");
    HIP_TRY(c, hipSetDevice(c->device));
    LaunchUntile(c->stream, c->dAccum, c->dLinear, c->width, c->height, c->sp.tilesX, 1, c->sp.shardTiles, false);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(rgba, c->dLinear, (size_t)c->width * c->height * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// Queues one launch of the persistent tracer over this context's shard.
static int LaunchShard(rt_context* c, bool count, bool ids, uint32_t nFrames) {
    // a wave takes one 8x8 tile from its queue at a time (measured best: 8.38 ms against 8.52 with two, 9.16 with four)
    const uint32_t nGroups = c->sp.shardTiles;
    const uint32_t nWaves = std::max<uint32_t>(1u, std::min<uint32_t>(nGroups, c->maxWaves));
    // (the tile queue heads were cleared by prepare_frames_kernel, which UploadFrames queued just before)
    // The order is rebuilt after launches 1, 2, 4, 8 and then after every 8th: the cost of a tile changes slowly from
    // frame to frame. Only those launches tally rays per tile (one atomic per path, about 1.5 % of a launch).
    // A launch that alone comes near the fold threshold starts from empty columns; with the full counters on, one
    // that would exceed it eight times over is refused rather than allowed to wrap.
    // (a frame of R rays per pixel fills the counters like R frames)
    const uint64_t launchTileFrames = (uint64_t)nFrames * (ids ? 1u : c->rayCount) * ((nGroups + nWaves - 1u) / nWaves);
    if (!ids && count && launchTileFrames > 8u * kFoldTileFrames)
        return Fail(c, RT_ERR_UNSUPPORTED, "rt_render: %u frames in one launch are too many for the 32-bit per-wave counters of counting mode; split the launch", nFrames);
    if (!ids && c->tileFramesSinceReduce != 0 && c->tileFramesSinceReduce + launchTileFrames >= kFoldTileFrames) {
        LaunchReduceTally(c->stream, c->dTally, c->maxWaves, c->dTotals, kCounterSlots);
        HIP_TRY(c, hipGetLastError());
        c->launchesSinceReduce = 0;
        c->tileFramesSinceReduce = 0;
    }
    const uint64_t k = ids ? 0 : c->launchesSinceAlloc + 1;
    const bool reorderAfter = !ids && ((k & (k - 1)) == 0 || (k & 7u) == 0);
    const uint32_t* order = (c->orderValid && !ids) ? c->dOrderVals + c->sp.shardTiles : nullptr;
    LaunchTrace(count, ids, !ids && c->rayCount > 1u, c->stream, nWaves, c->sp, c->dFrames, nFrames, c->dAccum, c->dTally, c->maxWaves, c->dIds, c->dSpill, c->dQueue, nGroups,
                order, reorderAfter ? c->dTileCost : nullptr);
    HIP_TRY(c, hipGetLastError());
    if (!ids) {
        c->launchesSinceAlloc = k;
        if (reorderAfter) { int rc = UpdateTileOrder(c); if (rc != RT_OK) return rc; }
        // a wave's tally column is 32 bits wide: fold the columns into the 64-bit totals before they can wrap
        // (stream-ordered, no host wait; rt_get_counters folds what is left)
        // What fills a column is paths per wave: frames x tiles per wave x 64, times the events per path (a few rays;
        // with the full counters on, some hundred box tests). Folding every 16 launches or every kFoldTileFrames
        // tile-frames per wave, whichever comes first, leaves three orders of magnitude of headroom at 2000 box tests
        // per path even if one wave takes eight times its share of the tiles.
        c->launchesSinceReduce += 1u;
        c->tileFramesSinceReduce += launchTileFrames;
        if (c->launchesSinceReduce >= 16u || c->tileFramesSinceReduce >= kFoldTileFrames) {
            LaunchReduceTally(c->stream, c->dTally, c->maxWaves, c->dTotals, kCounterSlots);
            HIP_TRY(c, hipGetLastError());
            c->launchesSinceReduce = 0;
            c->tileFramesSinceReduce = 0;
        }
    }
    return RT_OK;
}

static int CheckRenderable(rt_context* c, const char* who) {
    if (!c->dAccum || !c->dTileCost || !c->dOrderVals || !c->dSortTemp) return Fail(c, RT_ERR_STATE, "%s: no accumulation buffers (rt_set_shard failed earlier)", who);
    if (!c->haveNoise) return Fail(c, RT_ERR_STATE, "%s: no noise texture (rt_set_noise)", who);
    if (!c->haveSky) return Fail(c, RT_ERR_STATE, "%s: no skybox (rt_set_skybox)", who);
    if (!c->haveScene) return Fail(c, RT_ERR_STATE, "%s: no scene (rt_scene_upload)", who);
    return RT_OK;
}

// The per-frame record of frame `index` with seed triple `s` (Scene::Tick's two uniforms, src/Scene.cpp:30-31); the
// pixel jitter is filled in on the device.
static FrameParams MakeFrame(const rt_context* c, uint32_t index, const float* s) {
    const float BLf = (float)c->sp.bounceLimit;
    auto hash = [](uint32_t x) { x ^= x >> 17; x *= 0xed5ad4bbU; x ^= x >> 11; x *= 0xac4c1b51U; x ^= x >> 15; x *= 0x31848babU; x ^= x >> 14; return x; };
    auto hashf = [&](float f) { uint32_t u; std::memcpy(&u, &f, 4); return hash(u); };
    FrameParams f;
    f.frameIndex = index;
    f.seedY = s[1]; f.seedZ = s[2];
    // Fragment.glsl:392-393, the terms that do not change within a frame
    f.hashX = hash(0u) + hash(f.frameIndex) + hashf(Flush(Flush(s[0]) * BLf));
    f.hashY = hash(0u) + hashf(Flush(Flush(s[1]) * BLf));
    f.weight = 1.0f / (float)(f.frameIndex + 1u);     // Fragment.glsl:487
    f.jitterX = f.jitterY = 0.0f;
    return f;
}

// Hands the frame records of a launch to the device without waiting for it, and queues prepare_frames_kernel, which
// fills in their jitter and clears the tile queue heads. One frame: the record is an argument of that kernel.
// Several frames: the records go through a pinned staging slot and one asynchronous copy; the host only waits when
// all staging slots still have their copy queued, i.e. when it is four multi-frame launches ahead of the device.
// With more than one ray per pixel and frame (`rays`), one record holding the ray count and a table of rays x n pixel
// jitters (8 bytes each) follow the n frame records; prepare_rays_kernel fills both in.
static int UploadFrames(rt_context* c, uint32_t first, uint32_t n, const float* seeds3, uint32_t rays) {
    const uint64_t records = rays > 1u ? (uint64_t)n + 1u + ((uint64_t)n * rays * 8u + sizeof(FrameParams) - 1u) / sizeof(FrameParams) : n;
    // (the kernels index the jitter table with a 24-bit multiply of the frame number)
    if (records > 0x7fffffffull / sizeof(FrameParams) || (rays > 1u && n >= (1u << 24)))
        return Fail(c, RT_ERR_UNSUPPORTED, "rt_render: %u frames of %u rays are too many for one launch", n, rays);
    if (records > c->framesCap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->dFrames) (void)hipFree(c->dFrames);
        c->dFrames = nullptr; c->framesCap = 0;
        const uint32_t cap = std::max<uint32_t>((uint32_t)records, 64u);
        HIP_TRY(c, hipMalloc((void**)&c->dFrames, sizeof(FrameParams) * cap));
        c->framesCap = cap;
    }
    if (n == 1) {
        const FrameParams f = MakeFrame(c, first, seeds3);
        LaunchPrepareFrames(c->stream, c->sp, c->dFrames, 1, &f, c->dQueue);
        HIP_TRY(c, hipGetLastError());
        if (rays > 1u) { LaunchPrepareRays(c->stream, c->sp, c->dFrames, 1, rays); HIP_TRY(c, hipGetLastError()); }
        return RT_OK;
    }
    rt_context::FrameStage& sg = c->stage[c->stageNext++ % 4u];
    if (sg.pending) { HIP_TRY(c, hipEventSynchronize(sg.copied)); sg.pending = false; }
    if (!sg.copied) HIP_TRY(c, hipEventCreateWithFlags(&sg.copied, hipEventDisableTiming));
    if (n > sg.cap) {
        if (sg.host) (void)hipHostFree(sg.host);
        sg.host = nullptr; sg.cap = 0;
        uint32_t cap = 64; while (cap < n) cap *= 2u;
        HIP_TRY(c, hipHostMalloc((void**)&sg.host, sizeof(FrameParams) * cap, hipHostMallocDefault));
        sg.cap = cap;
    }
    for (uint32_t k = 0; k < n; ++k) sg.host[k] = MakeFrame(c, first + k, seeds3 + 3 * (size_t)k);
    HIP_TRY(c, hipMemcpyAsync(c->dFrames, sg.host, sizeof(FrameParams) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(sg.copied, c->stream));
    sg.pending = true;
    LaunchPrepareFrames(c->stream, c->sp, c->dFrames, n, nullptr, c->dQueue);
    HIP_TRY(c, hipGetLastError());
    if (rays > 1u) { LaunchPrepareRays(c->stream, c->sp, c->dFrames, n, rays); HIP_TRY(c, hipGetLastError()); }
    return RT_OK;
}

int rt_render(rt_context* c, uint32_t firstFrameIndex, uint32_t nFrames, const float* seeds3) {
    if (!c) return RT_ERR_INVALID;
    if (nFrames == 0) return RT_OK;
    if (!seeds3) return Fail(c, RT_ERR_INVALID, "rt_render: NULL seeds");
    int rc = CheckRenderable(c, "rt_render");
    if (rc != RT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = UploadFrames(c, firstFrameIndex, nFrames, seeds3, c->rayCount);
    if (rc != RT_OK) return rc;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!c->freeEvents.empty()) { ev = c->freeEvents.back(); c->freeEvents.pop_back(); }
    else { HIP_TRY(c, hipEventCreate(&ev.first)); HIP_TRY(c, hipEventCreate(&ev.second)); }
    HIP_TRY(c, hipEventRecord(ev.first, c->stream));
    rc = LaunchShard(c, c->countAll, false, nFrames);
    if (rc != RT_OK) return rc;
    HIP_TRY(c, hipEventRecord(ev.second, c->stream));
    c->pendingEvents.push_back(ev);
    if (!c->countAll && c->rayCount == 1u) c->hostPaths += c->shardPixels * nFrames;      // (with several rays per pixel the kernel counts its paths)
    if (c->pendingEvents.size() > 256) { rc = ResolveEvents(c); if (rc != RT_OK) return rc; }
    return RT_OK;
}

int rt_sync(rt_context* c) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_primary_ids(rt_context* c, const float* seed3, int32_t* out3) try {
    if (!c || !seed3 || !out3) return c ? Fail(c, RT_ERR_INVALID, "rt_primary_ids: NULL argument") : RT_ERR_INVALID;
    if (c->sp.shardWorld != 1) return Fail(c, RT_ERR_STATE, "rt_primary_ids: only on an unsharded context");
    int rc = CheckRenderable(c, "rt_primary_ids");
    if (rc != RT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = UploadFrames(c, 0, 1, seed3, 1u);
    if (rc != RT_OK) return rc;
    // what rt_render has tallied so far goes to the totals first: the id pass shares the tally columns
    LaunchReduceTally(c->stream, c->dTally, c->maxWaves, c->dTotals, kCounterSlots);
    HIP_TRY(c, hipGetLastError());
    c->launchesSinceReduce = 0;
    c->tileFramesSinceReduce = 0;
    const size_t n = (size_t)c->sp.shardTiles * 64;
    if (!c->dIds) HIP_TRY(c, hipMalloc((void**)&c->dIds, n * 3 * sizeof(int32_t)));
    HIP_TRY(c, hipMemsetAsync(c->dIds, 0xff, n * 3 * sizeof(int32_t), c->stream));
    rc = LaunchShard(c, true, true, 1);
    if (rc != RT_OK) return rc;
    std::vector<int32_t> tiled(n * 3);
    HIP_TRY(c, hipMemcpyAsync(tiled.data(), c->dIds, n * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t y = 0; y < c->height; ++y)
        for (uint32_t x = 0; x < c->width; ++x) {
            size_t slot = ((size_t)(y / kTile) * c->sp.tilesX + x / kTile) * 64 + (y % kTile) * kTile + x % kTile;
            std::memcpy(out3 + ((size_t)y * c->width + x) * 3, tiled.data() + slot * 3, 3 * sizeof(int32_t));
        }
    // the id pass ran the tallies too; drop its own so that the counters describe rt_render only
    HIP_TRY(c, hipMemsetAsync(c->dTally, 0, (size_t)kCounterSlots * c->maxWaves * sizeof(uint32_t), c->stream));
    return RT_OK;
} catch (const std::exception& e) {
    return Fail(c, RT_ERR_DEVICE, "out of host memory or internal error: %s", e.what());
}

int rt_probe_triangles(rt_context* c, uint32_t n, const float* tri12, const float* origins3, const float* directions3, uint8_t* hitOut,
                       float* tOut) try {
    if (!c) return RT_ERR_INVALID;
    if (n && (!tri12 || !origins3 || !directions3 || !hitOut || !tOut)) return Fail(c, RT_ERR_INVALID, "rt_probe_triangles: NULL argument");
    if (n > (1u << 24)) return Fail(c, RT_ERR_UNSUPPORTED, "rt_probe_triangles: %u pairs (limit %u)", n, 1u << 24);
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<TriRec> tris(n);
    for (uint32_t i = 0; i < n; ++i) tris[i] = MakeTriRec(tri12 + (size_t)i * 12, 0);
    std::vector<uint32_t> hitT(n);
    const size_t triBytes = sizeof(TriRec) * n, vecBytes = sizeof(float) * 3 * n, outBytes = sizeof(uint32_t) * n;
    char* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, triBytes + 2 * vecBytes + outBytes));
    float* dO = reinterpret_cast<float*>(d + triBytes);
    float* dD = reinterpret_cast<float*>(d + triBytes + vecBytes);
    uint32_t* dOut = reinterpret_cast<uint32_t*>(d + triBytes + 2 * vecBytes);
    hipError_t e = hipMemcpyAsync(d, tris.data(), triBytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dO, origins3, vecBytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dD, directions3, vecBytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { LaunchProbeTriangles(c->stream, d, dO, dD, n, dOut); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(hitT.data(), dOut, outBytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIP_TRY(c, e);
    for (uint32_t i = 0; i < n; ++i) {
        hitOut[i] = hitT[i] != 0xffffffffu;
        const uint32_t bits = hitOut[i] ? hitT[i] : 0u;
        std::memcpy(tOut + i, &bits, sizeof bits);
    }
    return RT_OK;
} catch (const std::exception& e) {
    return Fail(c, RT_ERR_DEVICE, "out of host memory or internal error: %s", e.what());
}

int rt_postprocess(rt_context* c, int bloom_on, float* rgba_out, uint8_t* rgb8_out) {
    if (!c) return RT_ERR_INVALID;
    if (c->sp.shardWorld != 1) return Fail(c, RT_ERR_STATE, "rt_postprocess: only on an unsharded context");
    if (!c->dAccum) return Fail(c, RT_ERR_STATE, "rt_postprocess: no accumulation buffer (rt_set_shard failed earlier)");
    HIP_TRY(c, hipSetDevice(c->device));
    std::string err;
    int rc;
    if (bloom_on) {
        // the chain reads the tile-major image as it is
        rc = RunPostprocess(c->stream, c->dAccum, c->sp.tilesX, c->width, c->height, 1, &c->postScratch, rgba_out, rgb8_out, err);
    } else {
        LaunchUntile(c->stream, c->dAccum, c->dLinear, c->width, c->height, c->sp.tilesX, 1, c->sp.shardTiles, false);
        HIP_TRY(c, hipGetLastError());
        rc = RunPostprocess(c->stream, c->dLinear, 0, c->width, c->height, 0, &c->postScratch, rgba_out, rgb8_out, err);
    }
    if (rc != RT_OK) return Fail(c, rc, "rt_postprocess: %s", err.c_str());
    return RT_OK;
}

int rt_display_read(rt_context* c, float* rgba_out, uint8_t* rgb8_out) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    std::string err;
    int rc = ReadDisplay(c->stream, &c->postScratch, c->width, c->height, rgba_out, rgb8_out, err);
    if (rc != RT_OK) return Fail(c, rc, "rt_display_read: %s", err.c_str());
    return RT_OK;
}

int rt_display_reset(rt_context* c) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    // the next rt_postprocess starts from a zeroed tmp image (queued on the stream; the buffers are kept)
    if (ResetPostprocess(c->stream, &c->postScratch) != RT_OK) return Fail(c, RT_ERR_DEVICE, "rt_display_reset: clearing the display image failed");
    return RT_OK;
}

int rt_get_counters(rt_context* c, rt_counters* out, int reset) {
    if (!c || !out) return c ? Fail(c, RT_ERR_INVALID, "rt_get_counters: NULL out") : RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    LaunchReduceTally(c->stream, c->dTally, c->maxWaves, c->dTotals, kCounterSlots);
    HIP_TRY(c, hipGetLastError());
    unsigned long long t[kCounterSlots];
    HIP_TRY(c, hipMemcpyAsync(t, c->dTotals, sizeof t, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->rays = t[kRays]; out->node_pops = t[kPops]; out->boxes_tested = t[kBoxes]; out->tris_tested = t[kTris];
    out->noise_fetches = t[kNoise]; out->sky_fetches = t[kSky]; out->paths = t[kPaths] + c->hostPaths; out->hits = t[kHits];
    for (int k = 0; k < 4; ++k) { out->sched_rounds[k] = t[kRoundsNew + k]; out->sched_lanes[k] = t[kLanesNew + k]; out->sched_cycles[k] = t[kCyclesNew + k]; }
    if (reset) c->hostPaths = 0;
    if (reset) HIP_TRY(c, hipMemsetAsync(c->dTotals, 0, sizeof t, c->stream));
    return RT_OK;
}

int rt_set_shard(rt_context* c, uint32_t rank, uint32_t world) {
    if (!c) return RT_ERR_INVALID;
    if (world == 0 || rank >= world) return Fail(c, RT_ERR_INVALID, "rt_set_shard: rank %u of %u", rank, world);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sp.shardRank = rank; c->sp.shardWorld = world;
    return AllocShard(c);
}

int rt_shard_tiles(const rt_context* c, uint32_t* tilesPerRank) {
    if (!c || !tilesPerRank) return RT_ERR_INVALID;
    *tilesPerRank = c->sp.shardTiles;
    return RT_OK;
}

int rt_shard_device_ptr(rt_context* c, void** p) {
    if (!c || !p) return RT_ERR_INVALID;
    *p = c->dAccum;
    return RT_OK;
}

int rt_shard_export(rt_context* c, void* dstDevice) {
    if (!c || !dstDevice) return c ? Fail(c, RT_ERR_INVALID, "rt_shard_export: NULL destination") : RT_ERR_INVALID;
    if (!c->dAccum) return Fail(c, RT_ERR_STATE, "rt_shard_export: no accumulation buffer (rt_set_shard failed earlier)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dstDevice, c->dAccum, (size_t)c->sp.shardTiles * 64 * 16, hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

int rt_accum_assemble(rt_context* c, const void* gatheredDevice, uint32_t world) {
    if (!c || !gatheredDevice || world == 0) return c ? Fail(c, RT_ERR_INVALID, "rt_accum_assemble: bad argument") : RT_ERR_INVALID;
    if (c->sp.shardWorld != 1) return Fail(c, RT_ERR_STATE, "rt_accum_assemble: the assembling context must be unsharded");
    if (!c->dAccum) return Fail(c, RT_ERR_STATE, "rt_accum_assemble: no accumulation buffer (rt_set_shard failed earlier)");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t nTiles = c->sp.tilesX * c->sp.tilesY, per = (nTiles + world - 1) / world;
    LaunchRetile(c->stream, gatheredDevice, c->dAccum, nTiles, world, per);      // one pass: tiles move as 1 KiB blocks
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

int rt_set_stream(rt_context* c, void* hipStream) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hipStream ? (hipStream_t)hipStream : c->ownStream;
    return RT_OK;
}

int rt_kernel_time(rt_context* c, double* ms_total, uint64_t* launches, int reset) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ResolveEvents(c);
    if (rc != RT_OK) return rc;
    if (ms_total) *ms_total = c->kernelMs;
    if (launches) *launches = c->kernelLaunches;
    if (reset) { c->kernelMs = 0.0; c->kernelLaunches = 0; }
    return RT_OK;
}

int rt_set_counting(rt_context* c, int on) {
    if (!c) return RT_ERR_INVALID;
    c->countAll = on != 0;
    return RT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// rt_group: one image rendered by several GPUs of this process (tile sharding), shards gathered over RCCL.
// RCCL is looked up at run time (dlopen) so that the library has no load-time dependency on it: programs that use
// one GPU never touch it, and in a process that already has an RCCL loaded (PyTorch brings its own) that one serves.
// ---------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl& LoadRccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.CommInitAll = (decltype(x.CommInitAll))dlsym(x.lib, "ncclCommInitAll");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.lib, "ncclCommDestroy");
        x.GroupStart = (decltype(x.GroupStart))dlsym(x.lib, "ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))dlsym(x.lib, "ncclGroupEnd");
        x.Send = (decltype(x.Send))dlsym(x.lib, "ncclSend");
        x.Recv = (decltype(x.Recv))dlsym(x.lib, "ncclRecv");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.lib, "ncclGetErrorString");
        x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv && x.GetErrorString;
        return x;
    }();
    return r;
}

}  // namespace

struct rt_group {
    std::vector<int> devices;
    std::vector<rt_context*> ranks;      // sharded contexts, rank r on devices[r]
    rt_context* root = nullptr;          // unsharded context on devices[0]: receives the assembled image
    std::vector<ncclComm_t> comms;
    // The exchange runs beside the tracing, not behind it. Per rank: two staging buffers and a stream of its own for the
    // send. A gather copies the rank's shard into a staging buffer on the tracer's stream (a local copy of a few MB;
    // the frames queued next follow at once) and sends from there on the communication stream. Rank 0 receives on its
    // communication stream into one of two receive buffers, and the root context assembles on the root's stream.
    struct PerRank {
        hipStream_t comm = nullptr;
        void* stage[2] = {nullptr, nullptr};
        hipEvent_t exported[2] = {nullptr, nullptr};   // on the tracer's stream: the shard is in stage[b]
        hipEvent_t sent[2] = {nullptr, nullptr};       // on the communication stream: stage[b] may be overwritten
        bool haveSent[2] = {false, false};
    };
    std::vector<PerRank> per;
    void* dGathered[2] = {nullptr, nullptr};           // on devices[0]: world shards back to back
    hipEvent_t gathered[2] = {nullptr, nullptr};       // on rank 0's communication stream behind the receives
    hipEvent_t assembled[2] = {nullptr, nullptr};      // on the root's stream behind the assembly (receives into that buffer wait for it)
    bool haveAssembled[2] = {false, false};
    uint64_t gathers = 0;
    size_t shardFloats = 0;
    std::string error;
};

namespace {
int GroupFail(rt_group* g, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (g) g->error = buf;
    g_globalError = buf;
    return code;
}
}  // namespace

extern "C" {

const char* rt_group_last_error(const rt_group* g) { return g ? g->error.c_str() : g_globalError.c_str(); }

void rt_group_destroy(rt_group* g) {
    if (!g) return;
    Rccl& nc = LoadRccl();
    for (size_t r = 0; r < g->ranks.size(); ++r) if (g->ranks[r]) (void)rt_sync(g->ranks[r]);
    for (size_t r = 0; r < g->per.size(); ++r)
        if (g->per[r].comm) { (void)hipSetDevice(g->devices[r]); (void)hipStreamSynchronize(g->per[r].comm); }
    if (g->root) (void)rt_sync(g->root);
    for (size_t r = 0; r < g->comms.size(); ++r)
        if (g->comms[r] && nc.ok) { (void)hipSetDevice(g->devices[r]); (void)nc.CommDestroy(g->comms[r]); }
    for (size_t r = 0; r < g->per.size(); ++r) {
        (void)hipSetDevice(g->devices[r]);
        rt_group::PerRank& pr = g->per[r];
        if (pr.comm) { (void)hipStreamSynchronize(pr.comm); (void)hipStreamDestroy(pr.comm); }
        for (int b = 0; b < 2; ++b) {
            if (pr.stage[b]) (void)hipFree(pr.stage[b]);
            if (pr.exported[b]) (void)hipEventDestroy(pr.exported[b]);
            if (pr.sent[b]) (void)hipEventDestroy(pr.sent[b]);
        }
    }
    if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
    for (int b = 0; b < 2; ++b) {
        if (g->gathered[b]) (void)hipEventDestroy(g->gathered[b]);
        if (g->assembled[b]) (void)hipEventDestroy(g->assembled[b]);
        if (g->dGathered[b]) (void)hipFree(g->dGathered[b]);
    }
    for (rt_context* c : g->ranks) rt_destroy(c);
    rt_destroy(g->root);
    delete g;
}

int rt_group_create(const int* devices, uint32_t nDevices, uint32_t width, uint32_t height, rt_group** out) try {
    if (!out) return GroupFail(nullptr, RT_ERR_INVALID, "rt_group_create: out is NULL");
    *out = nullptr;
    if (!devices || nDevices == 0 || nDevices > 64) return GroupFail(nullptr, RT_ERR_INVALID, "rt_group_create: need 1..64 devices");
    for (uint32_t a = 0; a < nDevices; ++a)
        for (uint32_t b = a + 1; b < nDevices; ++b)
            if (devices[a] == devices[b]) return GroupFail(nullptr, RT_ERR_INVALID, "rt_group_create: device %d listed twice (one rank per GPU)", devices[a]);
    Rccl& nc = LoadRccl();
    if (!nc.ok) return GroupFail(nullptr, RT_ERR_DEVICE, "rt_group_create: RCCL (librccl.so.1) could not be loaded");
    rt_group* g = new rt_group();
    g->devices.assign(devices, devices + nDevices);
    auto fail = [&](int rc, const char* what) { std::string msg = std::string(what) + ": " + rt_last_global_error(); rt_group_destroy(g); return GroupFail(nullptr, rc, "%s", msg.c_str()); };
    for (uint32_t r = 0; r < nDevices; ++r) {
        rt_context* c = nullptr;
        int rc = rt_create(devices[r], width, height, &c);
        if (rc != RT_OK) return fail(rc, "rt_group_create: rank context");
        g->ranks.push_back(c);
        rc = rt_set_shard(c, r, nDevices);
        if (rc != RT_OK) { g_globalError = c->error; return fail(rc, "rt_group_create: rt_set_shard"); }
    }
    int rc = rt_create(devices[0], width, height, &g->root);
    if (rc != RT_OK) return fail(rc, "rt_group_create: root context");
    g->shardFloats = (size_t)g->ranks[0]->sp.shardTiles * 64 * 4;
    const size_t shardBytes = g->shardFloats * sizeof(float);
    g->per.resize(nDevices);
    bool ok = true;
    for (uint32_t r = 0; r < nDevices && ok; ++r) {
        rt_group::PerRank& pr = g->per[r];
        ok = hipSetDevice(devices[r]) == hipSuccess && hipStreamCreateWithFlags(&pr.comm, hipStreamNonBlocking) == hipSuccess;
        for (int b = 0; b < 2 && ok; ++b)
            ok = hipMalloc(&pr.stage[b], shardBytes) == hipSuccess && hipEventCreateWithFlags(&pr.exported[b], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&pr.sent[b], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipSetDevice(devices[0]) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b)
        ok = hipMalloc(&g->dGathered[b], shardBytes * nDevices) == hipSuccess && hipEventCreateWithFlags(&g->gathered[b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&g->assembled[b], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g_globalError = "device memory, streams or events for the exchange";
        return fail(RT_ERR_DEVICE, "rt_group_create");
    }
    g->comms.assign(nDevices, nullptr);
    ncclResult_t nr = nc.CommInitAll(g->comms.data(), (int)nDevices, devices);
    if (nr != ncclSuccess) { g_globalError = nc.GetErrorString(nr); return fail(RT_ERR_DEVICE, "rt_group_create: ncclCommInitAll"); }
    *out = g;
    return RT_OK;
} catch (const std::exception& e) {
    return GroupFail(nullptr, RT_ERR_DEVICE, "rt_group_create: %s", e.what());
}

uint32_t rt_group_size(const rt_group* g) { return g ? (uint32_t)g->ranks.size() : 0u; }
rt_context* rt_group_rank(rt_group* g, uint32_t rank) { return (g && rank < g->ranks.size()) ? g->ranks[rank] : nullptr; }
rt_context* rt_group_root(rt_group* g) { return g ? g->root : nullptr; }

int rt_group_render(rt_group* g, uint32_t firstFrameIndex, uint32_t nFrames, const float* seeds3) {
    if (!g) return RT_ERR_INVALID;
    for (rt_context* c : g->ranks) {
        int rc = rt_render(c, firstFrameIndex, nFrames, seeds3);
        if (rc != RT_OK) return GroupFail(g, rc, "rt_group_render: %s", rt_last_error(c));
    }
    return RT_OK;
}

int rt_group_gather(rt_group* g) {
    if (!g) return RT_ERR_INVALID;
    Rccl& nc = LoadRccl();
    const uint32_t world = (uint32_t)g->ranks.size();
    const int b = (int)(g->gathers & 1u);
    const size_t shardBytes = g->shardFloats * sizeof(float);
    auto hipFail = [&](const char* what) { return GroupFail(g, RT_ERR_DEVICE, "rt_group_gather: %s failed", what); };
    // 1. every rank copies its shard into stage[b] behind its frames (tracer's stream), once the send of two gathers ago
    //    has left that buffer; its communication stream waits for the copy
    for (uint32_t r = 0; r < world; ++r) {
        rt_group::PerRank& pr = g->per[r];
        rt_context* c = g->ranks[r];
        if (hipSetDevice(g->devices[r]) != hipSuccess) return hipFail("hipSetDevice");
        if (pr.haveSent[b] && hipStreamWaitEvent(c->stream, pr.sent[b], 0) != hipSuccess) return hipFail("ordering the export behind the previous send");
        if (hipMemcpyAsync(pr.stage[b], c->dAccum, shardBytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return hipFail("exporting the shard");
        if (hipEventRecord(pr.exported[b], c->stream) != hipSuccess || hipStreamWaitEvent(pr.comm, pr.exported[b], 0) != hipSuccess)
            return hipFail("ordering the send behind the export");
    }
    // 2. the receives into dGathered[b] wait for the assembly that read it two gathers ago
    if (hipSetDevice(g->devices[0]) != hipSuccess) return hipFail("hipSetDevice");
    if (g->haveAssembled[b] && hipStreamWaitEvent(g->per[0].comm, g->assembled[b], 0) != hipSuccess) return hipFail("ordering the receives behind the previous assembly");
    // 3. one group call: every rank sends from its staging buffer on its communication stream, rank 0 receives the
    //    shards back to back. Every peer has its own xGMI link to rank 0, so the sends run side by side.
    ncclResult_t nr = nc.GroupStart();
    for (uint32_t r = 0; r < world && nr == ncclSuccess; ++r)
        nr = nc.Send(g->per[r].stage[b], g->shardFloats, ncclFloat, 0, g->comms[r], g->per[r].comm);
    for (uint32_t r = 0; r < world && nr == ncclSuccess; ++r)
        nr = nc.Recv(static_cast<float*>(g->dGathered[b]) + (size_t)r * g->shardFloats, g->shardFloats, ncclFloat, (int)r, g->comms[0], g->per[0].comm);
    ncclResult_t ne = nc.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return GroupFail(g, RT_ERR_DEVICE, "rt_group_gather: RCCL: %s", nc.GetErrorString(nr));
    for (uint32_t r = 0; r < world; ++r) {
        if (hipSetDevice(g->devices[r]) != hipSuccess || hipEventRecord(g->per[r].sent[b], g->per[r].comm) != hipSuccess) return hipFail("hipEventRecord");
        g->per[r].haveSent[b] = true;
    }
    // 4. the root context assembles on its own stream once the receives are done
    if (hipSetDevice(g->devices[0]) != hipSuccess) return hipFail("hipSetDevice");
    if (hipEventRecord(g->gathered[b], g->per[0].comm) != hipSuccess || hipStreamWaitEvent(g->root->stream, g->gathered[b], 0) != hipSuccess)
        return hipFail("ordering the assembly behind the receives");
    int rc = rt_accum_assemble(g->root, g->dGathered[b], world);
    if (rc != RT_OK) return GroupFail(g, rc, "rt_group_gather: %s", rt_last_error(g->root));
    if (hipEventRecord(g->assembled[b], g->root->stream) != hipSuccess) return hipFail("hipEventRecord");
    g->haveAssembled[b] = true;
    ++g->gathers;
    return RT_OK;
}

int rt_group_sync(rt_group* g) {
    if (!g) return RT_ERR_INVALID;
    for (rt_context* c : g->ranks) { int rc = rt_sync(c); if (rc != RT_OK) return GroupFail(g, rc, "rt_group_sync: %s", rt_last_error(c)); }
    for (size_t r = 0; r < g->per.size(); ++r)
        if (hipSetDevice(g->devices[r]) != hipSuccess || hipStreamSynchronize(g->per[r].comm) != hipSuccess)
            return GroupFail(g, RT_ERR_DEVICE, "rt_group_sync: communication stream of rank %zu", r);
    int rc = rt_sync(g->root);
    if (rc != RT_OK) return GroupFail(g, rc, "rt_group_sync: %s", rt_last_error(g->root));
    return RT_OK;
}

}  // extern "C"

