// C ABI over the host scene model (declared in include/rt_host.h).
#include "rt_host.h"

#include <new>
#include <string>
#include <vector>

#include "scene_cache.h"
#include "scene_data.h"

// Nothing may throw across the C boundary: allocation failures and whatever else the C++ below raises
// (std::system_error from the builder's threads, std::length_error ...) become status codes.
template <class F>
static int Guarded(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return RT_ERR_DEVICE; }      // out of memory
    catch (...) { return RT_ERR_INVALID; }
}

struct rth_scene {
    rta::SceneData data;
    std::vector<rt_material> mats;
};

namespace {

rt_material ToAbi(const rta::Material& m) {
    rt_material o;
    o.colour[0] = m.colour.x; o.colour[1] = m.colour.y; o.colour[2] = m.colour.z;
    o.specularColour[0] = m.specularColour.x; o.specularColour[1] = m.specularColour.y; o.specularColour[2] = m.specularColour.z;
    o.roughness = m.roughness; o.metallic = m.metallic; o.transparency = m.transparency;
    o.refractionIndex = m.refractionIndex;
    o.isLight = m.isLight != 0.0f ? 1 : 0;     // uploaded as an int 0/1 (src/Scene.cpp:150)
    return o;
}

rta::Material FromAbi(const rt_material& m) {
    rta::Material o;
    o.colour = {m.colour[0], m.colour[1], m.colour[2]};
    o.specularColour = {m.specularColour[0], m.specularColour[1], m.specularColour[2]};
    o.roughness = m.roughness; o.metallic = m.metallic; o.transparency = m.transparency;
    o.refractionIndex = m.refractionIndex; o.isLight = m.isLight ? 1.0f : 0.0f;
    return o;
}

rth_scene* Wrap(rta::SceneData&& d) {
    rth_scene* s = new rth_scene();
    s->data = std::move(d);
    for (const rta::Material& m : s->data.materials) s->mats.push_back(ToAbi(m));
    return s;
}

}  // namespace

extern "C" {

int rth_scene_load(const char* const* paths, uint32_t nPaths, int buildThreads, rth_scene** out) {
    if (!out || (nPaths && !paths)) return RT_ERR_INVALID;
    *out = nullptr;
    return Guarded([&] {
        std::vector<std::string> p;
        for (uint32_t i = 0; i < nPaths; ++i) p.emplace_back(paths[i] ? paths[i] : "");
        *out = Wrap(rta::LoadScene(p, buildThreads));
        return (int)RT_OK;
    });
}

int rth_scene_load_cached(const char* const* paths, uint32_t nPaths, int buildThreads, const char* cacheDir, rth_scene** out, int* cacheHit) {
    if (!out || (nPaths && !paths)) return RT_ERR_INVALID;
    *out = nullptr;
    if (cacheHit) *cacheHit = 0;
    return Guarded([&] {
        std::vector<std::string> p;
        for (uint32_t i = 0; i < nPaths; ++i) p.emplace_back(paths[i] ? paths[i] : "");
        rta::CacheResult r = rta::CacheResult::kDisabled;
        *out = Wrap(rta::LoadSceneCached(p, buildThreads, cacheDir ? cacheDir : "", &r));
        if (cacheHit) *cacheHit = r == rta::CacheResult::kHit ? 1 : 0;
        return (int)RT_OK;
    });
}

int rth_scene_build(const float* tri9, const int32_t* matSlot, uint32_t nTri, const rt_material* materials,
                    uint32_t nMaterials, int buildThreads, int maxLeaf, rth_scene** out) {
    if (!out || (nTri && (!tri9 || !matSlot)) || (nMaterials && !materials)) return RT_ERR_INVALID;
    *out = nullptr;
    return Guarded([&] {
        std::vector<rta::Triangle> tris;
        tris.reserve(nTri);
        for (uint32_t i = 0; i < nTri; ++i) {
            const float* p = tri9 + (size_t)i * 9;
            tris.emplace_back(rta::Vec3(p[0], p[1], p[2]), rta::Vec3(p[3], p[4], p[5]), rta::Vec3(p[6], p[7], p[8]), matSlot[i]);
        }
        std::vector<rta::Material> mats;
        for (uint32_t i = 0; i < nMaterials; ++i) mats.push_back(FromAbi(materials[i]));
        *out = Wrap(rta::BuildScene(std::move(tris), std::move(mats), buildThreads,
                                    maxLeaf > 0 ? maxLeaf : rta::BvhTree::kDefaultLeafTriangles));
        return (int)RT_OK;
    });
}

void rth_scene_free(rth_scene* s) { delete s; }
uint32_t rth_scene_triangle_count(const rth_scene* s) { return s->data.TriangleCount(); }
uint32_t rth_scene_node_count(const rth_scene* s) { return s->data.NodeCount(); }
uint32_t rth_scene_material_count(const rth_scene* s) { return (uint32_t)s->mats.size(); }
const float* rth_scene_tri12(const rth_scene* s) { return s->data.tri12.data(); }
const int32_t* rth_scene_matidx(const rth_scene* s) { return s->data.matIdx.data(); }
const float* rth_scene_node8(const rth_scene* s) { return s->data.node8.data(); }
const rt_material* rth_scene_materials(const rth_scene* s) { return s->mats.data(); }
void rth_scene_stats(const rth_scene* s, rth_bvh_stats* o) {
    const rta::BvhStats& st = s->data.stats;
    o->max_depth = st.maxDepth; o->leaves = st.leaves; o->splits = st.splits;
    o->degenerate_splits = st.degenerateSplits; o->leaf_depth_sum = st.leafDepthSum; o->build_ms = st.buildMs;
}
uint32_t rth_scene_warning_count(const rth_scene* s) { return (uint32_t)s->data.warnings.size(); }
const char* rth_scene_warning(const rth_scene* s, uint32_t i) { return i < s->data.warnings.size() ? s->data.warnings[i].c_str() : ""; }

int rth_scene_send(const rth_scene* s, rt_context* ctx) {
    if (!s || !ctx) return RT_ERR_INVALID;
    return Guarded([&] {
        int rc = rt_scene_upload(ctx, s->data.tri12.data(), s->data.matIdx.data(), s->data.TriangleCount(),
                                 s->data.node8.data(), s->data.NodeCount());
        if (rc != RT_OK) return rc;
        uint32_t n = (uint32_t)s->mats.size();
        return rt_set_materials(ctx, s->mats.data(), n > 16 ? 16 : n);
    });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// controls and headless frame loop
// ---------------------------------------------------------------------------------------------
#include <cstring>

#include "app.h"
#include "config_parser.h"
#include "controls.h"
#include "image_io.h"
#include "video.h"

struct rth_camera { rta::Camera cam; };
struct rth_app { rta::HeadlessApp app; rth_app(rt_context* c, uint32_t w, uint32_t h) : app(c, w, h) {} };

namespace {
void FillCamera(const rta::Camera& c, float* o) {
    o[0] = c.Position().x; o[1] = c.Position().y; o[2] = c.Position().z;
    o[3] = c.Facing().x; o[4] = c.Facing().y; o[5] = c.Facing().z;
    o[6] = c.Fov(); o[7] = c.ViewportWidth(); o[8] = c.ViewportDistance();
}
}  // namespace

extern "C" {

rth_camera* rth_camera_create(void) { return new rth_camera(); }
void rth_camera_destroy(rth_camera* c) { delete c; }
void rth_camera_key(rth_camera* c, int key, int action) { c->cam.OnEvent(rta::KeyEvent{key, action}); }
int rth_camera_tick(rth_camera* c) { return c->cam.Tick() ? 1 : 0; }
void rth_camera_get(const rth_camera* c, float* out9) { FillCamera(c->cam, out9); }
void rth_camera_set_pose(rth_camera* c, const float* p, const float* f) {
    c->cam.SetPosition({p[0], p[1], p[2]});
    c->cam.SetFacing({f[0], f[1], f[2]});
}
int rth_bounce_limit_from_key(int key) { auto b = rta::BounceLimitFromKey(key); return b ? (int)*b : -1; }
void rth_rand_seed_triples(uint32_t n, float* out3) { rta::GlibcRand r; for (uint32_t i = 0; i < n; ++i) r.NextSeedTriple(out3 + 3 * (size_t)i); }

int rth_ini_object_paths(const char* iniPath, char* buf, uint32_t bufSize, char* skyboxDir, uint32_t skyboxDirSize) {
    try {
        std::string sky;
        std::vector<std::string> paths = rta::HeadlessApp::ObjectPathsFromIni(iniPath, &sky);
        std::string joined;
        for (size_t i = 0; i < paths.size(); ++i) { if (i) joined += "\n"; joined += paths[i]; }
        if (buf && bufSize) { std::strncpy(buf, joined.c_str(), bufSize - 1); buf[bufSize - 1] = 0; }
        if (skyboxDir && skyboxDirSize) { std::strncpy(skyboxDir, sky.c_str(), skyboxDirSize - 1); skyboxDir[skyboxDirSize - 1] = 0; }
        return (int)paths.size();
    } catch (const std::exception&) {
        return -1;
    }
}
int rth_ini_values(const char* iniPath, const char* section, const char* name, char* buf, uint32_t bufSize, char* errBuf, uint32_t errSize) {
    if (!iniPath || !section || !name) return -1;
    auto put = [](char* dst, uint32_t cap, const std::string& text) {
        if (dst && cap) { std::strncpy(dst, text.c_str(), cap - 1); dst[cap - 1] = 0; }
    };
    try {
        rta::ConfigParser ini(iniPath);
        std::string joined;
        const std::vector<std::string> values = ini.Values(section, name);
        for (const std::string& v : values) { joined += v; joined += '\n'; }
        put(buf, bufSize, joined);
        put(errBuf, errSize, "");
        return (int)values.size();
    } catch (const std::exception& e) {
        put(errBuf, errSize, e.what());
        return -1;
    }
}
int rth_ini_number(const char* iniPath, const char* section, const char* name, uint32_t pos, int kind, double* out) {
    if (!iniPath || !section || !name || !out) return RT_ERR_INVALID;
    try {
        rta::ConfigParser ini(iniPath);
        if (kind == 0) *out = (double)ini.Get<int>(section, name, pos);
        else if (kind == 1) *out = (double)ini.Get<float>(section, name, pos);
        else if (kind == 2) *out = ini.Get<bool>(section, name, pos) ? 1.0 : 0.0;
        else return RT_ERR_INVALID;
        return RT_OK;
    } catch (const std::exception&) {
        return RT_ERR_INVALID;
    }
}
int rth_write_png_rgb8(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb, int flip) {
    if (!path || !rgb) return RT_ERR_INVALID;
    return Guarded([&] { return rta::WritePngRgb8(path, w, h, rgb, flip != 0) ? (int)RT_OK : (int)RT_ERR_INVALID; });
}

int rth_read_png(const char* path, int channels, uint32_t* width, uint32_t* height, uint8_t* pixels, uint64_t capacity) {
    if (!path) return RT_ERR_INVALID;
    return Guarded([&] {
        rta::Image8 img;
        if (!rta::LoadPng(path, channels, img)) return (int)RT_ERR_INVALID;
        if (width) *width = img.width;
        if (height) *height = img.height;
        if (pixels) {
            if (capacity < img.pixels.size()) return (int)RT_ERR_INVALID;
            std::memcpy(pixels, img.pixels.data(), img.pixels.size());
        }
        return (int)RT_OK;
    });
}

int rth_app_create(rt_context* ctx, uint32_t w, uint32_t h, rth_app** out) {
    if (!ctx || !out) return RT_ERR_INVALID;
    *out = nullptr;
    return Guarded([&] { *out = new rth_app(ctx, w, h); return (int)RT_OK; });
}
void rth_app_destroy(rth_app* a) { delete a; }
int rth_app_load_objects(rth_app* a, const char* const* paths, uint32_t n, int threads) {
    if (!a || (n && !paths)) return RT_ERR_INVALID;
    return Guarded([&] {
        std::vector<std::string> p;
        for (uint32_t i = 0; i < n; ++i) p.emplace_back(paths[i] ? paths[i] : "");
        return a->app.LoadObjects(p, threads);
    });
}
void rth_app_set_scene_cache(rth_app* a, const char* dir) { if (a) (void)Guarded([&] { a->app.sceneCacheDir = dir ? dir : ""; return (int)RT_OK; }); }
void rth_app_key(rth_app* a, int key, int action) { a->app.OnKey(key, action); }
int rth_app_tick(rth_app* a) { return a ? Guarded([&] { return a->app.Tick(); }) : (int)RT_ERR_INVALID; }
int rth_app_read_display(rth_app* a, float* rgba, uint8_t* rgb8) { return a ? Guarded([&] { return a->app.ReadDisplay(rgba, rgb8); }) : (int)RT_ERR_INVALID; }
void rth_app_get_state(const rth_app* a, rth_app_state* o) {
    rta::HeadlessApp& app = const_cast<rth_app*>(a)->app;
    FillCamera(app.GetCamera(), o->camera);
    for (int k = 0; k < 3; ++k) o->last_seed[k] = app.LastSeed()[k];
    o->next_frame_index = app.FrameIndex(); o->bounce_limit = app.BounceLimit();
    o->recorder_state = (int32_t)app.GetRecorder().GetState();
    o->poses_left = (uint32_t)app.GetRecorder().PosesLeft(); o->images_written = app.GetRecorder().ImagesWritten();
}
void rth_app_configure(rth_app* a, int bloom, int postEveryFrame, uint32_t framesPerImage, uint32_t recordEveryFrames, const char* outputDir,
                       int flip) {
    a->app.bloom = bloom != 0; a->app.postprocessEveryFrame = postEveryFrame != 0;
    rta::Recorder& r = a->app.GetRecorder();
    if (framesPerImage) r.framesPerImage = framesPerImage;
    if (recordEveryFrames) r.recordEveryFrames = recordEveryFrames;
    if (outputDir) r.outputDir = outputDir;
    r.flipVertically = flip != 0;
}
void rth_app_push_pose(rth_app* a, const float* p, const float* f) {
    a->app.GetRecorder().PushPose(rta::Pose{{p[0], p[1], p[2]}, {f[0], f[1], f[2]}});
}
int rth_app_save_poses(const rth_app* a, const char* path) {
    if (!a || !path) return RT_ERR_INVALID;
    return Guarded([&] { return const_cast<rth_app*>(a)->app.GetRecorder().SavePoses(path) ? (int)RT_OK : (int)RT_ERR_INVALID; });
}
int rth_app_load_poses(rth_app* a, const char* path) {
    if (!a || !path) return RT_ERR_INVALID;
    return Guarded([&] { return a->app.GetRecorder().LoadPoses(path) ? (int)RT_OK : (int)RT_ERR_INVALID; });
}

int rth_video_render(rt_context* ctx, uint32_t width, uint32_t height, const float* poses6, uint32_t nPoses, uint32_t spp, uint32_t rank,
                     uint32_t world, const char* outputDir, uint32_t flags, int compression, uint32_t writerThreads, rth_video_stats* out) {
    if (!ctx || (nPoses && !poses6)) return RT_ERR_INVALID;
    return Guarded([&] {
        rta::VideoJob job;
        for (uint32_t k = 0; k < nPoses; ++k) {
            const float* p = poses6 + 6 * (size_t)k;
            job.poses.push_back(rta::Pose{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}});
        }
        job.samplesPerPixel = spp; job.rank = rank; job.world = world;
        if (outputDir) job.outputDir = outputDir;
        job.bloom = (flags & RTH_VIDEO_BLOOM) != 0; job.flipVertically = (flags & RTH_VIDEO_FLIP) != 0;
        job.resume = (flags & RTH_VIDEO_RESUME) != 0; job.writeImages = (flags & RTH_VIDEO_WRITE) != 0;
        job.compression = compression; job.writerThreads = writerThreads ? writerThreads : 2;
        rta::VideoStats st;
        int rc = rta::RenderVideo(ctx, width, height, job, &st);
        if (out) {
            out->rendered = st.rendered; out->skipped = st.skipped; out->written = st.written; out->write_failures = st.writeFailures;
            out->wall_seconds = st.wallSeconds; out->trace_seconds = st.traceSeconds; out->encode_seconds = st.encodeSeconds;
            out->stall_seconds = st.stallSeconds; out->rays = st.rays;
        }
        return rc;
    });
}

void rth_video_seeds(uint32_t firstPose, uint32_t nPoses, uint32_t spp, float* out3) {
    if (!out3) return;
    (void)Guarded([&] {
        std::vector<float> s = rta::VideoSeeds(firstPose, nPoses, spp);
        std::memcpy(out3, s.data(), s.size() * sizeof(float));
        return (int)RT_OK;
    });
}

int rth_read_poses(const char* path, float* poses6, uint32_t capacity, uint32_t* nPoses) {
    if (!path || !nPoses) return RT_ERR_INVALID;
    return Guarded([&] {
        std::vector<rta::Pose> poses;
        if (!rta::ReadPoseFile(path, poses)) return (int)RT_ERR_INVALID;
        *nPoses = (uint32_t)poses.size();
        for (uint32_t k = 0; poses6 && k < capacity && k < poses.size(); ++k) {
            const rta::Pose& p = poses[k];
            float* o = poses6 + 6 * (size_t)k;
            o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z; o[3] = p.facing.x; o[4] = p.facing.y; o[5] = p.facing.z;
        }
        return (int)RT_OK;
    });
}

}  // extern "C"
