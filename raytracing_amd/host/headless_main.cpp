// raytracer_headless — the reference's executable without a window (src/main.cpp:337-367):
// reads RayTracer.ini, loads the sky box, the noise texture and the objects, then runs the frame
// loop for a number of frames (optionally replaying key events or a pose file) and writes PNGs.
//
//   raytracer_headless [--ini RayTracer.ini] [--noise ./Textures/Noise/rgbNoise.png]
//                      [--width 1280 --height 720] [--frames 64] [--device 0] [--threads 8]
//                      [--keys "K4 W+ 10 W- 50"] [--poses file --spp 256] [--out shots] [--no-bloom] [--flip]
//                      [--offline [--rank r --world n]] [--cache dir] [--ray-count 1]
// --ray-count N: the shader's compile-time RAY_COUNT (src/shaders/Fragment.glsl:5), rays per pixel and frame; 1 as checked in.
// --gpus N: one image on GPUs 0..N-1 of this process (tile sharding, shards gathered over RCCL: rt_group_*); renders
// --frames frames from the default camera (no key replay, no poses) and writes <out>.png. The bloom chain runs once, on the
// gathered image; the plain loop runs it for every displayed frame and carries its tmp texture over as the reference does,
// so with bloom on the two write the same image for --frames 1 only (without bloom: for any number of frames).
// --cache: keep the parsed objects and the built tree in that directory, keyed by the object files' bytes.
// --poses alone replays the Recorder's playback frame by frame (one launch and one bloom pass per displayed frame, as
// the reference does). --poses with --offline is the batch form: all samples of a pose in one launch, images written by
// writer threads, existing images skipped; start one process per GPU with --device r --rank r --world n to share the
// poses of a job among n GPUs.
// --keys: tokens separated by blanks: NAME = tap, NAME+ = press, NAME- = release, a number = that many frames.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "app.h"
#include "image_io.h"
#include "scene_cache.h"
#include "video.h"
#include "rt_mi355x.h"

using namespace rta;

static int Die(const char* what, rt_context* ctx = nullptr) {
    std::fprintf(stderr, "error: %s%s%s\n", what, ctx ? ": " : "", ctx ? rt_last_error(ctx) : "");
    return 1;
}

// One image on several GPUs of this process: every rank gets the scene and the textures, renders its tiles, and the
// shards are gathered to the root context (RCCL) for the bloom chain and the PNG.
static int RenderOnGroup(uint32_t gpus, uint32_t width, uint32_t height, uint32_t frames, int threads, const std::string& skyDir,
                         const std::string& noisePath, const std::vector<std::string>& objects, const std::string& cacheDir,
                         const std::string& outDir, bool bloom, bool flip, uint32_t rayCount) {
    std::vector<int> devices(gpus);
    for (uint32_t i = 0; i < gpus; ++i) devices[i] = (int)i;
    rt_group* group = nullptr;
    if (rt_group_create(devices.data(), gpus, width, height, &group) != RT_OK) { std::fprintf(stderr, "error: %s\n", rt_group_last_error(nullptr)); return 1; }
    static const char* kFaces[6] = {"right.png", "left.png", "top.png", "bottom.png", "front.png", "back.png"};
    std::vector<Image8> faces(6);
    const uint8_t* facePtr[6];
    std::string err;
    for (int f = 0; f < 6; ++f) {
        if (!LoadPng(skyDir + "/" + kFaces[f], 4, faces[f], &err)) { std::fprintf(stderr, "Failed to load texture at path: %s/%s (%s)\n", skyDir.c_str(), kFaces[f], err.c_str()); return 1; }
        facePtr[f] = faces[f].pixels.data();
    }
    Image8 noise;
    if (!LoadPng(noisePath, 3, noise, &err)) { std::fprintf(stderr, "Failed to load noise texture at path: %s (%s)\n", noisePath.c_str(), err.c_str()); return 1; }
    SceneData scene = LoadSceneCached(objects, threads, cacheDir);
    std::vector<rt_material> mats;
    for (size_t i = 0; i < scene.materials.size() && i < 16; ++i) {
        const Material& m = scene.materials[i];
        mats.push_back(rt_material{{m.colour.x, m.colour.y, m.colour.z}, {m.specularColour.x, m.specularColour.y, m.specularColour.z},
                                   m.roughness, m.metallic, m.transparency, m.refractionIndex, m.isLight != 0.0f ? 1 : 0});
    }
    Camera camera;
    rt_camera cam{{camera.Position().x, camera.Position().y, camera.Position().z}, {camera.Facing().x, camera.Facing().y, camera.Facing().z},
                  camera.Fov(), camera.ViewportWidth(), camera.ViewportDistance()};
    for (uint32_t r = 0; r < gpus; ++r) {
        rt_context* c = rt_group_rank(group, r);
        if (rt_set_skybox(c, facePtr, faces[0].width) != RT_OK || rt_set_noise(c, noise.pixels.data(), noise.width, noise.height) != RT_OK ||
            rt_scene_upload(c, scene.tri12.data(), scene.matIdx.data(), scene.TriangleCount(), scene.node8.data(), scene.NodeCount()) != RT_OK ||
            rt_set_materials(c, mats.data(), (uint32_t)mats.size()) != RT_OK || rt_set_camera(c, &cam) != RT_OK || rt_set_bounce_limit(c, 4) != RT_OK ||
            rt_set_ray_count(c, rayCount) != RT_OK)
            return Die("setting up a rank", c);
    }
    std::printf("triangles count: %u\n%u GPUs\n", scene.TriangleCount(), gpus);
    GlibcRand rng;
    for (uint32_t k = 0; k < frames; ++k) {
        float seed[3];
        rng.NextSeedTriple(seed);                            // src/Scene.cpp:31
        if (rt_group_render(group, k, 1, seed) != RT_OK) { std::fprintf(stderr, "error: %s\n", rt_group_last_error(group)); return 1; }
    }
    if (rt_group_gather(group) != RT_OK) { std::fprintf(stderr, "error: %s\n", rt_group_last_error(group)); return 1; }
    std::vector<uint8_t> rgb8((size_t)width * height * 3);
    if (rt_postprocess(rt_group_root(group), bloom ? 1 : 0, nullptr, rgb8.data()) != RT_OK) return Die("post-processing", rt_group_root(group));
    const std::string out = outDir + ".png";
    if (!WritePngRgb8(out, width, height, rgb8.data(), flip)) return Die("cannot write the image");
    uint64_t rays = 0;
    for (uint32_t r = 0; r < gpus; ++r) { rt_counters c; rt_get_counters(rt_group_rank(group, r), &c, 0); rays += c.rays; }
    std::printf("wrote %s after %u frames on %u GPUs (%llu rays)\n", out.c_str(), frames, gpus, (unsigned long long)rays);
    rt_group_destroy(group);
    return 0;
}

int main(int argc, char** argv) {
    std::string ini = "RayTracer.ini", noisePath = "./Textures/Noise/rgbNoise.png", keys, poses, outDir = "shots", cacheDir;
    uint32_t width = 1280, height = 720, frames = 64, spp = 256;   // window size of src/main.cpp:28-29
    int device = 0, threads = 8;
    uint32_t rank = 0, world = 1, gpus = 0, rayCount = 1;
    bool bloom = true, flip = false, offline = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--ini") ini = next(); else if (a == "--noise") noisePath = next(); else if (a == "--width") width = std::atoi(next());
        else if (a == "--height") height = std::atoi(next()); else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--device") device = std::atoi(next()); else if (a == "--threads") threads = std::atoi(next());
        else if (a == "--keys") keys = next(); else if (a == "--poses") poses = next(); else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--out") outDir = next(); else if (a == "--no-bloom") bloom = false; else if (a == "--flip") flip = true;
        else if (a == "--cache") cacheDir = next(); else if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--ray-count") rayCount = std::atoi(next());
        else if (a == "--offline") offline = true; else if (a == "--rank") rank = std::atoi(next()); else if (a == "--world") world = std::atoi(next());
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    std::string skyDir;
    std::vector<std::string> objects;
    try { objects = HeadlessApp::ObjectPathsFromIni(ini, &skyDir); }
    catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }

    if (gpus > 0) {
        if (!keys.empty() || !poses.empty()) { std::fprintf(stderr, "--gpus renders a fixed camera: not with --keys or --poses\n"); return 2; }
        return RenderOnGroup(gpus, width, height, frames, threads, skyDir, noisePath, objects, cacheDir, outDir, bloom, flip, rayCount);
    }
    rt_context* ctx = nullptr;
    if (rt_create(device, width, height, &ctx) != RT_OK) { std::fprintf(stderr, "error: %s\n", rt_last_global_error()); return 1; }
    if (rt_set_ray_count(ctx, rayCount) != RT_OK) return Die("--ray-count", ctx);

    // sky box: right, left, top, bottom, front, back -> +X,-X,+Y,-Y,+Z,-Z (src/main.cpp:52-70)
    static const char* kFaces[6] = {"right.png", "left.png", "top.png", "bottom.png", "front.png", "back.png"};
    std::vector<Image8> faces(6);
    const uint8_t* facePtr[6];
    for (int f = 0; f < 6; ++f) {
        std::string err, p = skyDir + "/" + kFaces[f];
        if (!LoadPng(p, 4, faces[f], &err)) { std::fprintf(stderr, "Failed to load texture at path: %s (%s)\n", p.c_str(), err.c_str()); return 1; }
        if (faces[f].width != faces[f].height || faces[f].width != faces[0].width) return Die("sky box faces must be square and of one size");
        facePtr[f] = faces[f].pixels.data();
    }
    if (rt_set_skybox(ctx, facePtr, faces[0].width) != RT_OK) return Die("rt_set_skybox", ctx);
    Image8 noise;
    std::string err;
    if (!LoadPng(noisePath, 3, noise, &err)) { std::fprintf(stderr, "Failed to load noise texture at path: %s (%s)\n", noisePath.c_str(), err.c_str()); return 1; }
    if (rt_set_noise(ctx, noise.pixels.data(), noise.width, noise.height) != RT_OK) return Die("rt_set_noise", ctx);

    HeadlessApp app(ctx, width, height);
    app.bloom = bloom;
    app.sceneCacheDir = cacheDir;
    if (app.LoadObjects(objects, threads) != RT_OK) return Die("loading objects", ctx);
    for (const std::string& w : app.Scene().warnings) std::printf("%s\n", w.c_str());
    const BvhStats& st = app.Scene().stats;
    std::printf("triangles count: %u\nnumber of bounding boxes: %u\nnumber of total splits: %llu\nnumber of leaf nodes: %u\n"
                "constructing BVH structure took: %.0fms\n", app.Scene().TriangleCount(), app.Scene().NodeCount() + 1,
                (unsigned long long)st.splits, st.leaves, st.buildMs);

    Recorder& rec = app.GetRecorder();
    rec.outputDir = outDir; rec.flipVertically = flip; rec.framesPerImage = spp;
    static const std::map<std::string, int> kNames = {
        {"W", kKeyW}, {"A", kKeyA}, {"S", kKeyS}, {"D", kKeyD}, {"C", kKeyC}, {"R", kKeyR}, {"P", kKeyP}, {"B", kKeyB}, {"SPACE", kKeySpace},
        {"CTRL", kKeyLeftControl}, {"LSHIFT", kKeyLeftShift}, {"RSHIFT", kKeyRightShift}, {"UP", kKeyUp}, {"DOWN", kKeyDown},
        {"LEFT", kKeyLeft}, {"RIGHT", kKeyRight}, {"GRAVE", kKeyGrave}, {"K1", kKey1}, {"K2", kKey2}, {"K3", kKey3}, {"K4", kKey4},
        {"K5", kKey5}, {"K6", kKey6}};
    auto run = [&](uint32_t n) { for (uint32_t k = 0; k < n; ++k) if (app.Tick() != RT_OK) return false; return true; };
    if (!poses.empty() && offline) {
        VideoJob job;
        if (!ReadPoseFile(poses, job.poses)) return Die("cannot read the pose file");
        job.samplesPerPixel = spp; job.rank = rank; job.world = world; job.outputDir = outDir; job.bloom = bloom; job.flipVertically = flip;
        VideoStats vs;
        if (RenderVideo(ctx, width, height, job, &vs) != RT_OK) return Die("rendering", ctx);
        std::printf("rank %u of %u: rendered %u poses, skipped %u, wrote %u images to %s in %.3f s (tracing %.3f s, %.1f Mrays/s)\n", rank, world,
                    vs.rendered, vs.skipped, vs.written, outDir.c_str(), vs.wallSeconds, vs.traceSeconds,
                    vs.wallSeconds > 0 ? vs.rays / vs.wallSeconds / 1e6 : 0.0);
    } else if (!poses.empty()) {
        if (!rec.LoadPoses(poses)) return Die("cannot read the pose file");
        const uint32_t total = (uint32_t)rec.PosesLeft();
        app.OnKey(kKeyP, kPress); app.OnKey(kKeyP, kRelease);
        if (!run((total + 1) * spp + 1)) return Die("rendering", ctx);
        std::printf("wrote %u images to %s\n", rec.ImagesWritten(), outDir.c_str());
    } else {
        std::istringstream ks(keys);
        std::string tok;
        uint32_t done = 0;
        while (ks >> tok) {
            if (std::isdigit((unsigned char)tok[0])) { uint32_t n = std::atoi(tok.c_str()); if (!run(n)) return Die("rendering", ctx); done += n; continue; }
            char mode = tok.back();
            std::string name = (mode == '+' || mode == '-') ? tok.substr(0, tok.size() - 1) : tok;
            auto it = kNames.find(name);
            if (it == kNames.end()) { std::fprintf(stderr, "unknown key %s\n", name.c_str()); return 2; }
            if (mode != '-') app.OnKey(it->second, kPress);
            if (mode != '+') app.OnKey(it->second, kRelease);
        }
        if (done < frames && !run(frames - done)) return Die("rendering", ctx);
        std::vector<uint8_t> rgb8((size_t)width * height * 3);
        if (app.ReadDisplay(nullptr, rgb8.data()) != RT_OK) return Die("reading the image", ctx);
        std::string out = outDir + ".png";
        if (!WritePngRgb8(out, width, height, rgb8.data(), flip)) return Die("cannot write the image");
        rt_counters c;
        rt_get_counters(ctx, &c, 0);
        std::printf("wrote %s after %u frames (%llu rays)\n", out.c_str(), app.FrameIndex(), (unsigned long long)c.rays);
    }
    rt_destroy(ctx);
    return 0;
}
