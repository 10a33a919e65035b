#include "app.h"

#include <sys/stat.h>
#include <zlib.h>

#include <cstdio>
#include <fstream>
#include <sstream>

#include "config_parser.h"
#include "scene_cache.h"
#include "rt_host.h"

namespace rta {

// ------------------------------------------------------------------------------------------ PNG
namespace {
void Put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void Chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t n) {
    Put32(out, (uint32_t)n);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data, data + n);
    Put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}
}  // namespace

bool WritePngRgb8(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgb, bool flip, int compression) {
    if (!width || !height || !rgb) return false;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t* row = rgb + (size_t)(flip ? height - 1 - y : y) * width * 3;
        raw.push_back(0);                                   // filter: none
        raw.insert(raw.end(), row, row + (size_t)width * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), compression < 0 ? 0 : (compression > 9 ? 9 : compression)) != Z_OK) return false;
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    png.reserve(zlen + 64);
    std::vector<uint8_t> ihdr;
    Put32(ihdr, width); Put32(ihdr, height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});               // 8 bits, colour type RGB
    Chunk(png, "IHDR", ihdr.data(), ihdr.size());
    Chunk(png, "IDAT", z.data(), zlen);
    Chunk(png, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

// ------------------------------------------------------------------------------------- Recorder
void Recorder::OnEvent(const KeyEvent& e) {                 // Include/Recorder.h:73-91
    if (e.action != kPress) return;
    if (e.key == kKeyR) {
        if (state_ == kNothing) state_ = kRecording;
        else if (state_ == kRecording) state_ = kNothing;
        return;
    }
    if (e.key == kKeyP && state_ == kNothing) { state_ = kPlayback; sinceSwitch_ = 0; }
}

void Recorder::Tick(HeadlessApp& app) {                     // Include/Recorder.h:36-55
    if (state_ == kRecording && ++sinceRecord_ >= recordEveryFrames) {
        sinceRecord_ = 0;
        poses_.push_back(Pose{app.GetCamera().Position(), app.GetCamera().Facing()});
    }
    if (state_ == kPlayback && ++sinceSwitch_ > framesPerImage) {
        sinceSwitch_ = 1;
        // the image on display (post-bloom result of the frames rendered so far) goes to disk first
        std::vector<uint8_t> rgb8((size_t)app.Width() * app.Height() * 3);
        if (app.ReadDisplay(nullptr, rgb8.data()) == RT_OK) {
            ::mkdir(outputDir.c_str(), 0777);               // the reference never creates the directory
            WritePngRgb8(outputDir + "/raytrace-out-" + std::to_string(imageCounter_++) + ".png", app.Width(), app.Height(), rgb8.data(),
                         flipVertically);
        }
        if (poses_.empty()) { state_ = kNothing; return; }  // the reference pops an empty queue here (undefined behaviour)
        Pose next = poses_.front();
        poses_.pop_front();
        app.GetCamera().SetPosition(next.position);
        app.GetCamera().SetFacing(next.facing);
        app.UploadCamera();
        app.ResetFrameIndex();
    }
}

bool Recorder::SavePoses(const std::string& path) const {
    std::ofstream out(path);
    if (!out) return false;
    out.precision(9);
    for (const Pose& p : poses_)
        out << p.position.x << ' ' << p.position.y << ' ' << p.position.z << ' ' << p.facing.x << ' ' << p.facing.y << ' ' << p.facing.z << '\n';
    return (bool)out;
}

bool Recorder::LoadPoses(const std::string& path) {
    std::ifstream in(path);
    if (!in) return false;
    Pose p;
    while (in >> p.position.x >> p.position.y >> p.position.z >> p.facing.x >> p.facing.y >> p.facing.z) poses_.push_back(p);
    return true;
}

// ------------------------------------------------------------------------------------------ App
HeadlessApp::HeadlessApp(rt_context* ctx, uint32_t width, uint32_t height) : ctx_(ctx), width_(width), height_(height) {
    rt_set_bounce_limit(ctx_, bounceLimit_);
    UploadCamera();
}

std::vector<std::string> HeadlessApp::ObjectPathsFromIni(const std::string& iniPath, std::string* skyboxDir) {
    ConfigParser ini(iniPath);
    if (skyboxDir) *skyboxDir = ini.Get<std::string>("SkyBox", "Path");
    std::string dir = ini.Get<std::string>("Objects", "DirectoryPath");
    std::vector<std::string> names = ini.GetAll<std::string>("Objects", "Filenames");
    for (std::string& n : names) n = dir + "/" + n;          // main.cpp:347-349
    return names;
}

int HeadlessApp::LoadObjects(const std::vector<std::string>& paths, int buildThreads) {
    scene_ = LoadSceneCached(paths, buildThreads, sceneCacheDir);
    int rc = rt_scene_upload(ctx_, scene_.tri12.data(), scene_.matIdx.data(), scene_.TriangleCount(), scene_.node8.data(), scene_.NodeCount());
    if (rc != RT_OK) return rc;
    std::vector<rt_material> mats;
    for (size_t i = 0; i < scene_.materials.size() && i < 16; ++i) {
        const Material& m = scene_.materials[i];
        rt_material o;
        o.colour[0] = m.colour.x; o.colour[1] = m.colour.y; o.colour[2] = m.colour.z;
        o.specularColour[0] = m.specularColour.x; o.specularColour[1] = m.specularColour.y; o.specularColour[2] = m.specularColour.z;
        o.roughness = m.roughness; o.metallic = m.metallic; o.transparency = m.transparency; o.refractionIndex = m.refractionIndex;
        o.isLight = m.isLight != 0.0f;
        mats.push_back(o);
    }
    return rt_set_materials(ctx_, mats.data(), (uint32_t)mats.size());
}

int HeadlessApp::UploadCamera() {                            // src/Camera.cpp:129-135
    rt_camera c;
    c.position[0] = camera_.Position().x; c.position[1] = camera_.Position().y; c.position[2] = camera_.Position().z;
    c.facing[0] = camera_.Facing().x; c.facing[1] = camera_.Facing().y; c.facing[2] = camera_.Facing().z;
    c.fov = camera_.Fov(); c.viewportWidth = camera_.ViewportWidth(); c.viewportDistance = camera_.ViewportDistance();
    return rt_set_camera(ctx_, &c);
}

void HeadlessApp::OnKey(int key, int action) {
    KeyEvent e{key, action};
    camera_.OnEvent(e);
    if (action == kPress) {                                  // src/BounceLimitManager.cpp:14-48
        if (auto bl = BounceLimitFromKey(key)) { bounceLimit_ = *bl; rt_set_bounce_limit(ctx_, bounceLimit_); frameIndex_ = 0; }
    }
    recorder_.OnEvent(e);
}

int HeadlessApp::Tick() {
    recorder_.Tick(*this);
    if (camera_.Tick()) { int rc = UploadCamera(); if (rc != RT_OK) return rc; frameIndex_ = 0; }   // src/Scene.cpp:27, Camera.cpp:122-126
    rand_.NextSeedTriple(lastSeed_);                         // src/Scene.cpp:31
    int rc = rt_render(ctx_, frameIndex_++, 1, lastSeed_);   // src/Scene.cpp:30 + main.cpp:232-233
    if (rc != RT_OK) return rc;
    displayValid_ = false;
    if (bloom && postprocessEveryFrame) { rc = rt_postprocess(ctx_, 1, nullptr, nullptr); displayValid_ = rc == RT_OK; }
    return rc;
}

int HeadlessApp::ReadDisplay(float* rgba, uint8_t* rgb8) {
    // With per-frame post-processing the chain has already run for the last frame; running it again
    // would advance the carried-over tmp image, so a second run is only made when none happened.
    if (bloom && displayValid_) return rt_display_read(ctx_, rgba, rgb8);
    int rc = rt_postprocess(ctx_, bloom ? 1 : 0, rgba, rgb8);
    displayValid_ = bloom && rc == RT_OK;
    return rc;
}

}  // namespace rta
