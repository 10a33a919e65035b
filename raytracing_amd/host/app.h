// Headless re-statement of the reference's frame loop (src/main.cpp:224-279) on top of the C ABI:
// key events in, frames out. There is no window: events come from a caller (script, test, UI) and
// "present" is replaced by reading images back.
//
//   per Tick():  recorder.Tick()                      main.cpp:228, Include/Recorder.h:36-55
//                scene tick: camera.Tick(); if the camera moved -> upload it, frame index = 0;
//                            u_FrameIndex = frameIndex++; u_RandSeed = three rand() values
//                                                      main.cpp:230, src/Scene.cpp:25-32
//                tracer pass                           main.cpp:232-233
//                bloom chain + final composite         main.cpp:241-276 (optional per frame here)
#pragma once

#include <cstdint>
#include <deque>
#include <string>
#include <vector>

#include "controls.h"
#include "rt_mi355x.h"
#include "scene_data.h"

namespace rta {

struct Pose { Vec3 position, facing; };

class HeadlessApp;

// Include/Recorder.h re-stated with frame counts in place of wall-clock intervals: the reference
// samples the camera every 10 ms while recording and, during playback, writes the displayed image
// and moves to the next recorded pose every 5000 ms (however many frames that happened to be).
class Recorder {
public:
    enum State { kNothing, kRecording, kPlayback };

    void OnEvent(const KeyEvent& e);                    // R toggles recording, P starts playback
    void Tick(HeadlessApp& app);
    State GetState() const { return state_; }
    size_t PosesLeft() const { return poses_.size(); }
    uint32_t ImagesWritten() const { return imageCounter_; }

    // Pose files: one pose per line, "px py pz fx fy fz" (the reference keeps poses in memory only).
    bool SavePoses(const std::string& path) const;
    bool LoadPoses(const std::string& path);
    void PushPose(const Pose& p) { poses_.push_back(p); }

    uint32_t framesPerImage = 256;      // playback: samples per pixel accumulated before an image is written
    uint32_t recordEveryFrames = 1;     // recording: one pose every this many frames
    std::string outputDir = "shots";    // images go to <outputDir>/raytrace-out-<n>.png
    bool flipVertically = false;        // the reference writes rows bottom-up (its PNGs are upside down)

private:
    State state_ = kNothing;
    std::deque<Pose> poses_;
    uint32_t imageCounter_ = 0, sinceSwitch_ = 0, sinceRecord_ = 0;
};

class HeadlessApp {
public:
    // The context must outlive the app; width/height are those it was created with.
    HeadlessApp(rt_context* ctx, uint32_t width, uint32_t height);

    // RayTracer.ini keys: [SkyBox] Path, [Objects] DirectoryPath + Filenames (RayTracer.ini:1-13);
    // the shader paths are read and ignored. Textures are not loaded here (callers pass pixels to
    // rt_set_noise / rt_set_skybox); returns the object paths the ini names.
    static std::vector<std::string> ObjectPathsFromIni(const std::string& iniPath, std::string* skyboxDir = nullptr);

    int LoadObjects(const std::vector<std::string>& paths, int buildThreads = 1);   // Scene::LoadObjects incl. upload
    const SceneData& Scene() const { return scene_; }

    void OnKey(int key, int action);        // what KeyEventNotifier::notify fans out (main.cpp:289-296)
    int Tick();                             // one iteration of the frame loop; returns an rt_status

    // The displayed image of the last Tick (post-processed when bloom is on).
    int ReadDisplay(float* rgba, uint8_t* rgb8);

    Camera& GetCamera() { return camera_; }
    Recorder& GetRecorder() { return recorder_; }
    uint32_t FrameIndex() const { return frameIndex_; }     // value the NEXT tick will use
    void ResetFrameIndex() { frameIndex_ = 0; }
    uint32_t BounceLimit() const { return bounceLimit_; }
    int UploadCamera();
    rt_context* Context() { return ctx_; }
    uint32_t Width() const { return width_; }
    uint32_t Height() const { return height_; }
    const float* LastSeed() const { return lastSeed_; }

    std::string sceneCacheDir;              // LoadObjects keeps parsed + built scenes here, keyed by the files' bytes (empty = off)
    bool bloom = true;                      // the reference's B key never takes effect (its key table is never written): bloom is always on
    bool postprocessEveryFrame = true;      // as the reference does; the tmp image then carries over frame to frame

private:
    rt_context* ctx_;
    uint32_t width_, height_;
    SceneData scene_;
    Camera camera_;
    Recorder recorder_;
    GlibcRand rand_;
    uint32_t frameIndex_ = 0;
    uint32_t bounceLimit_ = 4;              // src/Scene.cpp:19
    float lastSeed_[3] = {0, 0, 0};
    bool displayValid_ = false;
};

// PNG writer (8-bit RGB; `compression` is the zlib level of the image data, 0 = stored). Rows are written in the order
// given, or bottom row first with flipVertically.
bool WritePngRgb8(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgb, bool flipVertically, int compression = 0);

}  // namespace rta
