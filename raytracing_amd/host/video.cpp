#include "video.h"

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <fstream>
#include <mutex>
#include <thread>

namespace rta {

namespace {

using Clock = std::chrono::steady_clock;
double Since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// Finished frames on their way to disk. Bounded: a render loop that gets more than `capacity` frames ahead of the
// writers waits (and the wait is reported), so host memory stays bounded however slow the disk is.
class FrameWriters {
public:
    FrameWriters(uint32_t threads, size_t capacity, uint32_t width, uint32_t height, bool flip, int compression)
        : capacity_(capacity), width_(width), height_(height), flip_(flip), compression_(compression) {
        for (uint32_t i = 0; i < threads; ++i) pool_.emplace_back([this] { Run(); });
    }
    ~FrameWriters() { Finish(); }

    // Returns the seconds spent waiting for room in the queue.
    double Push(std::string path, std::vector<uint8_t>&& rgb8) {
        const auto t0 = Clock::now();
        std::unique_lock<std::mutex> lock(m_);
        room_.wait(lock, [this] { return queue_.size() < capacity_; });
        const double waited = Since(t0);
        queue_.push_back(Item{std::move(path), std::move(rgb8)});
        work_.notify_one();
        return waited;
    }
    void Finish() {
        { std::lock_guard<std::mutex> lock(m_); closing_ = true; }
        work_.notify_all();
        for (std::thread& t : pool_) if (t.joinable()) t.join();
        pool_.clear();
    }
    uint32_t written = 0, failures = 0;     // valid after Finish()
    double busySeconds = 0;

private:
    struct Item { std::string path; std::vector<uint8_t> rgb8; };
    void Run() {
        for (;;) {
            Item item;
            {
                std::unique_lock<std::mutex> lock(m_);
                work_.wait(lock, [this] { return closing_ || !queue_.empty(); });
                if (queue_.empty()) return;
                item = std::move(queue_.front());
                queue_.pop_front();
                room_.notify_one();
            }
            const auto t0 = Clock::now();
            // written under a temporary name and renamed: a file that exists is a file that is complete, which is what
            // resuming relies on
            const std::string tmp = item.path + ".part" + std::to_string((long)::getpid());
            bool ok = WritePngRgb8(tmp, width_, height_, item.rgb8.data(), flip_, compression_) && std::rename(tmp.c_str(), item.path.c_str()) == 0;
            if (!ok) std::remove(tmp.c_str());
            const double dt = Since(t0);
            std::lock_guard<std::mutex> lock(m_);
            busySeconds += dt;
            if (ok) ++written; else ++failures;
        }
    }
    std::mutex m_;
    std::condition_variable work_, room_;
    std::deque<Item> queue_;
    std::vector<std::thread> pool_;
    size_t capacity_;
    uint32_t width_, height_;
    bool flip_, closing_ = false;
    int compression_;
};

bool FileExists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0; }

}  // namespace

std::vector<float> VideoSeeds(uint32_t firstPose, uint32_t nPoses, uint32_t spp) {
    GlibcRand rng;
    float skip[3];
    for (uint64_t i = 0; i < (uint64_t)firstPose * spp; ++i) rng.NextSeedTriple(skip);
    std::vector<float> seeds((size_t)nPoses * spp * 3);
    for (size_t i = 0; i < (size_t)nPoses * spp; ++i) rng.NextSeedTriple(&seeds[3 * i]);
    return seeds;
}

std::string VideoImagePath(const std::string& outputDir, uint32_t pose) {
    return outputDir + "/raytrace-out-" + std::to_string(pose) + ".png";     // Include/Recorder.h:94-96
}

bool ReadPoseFile(const std::string& path, std::vector<Pose>& out) {
    std::ifstream in(path);
    if (!in) return false;
    Pose p;
    while (in >> p.position.x >> p.position.y >> p.position.z >> p.facing.x >> p.facing.y >> p.facing.z) out.push_back(p);
    return true;
}

int RenderVideo(rt_context* ctx, uint32_t width, uint32_t height, const VideoJob& job, VideoStats* stats) {
    VideoStats st;
    if (!ctx || job.world == 0 || job.rank >= job.world || job.samplesPerPixel == 0) return RT_ERR_INVALID;
    const auto t0 = Clock::now();
    if (job.writeImages) ::mkdir(job.outputDir.c_str(), 0777);          // the reference never creates the directory
    rt_counters before{}, after{};
    double ms0 = 0, ms1 = 0;
    int rc = rt_get_counters(ctx, &before, 0);
    if (rc == RT_OK) rc = rt_kernel_time(ctx, &ms0, nullptr, 0);
    if (rc != RT_OK) return rc;

    const uint32_t nPoses = (uint32_t)job.poses.size(), spp = job.samplesPerPixel;
    // one walk over the seed stream: every pose's slice is cut from it (a rank renders every world-th pose)
    std::vector<float> seeds;
    if (nPoses) seeds = VideoSeeds(0, nPoses, spp);
    {
        FrameWriters writers(job.writeImages ? std::max(1u, job.writerThreads) : 0u, 2 * (size_t)std::max(1u, job.writerThreads) + 1, width, height,
                             job.flipVertically, job.compression);
        Camera camera;
        for (uint32_t k = job.rank; k < nPoses && rc == RT_OK; k += job.world) {
            const std::string path = VideoImagePath(job.outputDir, k);
            if (job.writeImages && job.resume && FileExists(path)) { ++st.skipped; continue; }
            // Recorder::Tick's pose switch, Include/Recorder.h:50-53: position, normalised facing, upload, frame index 0
            camera.SetPosition(job.poses[k].position);
            camera.SetFacing(job.poses[k].facing);
            rt_camera c;
            c.position[0] = camera.Position().x; c.position[1] = camera.Position().y; c.position[2] = camera.Position().z;
            c.facing[0] = camera.Facing().x; c.facing[1] = camera.Facing().y; c.facing[2] = camera.Facing().z;
            c.fov = camera.Fov(); c.viewportWidth = camera.ViewportWidth(); c.viewportDistance = camera.ViewportDistance();
            rc = rt_set_camera(ctx, &c);
            if (rc == RT_OK) rc = rt_accum_clear(ctx);       // frame index 0 gives the first frame weight 1; cleared for definiteness
            if (rc == RT_OK) rc = rt_display_reset(ctx);     // every image starts from a cleared tmp texture: images do not depend on their neighbours
            if (rc == RT_OK) rc = rt_render(ctx, 0, spp, &seeds[(size_t)k * spp * 3]);
            std::vector<uint8_t> rgb8((size_t)width * height * 3);
            if (rc == RT_OK) rc = rt_postprocess(ctx, job.bloom ? 1 : 0, nullptr, rgb8.data());   // waits for the pose; Recorder::WriteTexture's read-back
            if (rc != RT_OK) break;
            ++st.rendered;
            if (job.writeImages) st.stallSeconds += writers.Push(path, std::move(rgb8));
        }
        writers.Finish();
        st.written = writers.written; st.writeFailures = writers.failures; st.encodeSeconds = writers.busySeconds;
    }
    if (rc == RT_OK) rc = rt_get_counters(ctx, &after, 0);
    if (rc == RT_OK) rc = rt_kernel_time(ctx, &ms1, nullptr, 0);
    st.rays = after.rays - before.rays;
    st.traceSeconds = (ms1 - ms0) * 1e-3;
    st.wallSeconds = Since(t0);
    if (stats) *stats = st;
    if (rc == RT_OK && st.writeFailures) return RT_ERR_INVALID;
    return rc;
}

}  // namespace rta
