// Offline video: the playback half of the reference's Recorder (Include/Recorder.h:36-55,93-103) as a batch job.
//
// The reference, during playback, shows one recorded camera pose for 5000 ms of wall clock (however many progressive
// frames the GPU manages in that time), writes the displayed post-bloom image as shots/raytrace-out-<n>.png and moves
// on to the next pose. Here a pose gets a fixed number of samples per pixel instead, rendered in ONE launch (a pixel's
// running mean stays in registers), followed by one run of the bloom chain; poses are independent of each other, so
// they are dealt round-robin over ranks (GPUs) and every rank writes the images of its own poses. PNG encoding and
// file writing happen on writer threads while the GPU traces the next pose, and poses whose image already exists
// are skipped, so an interrupted job is resumed by running it again.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "app.h"

namespace rta {

struct VideoJob {
    std::vector<Pose> poses;            // pose k -> <outputDir>/raytrace-out-<k>.png
    uint32_t samplesPerPixel = 256;
    uint32_t rank = 0, world = 1;       // this process renders poses k with k % world == rank
    std::string outputDir = "shots";
    bool bloom = true;                  // the reference's chain is always on
    bool flipVertically = false;        // the reference writes rows bottom-up (its PNGs are upside down)
    bool resume = true;                 // skip poses whose image file exists
    bool writeImages = true;            // false: render and read back only (timing runs)
    int compression = 1;                // zlib level of the PNG data, 0 = stored
    uint32_t writerThreads = 2;
};

struct VideoStats {
    uint32_t rendered = 0, skipped = 0, written = 0, writeFailures = 0;
    double wallSeconds = 0;             // whole job on this rank, last file closed
    double traceSeconds = 0;            // tracer kernel time (HIP events)
    double encodeSeconds = 0;           // writer threads busy (summed over threads; overlaps with tracing)
    double stallSeconds = 0;            // render loop blocked because the writers were behind
    uint64_t rays = 0;
};

// The u_RandSeed triples pose k uses: frames k*spp .. (k+1)*spp-1 of the reference's unseeded rand() stream
// (src/Scene.cpp:31), so that an image does not depend on how many ranks share the job or on which poses were skipped.
std::vector<float> VideoSeeds(uint32_t firstPose, uint32_t nPoses, uint32_t samplesPerPixel);

std::string VideoImagePath(const std::string& outputDir, uint32_t pose);

// Renders this rank's poses on `ctx` (scene, textures and bounce limit already set; width x height as created).
// Returns an rt_status; on failure the images finished so far stay on disk.
int RenderVideo(rt_context* ctx, uint32_t width, uint32_t height, const VideoJob& job, VideoStats* stats);

bool ReadPoseFile(const std::string& path, std::vector<Pose>& out);

}  // namespace rta
