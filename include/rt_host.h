/*
 * rt_host.h — C ABI of the host-side scene model (part of librt_mi355x.so, CPU only).
 *
 * These functions produce the flat arrays that rt_scene_upload / rt_set_materials consume, from
 * the reference's own file formats, with the reference's BVH build order. They exist so that
 * programs in any language (the tests use Python/ctypes) can drive the whole path
 *   object files -> triangles -> BVH -> flat arrays -> GPU
 * the way the reference's Scene::LoadObjects does (src/Scene.cpp:87-131).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#include "rt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

typedef struct rth_bvh_stats {      /* what BvhTree::BuildTree prints (src/BvhTree.cpp:143-148) */
    int32_t max_depth;
    uint32_t leaves;
    uint64_t splits;
    uint64_t degenerate_splits;
    uint64_t leaf_depth_sum;
    double build_ms;
} rth_bvh_stats;

/* Loads and merges the listed object files (ObjectLoader / OFFLoader formats), builds the BVH
 * and flattens. Replaces Scene::LoadObjects up to, not including, the GPU uploads. */
int rth_scene_load(const char *const *paths, uint32_t nPaths, int buildThreads, rth_scene **out);

/* The same through a binary cache (no counterpart in the reference, which parses and builds on every start:
 * src/Scene.cpp:87-127): the flat arrays are stored in cacheDir under the SHA-256 of a format version and of the
 * input files' bytes in order, and loaded from there when that key is found again; a blob that is damaged, of another
 * version or made from other bytes is ignored and rewritten. cacheDir NULL or "" = rth_scene_load.
 * *cacheHit (may be NULL) receives 1 when the scene came from the cache, else 0. */
int rth_scene_load_cached(const char *const *paths, uint32_t nPaths, int buildThreads, const char *cacheDir,
                          rth_scene **out, int *cacheHit);

/* Same from memory: tri9 = 9 floats per triangle (three corners), matSlot = material slot per
 * triangle, materials = the material table. maxLeaf <= 0 selects the reference's default of 3. */
int rth_scene_build(const float *tri9, const int32_t *matSlot, uint32_t nTri, const rt_material *materials,
                    uint32_t nMaterials, int buildThreads, int maxLeaf, rth_scene **out);

void rth_scene_free(rth_scene *scene);

uint32_t rth_scene_triangle_count(const rth_scene *scene);
uint32_t rth_scene_node_count(const rth_scene *scene);
uint32_t rth_scene_material_count(const rth_scene *scene);
const float *rth_scene_tri12(const rth_scene *scene);          /* Scene::FlattenTrianglesVertices */
const int32_t *rth_scene_matidx(const rth_scene *scene);       /* Scene::FlattenTrianglesMatIdx */
const float *rth_scene_node8(const rth_scene *scene);          /* Scene::FlattenBoundingBoxes */
const rt_material *rth_scene_materials(const rth_scene *scene);
void rth_scene_stats(const rth_scene *scene, rth_bvh_stats *out);
uint32_t rth_scene_warning_count(const rth_scene *scene);
const char *rth_scene_warning(const rth_scene *scene, uint32_t i);

/* Uploads a scene and its material table to a renderer: rt_scene_upload + rt_set_materials. */
int rth_scene_send(const rth_scene *scene, rt_context *ctx);

/* ---- controls (CPU only): the reference's camera and key handling without a window ---- */
typedef struct rth_camera rth_camera;
rth_camera *rth_camera_create(void);                              /* defaults of Include/Camera.h:7-11 */
void rth_camera_destroy(rth_camera *cam);
void rth_camera_key(rth_camera *cam, int glfwKey, int action);    /* Camera::OnEvent, src/Camera.cpp:137-143 */
int rth_camera_tick(rth_camera *cam);                             /* Camera::Tick; 1 when the camera changed */
/* out[9] = position xyz, facing xyz, fov (rad), viewportWidth, viewportDistance */
void rth_camera_get(const rth_camera *cam, float *out9);
void rth_camera_set_pose(rth_camera *cam, const float *position3, const float *facing3);   /* SetPosition + SetFacing */
/* Bounce limit a key press selects (src/BounceLimitManager.cpp:14-48), or -1 for other keys. */
int rth_bounce_limit_from_key(int glfwKey);
/* u_RandSeed triples of the first nFrames frames of a fresh process (src/Scene.cpp:31). */
void rth_rand_seed_triples(uint32_t nFrames, float *out3);
/* Object paths a RayTracer.ini names, newline separated, into buf; returns their number or -1.
 * skyboxDir (may be NULL) receives [SkyBox] Path. */
int rth_ini_object_paths(const char *iniPath, char *buf, uint32_t bufSize, char *skyboxDir, uint32_t skyboxDirSize);
/* The values a RayTracer.ini holds under [section] name, as the reference's ConfigParser::aConfigVec<std::string>
 * returns them (src/ConfigParser/ConfigParser.cpp:31-86, Include/ConfigParser/ConfigParser.hpp:68-87): each value
 * followed by '\n' in buf; returns their number (0 for a key the file does not define), or -1 when the file does not
 * parse, with the reference's message ("<path>:<line>: parsing error \n<line>") in errBuf (may be NULL). */
int rth_ini_values(const char *iniPath, const char *section, const char *name, char *buf, uint32_t bufSize, char *errBuf, uint32_t errSize);
/* One value converted like ConfigParser::aConfig<T> (ConfigParser.hpp:52-66, ConfigParser.cpp:88-108): kind 0 = int,
 * 1 = float, 2 = bool; stream extraction, hexadecimal when the text contains "0x". A missing key or position gives 0
 * (the reference indexes without a check there). */
int rth_ini_number(const char *iniPath, const char *section, const char *name, uint32_t pos, int kind, double *out);
int rth_write_png_rgb8(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb, int flipVertically);
/* PNG decode as the reference's stbi_load(path, .., channels) (src/main.cpp:64,96): 8-bit pixels, rows in file
 * order, converted to 3 or 4 channels. Call with pixels == NULL to get the size. Returns 0 or RT_ERR_INVALID. */
int rth_read_png(const char *path, int channels, uint32_t *width, uint32_t *height, uint8_t *pixels, uint64_t capacity);

/* ---- headless frame loop: src/main.cpp:224-279 without a window ---- */
typedef struct rth_app rth_app;
typedef struct rth_app_state {
    float camera[9];             /* as rth_camera_get */
    float last_seed[3];          /* u_RandSeed of the last tick */
    uint32_t next_frame_index;   /* u_FrameIndex the next tick will use */
    uint32_t bounce_limit;
    int32_t recorder_state;      /* 0 nothing, 1 recording, 2 playback */
    uint32_t poses_left;
    uint32_t images_written;
} rth_app_state;
int rth_app_create(rt_context *ctx, uint32_t width, uint32_t height, rth_app **out);
void rth_app_destroy(rth_app *app);
int rth_app_load_objects(rth_app *app, const char *const *paths, uint32_t nPaths, int buildThreads);
/* Directory for rth_scene_load_cached's blobs, used by the app's later load_objects calls (NULL or "" = no cache). */
void rth_app_set_scene_cache(rth_app *app, const char *cacheDir);
void rth_app_key(rth_app *app, int glfwKey, int action);          /* key_callback, src/main.cpp:289-296 */
int rth_app_tick(rth_app *app);                                   /* one loop iteration */
int rth_app_read_display(rth_app *app, float *rgba, uint8_t *rgb8);
void rth_app_get_state(const rth_app *app, rth_app_state *out);
/* bloom: run the bloom chain (the reference always does); postEveryFrame: run it on every tick as
 * the reference does (0 = only when an image is read); framesPerImage / recordEveryFrames replace the
 * Recorder's 5000 ms / 10 ms intervals; outputDir receives raytrace-out-<n>.png. */
void rth_app_configure(rth_app *app, int bloom, int postEveryFrame, uint32_t framesPerImage, uint32_t recordEveryFrames,
                       const char *outputDir, int flipVertically);
void rth_app_push_pose(rth_app *app, const float *position3, const float *facing3);
int rth_app_save_poses(const rth_app *app, const char *path);
int rth_app_load_poses(rth_app *app, const char *path);

/* ---- offline video: the playback half of Include/Recorder.h:36-55,93-103 as a batch job ----
 * Pose k (poses6 = px py pz fx fy fz per pose) gets `samplesPerPixel` progressive frames in one launch, seeded with
 * frames k*spp .. (k+1)*spp-1 of the reference's rand() stream, then one run of the bloom chain, and is written as
 * <outputDir>/raytrace-out-<k>.png (Recorder::WriteTexture's name and row order). This process renders the poses with
 * k % world == rank on `ctx` (scene, textures, bounce limit already set); encoding and writing run on writer threads
 * while the next pose is traced; with RTH_VIDEO_RESUME poses whose file exists are skipped. */
typedef struct rth_video_stats {
    uint32_t rendered, skipped, written, write_failures;
    double wall_seconds;      /* the whole call */
    double trace_seconds;     /* tracer kernel time, HIP events */
    double encode_seconds;    /* writer threads busy, summed (overlaps with tracing) */
    double stall_seconds;     /* render loop blocked because the writers were behind */
    uint64_t rays;
} rth_video_stats;
enum { RTH_VIDEO_BLOOM = 1, RTH_VIDEO_FLIP = 2, RTH_VIDEO_RESUME = 4, RTH_VIDEO_WRITE = 8 };
int rth_video_render(rt_context *ctx, uint32_t width, uint32_t height, const float *poses6, uint32_t nPoses,
                     uint32_t samplesPerPixel, uint32_t rank, uint32_t world, const char *outputDir, uint32_t flags,
                     int compression, uint32_t writerThreads, rth_video_stats *out);
/* The u_RandSeed triples of poses firstPose .. firstPose+nPoses-1: out3 = float[nPoses * samplesPerPixel * 3]. */
void rth_video_seeds(uint32_t firstPose, uint32_t nPoses, uint32_t samplesPerPixel, float *out3);
/* Reads a pose file (one pose per line: px py pz fx fy fz, what rth_app_save_poses writes). *nPoses receives the
 * number of poses in the file; at most `capacity` of them are stored in poses6 (which may be NULL to count). */
int rth_read_poses(const char *path, float *poses6, uint32_t capacity, uint32_t *nPoses);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
