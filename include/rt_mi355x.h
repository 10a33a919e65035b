/*
 * rt_mi355x.h — C ABI of the MI355X path-tracing core (librt_mi355x.so).
 *
 * The reference renderer has no plugin or FFI layer: its tracer sits behind OpenGL, i.e. behind
 * a fixed set of named uniforms, buffers and one full-screen draw per frame. Each entry point
 * below replaces one of those GL-side interfaces (cited per function as file:line of the
 * reference). A host program that today fills those uniforms/buffers and calls glDrawArrays can
 * call these functions instead and read the same image back.
 *
 * Conventions
 *   - every function returns 0 on success, a negative rt_status otherwise; rt_last_error()
 *     gives the message. Nothing aborts, nothing throws across this boundary.
 *   - pointers are caller-owned host memory unless the name says "device"; they are only read
 *     (or written) during the call.
 *   - one caller thread per context (the reference is single-threaded around one GL context).
 *   - state is sticky between frames, like GL uniforms.
 *   - work is queued on the context's HIP stream; rt_sync / rt_accum_read / rt_postprocess wait.
 *   - images are rows of RGBA float (or RGB byte) pixels, row 0 = BOTTOM row, x to the right,
 *     exactly what glGetTexImage returns for the reference's textures.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / inconsistent scene arrays */
    RT_ERR_DEVICE = -2,      /* HIP runtime error (no GPU, out of memory, launch failure) */
    RT_ERR_STATE = -3,       /* call made before the data it needs was uploaded */
    RT_ERR_UNSUPPORTED = -4  /* scene exceeds a documented limit */
};

/* One entry of the shader's `uniform Material u_Materials[16]` (src/shaders/Fragment.glsl:15-23,90),
 * filled by Scene::SendSceneMaterials (src/Scene.cpp:133-153). */
typedef struct rt_material {
    float colour[3];
    float specularColour[3];
    float roughness;
    float metallic;
    float transparency;      /* luminance for light sources (Include/Materials.h:46-48) */
    float refractionIndex;
    int32_t isLight;
} rt_material;

/* `uniform Camera u_Camera` (src/shaders/Fragment.glsl:57-63,74); set by Scene::Scene
 * (src/Scene.cpp:20-22) and Camera::UploadInfo (src/Camera.cpp:129-135). */
typedef struct rt_camera {
    float position[3];
    float facing[3];
    float fov;               /* radians; carried for completeness, the tracer does not read it */
    float viewportWidth;
    float viewportDistance;
} rt_camera;

/* Per-context counters accumulated by rt_render since the last reset. */
typedef struct rt_counters {
    uint64_t rays;           /* BVH traversals (primary + bounce segments): Fragment.glsl:385 */
    uint64_t node_pops;      /* sum of the shader's iterationsCount */
    uint64_t boxes_tested;   /* distinct bounding boxes tested */
    uint64_t tris_tested;    /* triangle intersection tests */
    uint64_t noise_fetches;  /* texture(u_RgbNoise, ...) calls */
    uint64_t sky_fetches;    /* texture(u_Skybox, ...) calls */
    uint64_t paths;          /* pixel samples finished */
    uint64_t hits;           /* rays that hit a triangle (full counting only) */
    /* wave scheduler statistics (full counting only), indexed 0 path starts / 1 interior / 2 leaf / 3 shade:
     * rounds a wave spent in that code and how many of its 64 lanes took part. Path starts run inside the shading
     * round (after shading and the hand-out of pixels), so rounds[0] == rounds[3] and cycles[0] is 0. */
    uint64_t sched_rounds[4];
    uint64_t sched_lanes[4];
    uint64_t sched_cycles[4];   /* shader clock cycles the waves spent in those rounds (in units of 64 cycles) */
} rt_counters;

/* Library-level message for failures that happen before a context exists (rt_create). */
const char *rt_last_global_error(void);
const char *rt_last_error(const rt_context *ctx);

/* Creates a renderer for a width x height image on HIP device `device`.
 * Replaces: window/FBO/accumulation-texture setup and `uniform vec2 screenResolution`
 * (src/main.cpp:28-29,38-39,144-158). The accumulation image starts at zero. */
int rt_create(int device, uint32_t width, uint32_t height, rt_context **out);
void rt_destroy(rt_context *ctx);

/* Uploads the flattened scene. Replaces the three buffer uploads of Scene::LoadObjects
 * (src/Scene.cpp:120-126):
 *   tri12   float[12*nTri]  xyz0 per corner           (SSBO binding 0, Fragment.glsl:44-47)
 *   matIdx  int32[nTri]     material slot per triangle (u_MaterialsIndex, Fragment.glsl:85)
 *   node8   float[8*nNode]  max.xyz, min.xyz, float(triCount), float(triStart | rightChild),
 *                           nodes in pre-order, first child = parent + 1
 *                           (u_BoundingBoxes / u_BoundingBoxesCount, Fragment.glsl:87-88,195-205)
 * The arrays are validated (indices in range, tree well-formed, interior nodes at most 63 levels
 * deep with the root as level 1 -- what the shader's stack[64] can hold --, leaves <= 16
 * triangles, nTri < 2^24) and re-laid-out for the GPU; nNode == 0 gives an empty scene. */
int rt_scene_upload(rt_context *ctx, const float *tri12, const int32_t *matIdx, uint32_t nTri,
                    const float *node8, uint32_t nNode);

/* Replaces Scene::SendSceneMaterials (src/Scene.cpp:133-153). count <= 16. */
int rt_set_materials(rt_context *ctx, const rt_material *materials, uint32_t count);

/* Replaces Camera::UploadInfo + the constructor-time camera uniforms. */
int rt_set_camera(rt_context *ctx, const rt_camera *camera);

/* Replaces glUniform1ui(u_BounceLimit) (src/Scene.cpp:19, src/BounceLimitManager.cpp:14-48).
 * 0 = BVH heat map, 1 = |normal| of the primary hit, n >= 2 = path tracing. */
int rt_set_bounce_limit(rt_context *ctx, uint32_t bounceLimit);

/* Replaces the shader's compile-time constant RAY_COUNT (src/shaders/Fragment.glsl:5, used by the loop of
 * Fragment.glsl:459-483): rays traced per pixel and frame. Ray i takes its pixel jitter at u_RandSeed.yz + 0.007 i and
 * adds Hash(i) to both per-bounce seeds; from the second ray on the loop skips up to three rays when a ray's colour is
 * close to the sum of those before it, and the frame's colour is the mean of the rays traced. The reference is checked
 * in with 1, which is also the default here; values above 1 reproduce the shader recompiled with that constant.
 * 1 <= rays <= RT_MAX_RAY_COUNT. Takes effect with the next rt_render; the caller restarts its accumulation
 * (frame index 0) as it would after any change of the shader. rt_primary_ids always describes ray 0. */
#define RT_MAX_RAY_COUNT 64u
int rt_set_ray_count(rt_context *ctx, uint32_t rays);

/* Replaces LoadNoiseTexture (src/main.cpp:89-114): RGB8, rows bottom-up as uploaded to GL
 * (i.e. in file order, the reference does not flip), LINEAR filter, REPEAT wrap. */
int rt_set_noise(rt_context *ctx, const uint8_t *rgb, uint32_t width, uint32_t height);

/* Replaces LoadSkybox (src/main.cpp:45-87): six RGBA8 size x size faces in GL order
 * +X,-X,+Y,-Y,+Z,-Z (= right,left,top,bottom,front,back), LINEAR, CLAMP_TO_EDGE. */
int rt_set_skybox(rt_context *ctx, const uint8_t *const faces[6], uint32_t size);

/* Accumulation image (`u_Accumulation`, also the render target: src/main.cpp:148-158). */
int rt_accum_clear(rt_context *ctx);
int rt_accum_write(rt_context *ctx, const float *rgba);
int rt_accum_read(rt_context *ctx, float *rgba);

/* Renders nFrames progressive frames. Frame k uses u_FrameIndex = firstFrameIndex + k and
 * u_RandSeed = seeds3[3k..3k+2]; replaces Scene::Tick's two uniforms plus the tracer draw
 * (src/Scene.cpp:30-31, src/main.cpp:232-233). Seeds are always explicit. */
int rt_render(rt_context *ctx, uint32_t firstFrameIndex, uint32_t nFrames, const float *seeds3);
int rt_sync(rt_context *ctx);

/* Parity channel: traces only the primary rays of one frame and returns, per pixel,
 * {triangle index or -1, bit pattern of t, iterationsCount}. No counterpart in the reference's
 * output (it emits colours only); it exposes the values Fragment.glsl:265-326 computes. */
int rt_primary_ids(rt_context *ctx, const float *seed3, int32_t *out3);

/* Parity channel: the tracer's ray/triangle test, hitTriangle (Fragment.glsl:207-240), on n independent pairs of one
 * triangle (tri12: three corners, xyz0 each, as in rt_scene_upload) and one ray (origin and direction, 3 floats each;
 * the direction is used as given). hitOut[i] = 1 and tOut[i] = the distance when ray i hits triangle i, else 0 and 0.
 * Runs the device function the render uses, on the triangle record the upload makes; needs no scene. Host arrays. */
int rt_probe_triangles(rt_context *ctx, uint32_t n, const float *tri12, const float *origins3, const float *directions3,
                       uint8_t *hitOut, float *tOut);

/* Parity channel: the plane coefficients with which the bloom chain evaluates a framebuffer blit (glBlitNamedFramebuffer,
 * src/main.cpp:244-248 and :261-265, as Mesa llvmpipe executes it: two triangles, one plane c0 + cx * x + cy * y per
 * triangle and texture coordinate, in fp32) from the source rectangle [0,srcW) x [0,srcH) of a texW x texH texture onto
 * [0,dstW) x [0,dstH) of a surfW x surfH render target. planes12 = {c0, cx, cy} of the horizontal coordinate for the
 * triangle below and above the rectangle's diagonal, then of the vertical coordinate. Host arithmetic only: needs no
 * context and no GPU. */
int rt_probe_blit_planes(uint32_t srcW, uint32_t srcH, uint32_t texW, uint32_t texH, uint32_t dstW, uint32_t dstH,
                         uint32_t surfW, uint32_t surfH, float *planes12);

/* Bloom chain + final composite of one frame (src/main.cpp:242-275 with
 * extractBrightFragment.glsl, bloomFragment.glsl, finalFragment.glsl); bloom_on == 0 is the
 * raw blit of src/main.cpp:235-240. Either output may be NULL. rgb8_out is what
 * Recorder::WriteTexture reads back (Include/Recorder.h:93-100). */
int rt_postprocess(rt_context *ctx, int bloom_on, float *rgba_out, uint8_t *rgb8_out);

/* The bloom chain + final composite as one call that allocates nothing. It has no counterpart in the reference; it
 * is for callers that cannot allocate device memory while they run: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. Same kernels and same results as
 *     rt_display_reset(ctx); rt_accum_write(ctx, accum); rt_postprocess(ctx, 1, rgba, rgb8);
 * but every image is DEVICE memory of the current HIP device: accumDevice = width*height RGBA floats, rows bottom-up
 * (what rt_accum_read returns); rgbaOutDevice (width*height*4 floats) and rgb8OutDevice (width*height*3 bytes)
 * receive the composite, either may be NULL. The work is queued on hipStream (a hipStream_t, NULL = the default
 * stream); the call does not wait for it. The chain's working images live in one fixed 20 MiB array in device
 * memory that every GPU the library runs on holds until the process ends: images up to 3840x2160 fit, larger ones
 * are refused with RT_ERR_UNSUPPORTED (message: rt_last_global_error). Nothing is carried over from call to call
 * (each call starts from a cleared tmp texture, like the first frame of a fresh process). All calls on one GPU share
 * that array: calls that overlap in time on one GPU race; order them on one stream or with events. The array's
 * address is looked up once per device, by rt_create or by the first call on that device; make that first call
 * (or any rt_create) outside graph capture. */
int rt_postprocess_static(const float *accumDevice, uint32_t width, uint32_t height, float *rgbaOutDevice,
                          uint8_t *rgb8OutDevice, void *hipStream);

/* Reads the image the last rt_postprocess(bloom_on = 1) left on display (the reference's tmp
 * texture) without running the chain again. */
int rt_display_read(rt_context *ctx, float *rgba_out, uint8_t *rgb8_out);
/* Clears what the post-processing chain carries over from earlier frames (the reference's tmp texture),
 * as in a freshly started process. */
int rt_display_reset(rt_context *ctx);

int rt_get_counters(rt_context *ctx, rt_counters *out, int reset);
/* full != 0: rt_render also tallies node pops, boxes, triangles and texture fetches (an
 * instrumented kernel, a few percent slower). full == 0 (default): only rays and paths. */
int rt_set_counting(rt_context *ctx, int full);

/* ---- multi-GPU tile sharding (no counterpart in the reference, which is single-GPU) ----
 * The image is cut into 8x8-pixel tiles, tile t = ty * tilesX + tx. With a shard (rank, world)
 * a context renders only tiles with t % world == rank and keeps them contiguous in device
 * memory, 64 RGBA float pixels per tile, local tile j = global tile j * world + rank. */
int rt_set_shard(rt_context *ctx, uint32_t rank, uint32_t world);
/* Number of tiles every rank's shard buffer holds (ceil(tiles / world); tail tiles are padding). */
int rt_shard_tiles(const rt_context *ctx, uint32_t *tilesPerRank);
/* Device pointer of this context's shard: tilesPerRank * 64 * 4 floats. */
int rt_shard_device_ptr(rt_context *ctx, void **devicePtr);
/* Queues a device-to-device copy of the shard into `dstDevice` (tilesPerRank * 64 * 4 floats) on the
 * context's stream, i.e. after the frames rendered so far and before those rendered next: the
 * send buffer of a gather that overlaps with further rendering. */
int rt_shard_export(rt_context *ctx, void *dstDevice);
/* On the gathering rank: `gathered` is a device buffer holding world shards back to back in
 * rank order (what a gather collective delivers). Scatters them into this context's image. */
int rt_accum_assemble(rt_context *ctx, const void *gatheredDevice, uint32_t world);

/* ---- several GPUs in one process: the tile shards of one image, gathered over RCCL (no counterpart in the reference) ----
 * rt_group_create makes one sharded context per listed device (rank r renders tiles t % n == r on devices[r]), an
 * unsharded root context on devices[0] that receives the assembled image, and an RCCL communicator over the devices
 * (RCCL is loaded at run time; a process that never creates a group does not need it). Scene, materials, textures,
 * camera and bounce limit are uploaded per rank through rt_group_rank(g, r) with the calls above (every rank holds the
 * whole scene). rt_group_render queues the frames on every rank. rt_group_gather sends every rank's shard to rank 0
 * (ncclSend / ncclRecv in one group call: each peer pushes over its own xGMI link) and scatters them into the root
 * context's image, all queued on the devices' streams; read the image with rt_accum_read / rt_postprocess on
 * rt_group_root(g). The assembled image equals an unsharded render of the same frames bit for bit. */
typedef struct rt_group rt_group;
int rt_group_create(const int *devices, uint32_t nDevices, uint32_t width, uint32_t height, rt_group **out);
void rt_group_destroy(rt_group *group);
uint32_t rt_group_size(const rt_group *group);
rt_context *rt_group_rank(rt_group *group, uint32_t rank);
rt_context *rt_group_root(rt_group *group);
int rt_group_render(rt_group *group, uint32_t firstFrameIndex, uint32_t nFrames, const float *seeds3);
int rt_group_gather(rt_group *group);
int rt_group_sync(rt_group *group);
const char *rt_group_last_error(const rt_group *group);

/* Queue this context's work on a caller-provided hipStream_t (NULL = the context's own). */
int rt_set_stream(rt_context *ctx, void *hipStream);

/* Timing of the tracer kernel launches made by rt_render since the last reset, measured with
 * HIP events on the context's stream: total milliseconds and number of launches. */
int rt_kernel_time(rt_context *ctx, double *ms_total, uint64_t *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
