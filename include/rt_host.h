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

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
