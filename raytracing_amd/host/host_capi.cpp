// C ABI over the host scene model (declared in include/rt_host.h).
#include "rt_host.h"

#include <string>
#include <vector>

#include "scene_data.h"

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
    std::vector<std::string> p;
    for (uint32_t i = 0; i < nPaths; ++i) p.emplace_back(paths[i] ? paths[i] : "");
    *out = Wrap(rta::LoadScene(p, buildThreads));
    return RT_OK;
}

int rth_scene_build(const float* tri9, const int32_t* matSlot, uint32_t nTri, const rt_material* materials,
                    uint32_t nMaterials, int buildThreads, int maxLeaf, rth_scene** out) {
    if (!out || (nTri && (!tri9 || !matSlot)) || (nMaterials && !materials)) return RT_ERR_INVALID;
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
    return RT_OK;
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
    int rc = rt_scene_upload(ctx, s->data.tri12.data(), s->data.matIdx.data(), s->data.TriangleCount(),
                             s->data.node8.data(), s->data.NodeCount());
    if (rc != RT_OK) return rc;
    uint32_t n = (uint32_t)s->mats.size();
    return rt_set_materials(ctx, s->mats.data(), n > 16 ? 16 : n);
}

}  // extern "C"
