// ref_host — C shim around the reference's own GL-free host code, compiled from the sources
// where they lie under /root/reference (never copied): src/BvhTree.cpp, src/ObjectLoader.cpp,
// Include/{BvhTree,ObjectLoader,Tri,Materials,Math3D}.h.
//
// TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libref_host.so by oracle/Makefile, in the
// build container. It is used to generate the BVH / loader golden fixtures and to check the
// product's builder against the real thing while the reference is present.
//
// Only the flattening (Scene::Flatten*, src/Scene.cpp:34-85) is re-stated here, because
// Scene.h drags in OpenGL headers; it is three trivial loops.
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "BvhTree.h"
#include "ObjectLoader.h"

namespace {
struct RefScene {
    std::vector<float> tri12;
    std::vector<int32_t> matIdx;
    std::vector<float> node8;
    std::vector<float> mats11;
    std::string log;
};

void Flatten(RefScene& s, const std::vector<BoundingBox>& boxes, const std::vector<Tri>& tris) {
    for (const Tri& t : tris) {
        const Vector3f* ps[3] = {&t.pos1, &t.pos2, &t.pos3};
        for (const Vector3f* p : ps) { s.tri12.push_back(p->x); s.tri12.push_back(p->y); s.tri12.push_back(p->z); s.tri12.push_back(0); }
        s.matIdx.push_back(t.materialsIndex);
    }
    for (const BoundingBox& b : boxes) {
        s.node8.push_back(b.maxi.x); s.node8.push_back(b.maxi.y); s.node8.push_back(b.maxi.z);
        s.node8.push_back(b.mini.x); s.node8.push_back(b.mini.y); s.node8.push_back(b.mini.z);
        s.node8.push_back(static_cast<float>(b.triangleCount));
        s.node8.push_back(static_cast<float>(b.triangleCount == 0 ? b.rightChildIndex : b.triangleStartIndex));
    }
}

void PushMaterial(RefScene& s, const Material::Material& m) {
    const float v[11] = {m.colour.x, m.colour.y, m.colour.z, m.specularColour.x, m.specularColour.y, m.specularColour.z,
                         m.roughness, m.metallic, m.transparency, m.refractionIndex, m.isLight ? 1.0f : 0.0f};
    s.mats11.insert(s.mats11.end(), v, v + 11);
}

// BvhTree::BuildTree prints to std::cout; capture it so callers can read the statistics.
struct CoutCapture {
    std::ostringstream os; std::streambuf* old;
    CoutCapture() : old(std::cout.rdbuf(os.rdbuf())) {}
    ~CoutCapture() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

void* ref_scene_build(const float* tri9, const int32_t* mat, uint32_t n) {
    RefScene* s = new RefScene();
    std::vector<Tri> tris;
    for (uint32_t i = 0; i < n; ++i) {
        const float* p = tri9 + (size_t)i * 9;
        tris.push_back(Tri(Vector3f(p[0], p[1], p[2]), Vector3f(p[3], p[4], p[5]), Vector3f(p[6], p[7], p[8]), mat[i]));
    }
    CoutCapture cap;
    BvhTree tree(tris);
    auto [boxes, ordered] = tree.BuildTree();
    Flatten(*s, boxes, ordered);
    s->log = cap.os.str();
    return s;
}

// Follows Scene::LoadObjects (src/Scene.cpp:87-118). The reference numbers materials with a
// process-wide counter (ObjectLoader::materialsLoaded); `firstSlot` receives its value before
// this call so callers can rebase.
void* ref_scene_load(const char* const* paths, uint32_t nPaths, int32_t* firstSlot) {
    RefScene* s = new RefScene();
    CoutCapture cap;
    std::vector<Tri> triangles;
    int32_t first = -1;
    for (uint32_t i = 0; i < nPaths; ++i) {
        std::string path = paths[i];
        std::unique_ptr<ObjectLoader> loader;
        if (path.find(".off") != std::string::npos || path.find(".OFF") != std::string::npos) loader = std::make_unique<OFFLoader>();
        else loader = std::make_unique<ObjectLoader>();
        if (!loader->TargetFile(path)) { std::cout << "unable to read from: " << path << std::endl; continue; }
        std::optional<Material::Material> material = loader->ExtractMaterial();
        if (!material.has_value()) { std::cout << "unable to read material from: " << path << std::endl; continue; }
        PushMaterial(*s, material.value());
        std::optional<std::vector<Tri>> tris = loader->ExtractTriangles();
        if (!tris.has_value()) { std::cout << "unable to read triangles from: " << path << std::endl; continue; }
        if (first < 0 && !tris.value().empty()) first = tris.value()[0].materialsIndex;
        triangles.insert(triangles.end(), tris.value().begin(), tris.value().end());
    }
    if (firstSlot) *firstSlot = first;
    BvhTree tree(triangles);
    auto [boxes, ordered] = tree.BuildTree();
    Flatten(*s, boxes, ordered);
    s->log = cap.os.str();
    return s;
}

void ref_scene_free(void* h) { delete (RefScene*)h; }
uint32_t ref_scene_triangle_count(void* h) { return (uint32_t)((RefScene*)h)->matIdx.size(); }
uint32_t ref_scene_node_count(void* h) { return (uint32_t)(((RefScene*)h)->node8.size() / 8); }
uint32_t ref_scene_material_count(void* h) { return (uint32_t)(((RefScene*)h)->mats11.size() / 11); }
const float* ref_scene_tri12(void* h) { return ((RefScene*)h)->tri12.data(); }
const int32_t* ref_scene_matidx(void* h) { return ((RefScene*)h)->matIdx.data(); }
const float* ref_scene_node8(void* h) { return ((RefScene*)h)->node8.data(); }
const float* ref_scene_mats11(void* h) { return ((RefScene*)h)->mats11.data(); }
const char* ref_scene_log(void* h) { return ((RefScene*)h)->log.c_str(); }

}  // extern "C"
