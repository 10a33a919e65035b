#include "scene_data.h"

#include "object_loader.h"

namespace rta {

std::vector<float> FlattenTriangleVertices(const std::vector<Triangle>& ordered) {
    std::vector<float> out(ordered.size() * 12);
    float* w = out.data();
    for (const Triangle& t : ordered) {
        for (const Vec3* p : {&t.p0, &t.p1, &t.p2}) { *w++ = p->x; *w++ = p->y; *w++ = p->z; *w++ = 0.0f; }
    }
    return out;
}

std::vector<int32_t> FlattenTriangleMaterials(const std::vector<Triangle>& ordered) {
    std::vector<int32_t> out(ordered.size());
    for (size_t i = 0; i < ordered.size(); ++i) out[i] = ordered[i].material;
    return out;
}

std::vector<float> FlattenNodes(const std::vector<BvhNode>& nodes) {
    std::vector<float> out(nodes.size() * 8);
    float* w = out.data();
    for (const BvhNode& n : nodes) {
        *w++ = n.hi.x; *w++ = n.hi.y; *w++ = n.hi.z;
        *w++ = n.lo.x; *w++ = n.lo.y; *w++ = n.lo.z;
        *w++ = (float)n.triCount;
        *w++ = (float)(n.triCount == 0 ? n.rightChild : n.triStart);
    }
    return out;
}

SceneData BuildScene(std::vector<Triangle> triangles, std::vector<Material> materials, int buildThreads, int maxLeaf) {
    SceneData s;
    s.materials = std::move(materials);
    BvhTree tree(std::move(triangles), maxLeaf);
    tree.Build(buildThreads);
    s.tri12 = FlattenTriangleVertices(tree.Triangles());
    s.matIdx = FlattenTriangleMaterials(tree.Triangles());
    s.node8 = FlattenNodes(tree.Nodes());
    s.stats = tree.Stats();
    return s;
}

SceneData LoadScene(const std::vector<std::string>& objectPaths, int buildThreads) {
    std::vector<Triangle> triangles;
    std::vector<Material> materials;
    std::vector<std::string> warnings;
    for (const std::string& path : objectPaths) {
        LoadedObject obj;
        std::string msg;
        LoadStatus st = LoadObjectFile(path, (int)materials.size(), obj, &msg);
        if (st == LoadStatus::CannotOpen) { warnings.push_back("unable to read from: " + path); continue; }
        if (st == LoadStatus::BadMaterial) { warnings.push_back("unable to read material from: " + path + " (" + msg + ")"); continue; }
        materials.push_back(obj.material);   // registered even when the geometry then fails
        if (st == LoadStatus::BadGeometry) { warnings.push_back("unable to read triangles from: " + path + " (" + msg + ")"); continue; }
        triangles.insert(triangles.end(), obj.triangles.begin(), obj.triangles.end());
    }
    SceneData s = BuildScene(std::move(triangles), std::move(materials), buildThreads);
    s.warnings = std::move(warnings);
    return s;
}

}  // namespace rta
