// Scene assembly on the host: object files -> triangles + material table -> BVH -> the three
// flat arrays the renderer consumes (reference: src/Scene.cpp:34-131).
#pragma once

#include <string>
#include <vector>

#include "bvh_tree.h"
#include "geometry.h"

namespace rta {

struct SceneData {
    std::vector<Material> materials;    // at most 16 are usable by the renderer
    std::vector<float> tri12;           // x y z 0 per corner, leaf order
    std::vector<int32_t> matIdx;        // material slot per triangle, leaf order
    std::vector<float> node8;           // max.xyz min.xyz float(count) float(count ? triStart : rightChild)
    BvhStats stats;
    std::vector<std::string> warnings;  // what the reference prints and carries on from

    uint32_t TriangleCount() const { return (uint32_t)matIdx.size(); }
    uint32_t NodeCount() const { return (uint32_t)(node8.size() / 8); }
};

// Loads every listed file (all objects merge into one scene; the k-th file whose material line
// parses gets material slot k), builds the BVH and flattens. Files that fail are skipped with a
// warning, as in Scene::LoadObjects.
SceneData LoadScene(const std::vector<std::string>& objectPaths, int buildThreads = 1);

// Same, from triangles already in memory.
SceneData BuildScene(std::vector<Triangle> triangles, std::vector<Material> materials, int buildThreads = 1,
                     int maxTrianglesPerLeaf = BvhTree::kDefaultLeafTriangles);

// The three flatteners on their own (src/Scene.cpp:34-85).
std::vector<float> FlattenTriangleVertices(const std::vector<Triangle>& ordered);
std::vector<int32_t> FlattenTriangleMaterials(const std::vector<Triangle>& ordered);
std::vector<float> FlattenNodes(const std::vector<BvhNode>& nodes);

}  // namespace rta
