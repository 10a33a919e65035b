// Bounding-volume hierarchy builder that reproduces the reference's tree exactly:
// same candidate planes, same scoring, same tie-breaks, same in-place partition order and the
// same pre-order (depth-first) node numbering, so that node and triangle arrays are
// bit-identical to what the reference uploads (reference: src/BvhTree.cpp, Include/BvhTree.h).
#pragma once

#include <cstdint>
#include <vector>

#include "geometry.h"

namespace rta {

struct BvhNode {
    Vec3 hi, lo;
    int32_t rightChild = -1;   // interior: index of the second child (the first is this + 1)
    int32_t triStart = -1;     // leaf: first triangle of the reordered list
    int32_t triCount = 0;      // leaf: number of triangles (> 0 only for leaves)
};

struct BvhStats {
    int maxDepth = 0;              // deepest interior node, root = 1
    uint64_t splits = 0;
    uint64_t degenerateSplits = 0;
    uint32_t leaves = 0;
    uint64_t leafDepthSum = 0;
    double buildMs = 0.0;
};

class BvhTree {
public:
    static constexpr int kDefaultLeafTriangles = 3;    // Include/BvhTree.h:31

    explicit BvhTree(std::vector<Triangle> triangles, int maxTrianglesPerLeaf = kDefaultLeafTriangles);

    // Builds the tree. `threads` > 1 builds independent subtrees concurrently; the result does
    // not depend on the thread count.
    void Build(int threads = 1);

    const std::vector<BvhNode>& Nodes() const { return nodes_; }
    // Triangles in leaf order (what node.triStart indexes).
    const std::vector<Triangle>& Triangles() const { return triangles_; }
    const BvhStats& Stats() const { return stats_; }

private:
    std::vector<Triangle> triangles_;
    std::vector<BvhNode> nodes_;
    BvhStats stats_;
    int maxLeaf_;
};

}  // namespace rta
