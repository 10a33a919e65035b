#include "bvh_tree.h"

#include <cfloat>
#include <chrono>
#include <future>

namespace rta {
namespace {

// Per-triangle record the builder shuffles: 40 bytes instead of the whole triangle, so that the
// 57 trial partitions per node stream through memory.
struct Item {
    float c[3];    // centroid
    float lo[3];
    float hi[3];
    uint32_t id;   // index into the input triangle list
};

struct Bounds { Vec3 lo, hi; };

// The 19 trial ratios (reference: src/BvhTree.cpp:3-6).
constexpr int kRatios = 19;
const float kSplitRatios[kRatios] = {0.05f, 0.10f, 0.15f, 0.20f, 0.25f, 0.30f, 0.35f, 0.40f, 0.45f, 0.50f,
                                     0.55f, 0.60f, 0.65f, 0.70f, 0.75f, 0.80f, 0.85f, 0.90f, 0.95f};

inline Bounds BoundsOf(const Item* first, const Item* last) {   // src/BvhTree.cpp:64-76
    Bounds b{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    for (const Item* it = first; it != last; ++it) {
        b.lo.x = std::min(b.lo.x, it->lo[0]); b.lo.y = std::min(b.lo.y, it->lo[1]); b.lo.z = std::min(b.lo.z, it->lo[2]);
        b.hi.x = std::max(b.hi.x, it->hi[0]); b.hi.y = std::max(b.hi.y, it->hi[1]); b.hi.z = std::max(b.hi.z, it->hi[2]);
    }
    return b;
}

inline float HalfAreaTimesTwo(const Vec3& e) { return 2.0f * (e.x * e.y + e.x * e.z + e.y * e.z); }   // :19-21

// In-place partition "centroid[axis] < plane first", with the element order libstdc++'s
// std::partition produces for bidirectional iterators (the reference relies on that order:
// src/BvhTree.cpp:78-90): scan from the left to the first element that fails, scan from the
// right to the first that passes, swap, repeat.
inline Item* SplitRange(Item* first, Item* last, int axis, float plane) {
    for (;;) {
        for (;;) {
            if (first == last) return first;
            if (first->c[axis] < plane) ++first; else break;
        }
        --last;
        for (;;) {
            if (first == last) return first;
            if (!(last->c[axis] < plane)) --last; else break;
        }
        std::swap(*first, *last);
        ++first;
    }
}

struct Trial { float score, plane; int axis; };

// Strict weak order of std::tuple<float,float,int> (score, plane, axis): src/BvhTree.cpp:58.
inline bool TrialLess(const Trial& a, const Trial& b) {
    if (a.score < b.score) return true;
    if (b.score < a.score) return false;
    if (a.plane < b.plane) return true;
    if (b.plane < a.plane) return false;
    return a.axis < b.axis;
}

// Chooses the split plane for [first,last); physically re-partitions the range once per trial,
// exactly as the reference does (src/BvhTree.cpp:38-62), so the element order afterwards matches.
inline void ChooseSplit(Item* first, Item* last, const Bounds& box, int& axisOut, float& planeOut) {
    Vec3 planes[kRatios];
    for (int k = 0; k < kRatios; ++k) planes[k] = box.lo * (1 - kSplitRatios[k]) + box.hi * kSplitRatios[k];
    Trial best{}; bool have = false;
    for (int axis = 0; axis < 3; ++axis) {
        for (int k = 0; k < kRatios; ++k) {
            const float plane = planes[k][axis];
            Item* mid = SplitRange(first, last, axis, plane);
            if (mid == first || mid == last) continue;
            Bounds l = BoundsOf(first, mid), r = BoundsOf(mid, last);
            float areaL = HalfAreaTimesTwo(l.hi - l.lo), areaR = HalfAreaTimesTwo(r.hi - r.lo);
            unsigned countL = (unsigned)(mid - first), countR = (unsigned)(last - mid);
            Trial t{areaL * countL + areaR * countR, plane, axis};       // :23-36
            if (!have || TrialLess(t, best)) { best = t; have = true; }  // first minimum wins
        }
    }
    if (have) { axisOut = best.axis; planeOut = best.plane; return; }
    // No plane separates anything: halve the longest extent (src/BvhTree.cpp:8-17, evaluated in double).
    Vec3 d = box.hi - box.lo;
    if (d.x >= d.y && d.x >= d.z) { axisOut = 0; planeOut = (float)(box.lo.x + d.x * 0.5); }
    else if (d.y >= d.x && d.y >= d.z) { axisOut = 1; planeOut = (float)(box.lo.y + d.y * 0.5); }
    else { axisOut = 2; planeOut = (float)(box.lo.z + d.z * 0.5); }
}

struct Builder {
    Item* items;
    int maxLeaf;

    // Emits node + both subtrees (pre-order) into `out`; indices are positions in `out`.
    // Returns the index of the emitted node (src/BvhTree.cpp:92-122).
    int Emit(int l, int r, int depth, std::vector<BvhNode>& out, BvhStats& st) const {
        Bounds box = BoundsOf(items + l, items + r);
        const int me = (int)out.size();
        out.push_back(BvhNode{box.hi, box.lo, -1, -1, 0});
        if (r - l > maxLeaf) {
            st.maxDepth = std::max(st.maxDepth, depth);
            int mid = Divide(l, r, box, st);
            Emit(l, mid, depth + 1, out, st);
            int right = Emit(mid, r, depth + 1, out, st);
            out[me].rightChild = right;
            st.splits++;
        } else {
            out[me].triStart = l;
            out[me].triCount = r - l;
            st.leafDepthSum += (uint64_t)depth;
            st.leaves++;
        }
        return me;
    }

    int Divide(int l, int r, const Bounds& box, BvhStats& st) const {
        int axis; float plane;
        ChooseSplit(items + l, items + r, box, axis, plane);
        Item* m = SplitRange(items + l, items + r, axis, plane);
        int mid = (int)(m - items);
        if (mid == l || mid == r) {                       // all on one side: cut the list in half
            mid = l + (int)((r - l) * 0.5);
            st.degenerateSplits++;
        }
        return mid;
    }

    // Same tree, with the two subtrees of large nodes built concurrently and stitched back
    // into pre-order. Indices inside the returned vector are local to it.
    std::vector<BvhNode> EmitParallel(int l, int r, int depth, int forkLevels, BvhStats& st) const {
        std::vector<BvhNode> out;
        if (forkLevels <= 0 || r - l < 8192) { Emit(l, r, depth, out, st); return out; }
        Bounds box = BoundsOf(items + l, items + r);
        out.push_back(BvhNode{box.hi, box.lo, -1, -1, 0});
        st.maxDepth = std::max(st.maxDepth, depth);
        int mid = Divide(l, r, box, st);
        st.splits++;
        BvhStats stL, stR;
        auto left = std::async(std::launch::async, [&] { return EmitParallel(l, mid, depth + 1, forkLevels - 1, stL); });
        std::vector<BvhNode> rightNodes = EmitParallel(mid, r, depth + 1, forkLevels - 1, stR);
        std::vector<BvhNode> leftNodes = left.get();
        const int offL = 1, offR = 1 + (int)leftNodes.size();
        out.reserve(1 + leftNodes.size() + rightNodes.size());
        for (BvhNode n : leftNodes) { if (n.triCount == 0) n.rightChild += offL; out.push_back(n); }
        for (BvhNode n : rightNodes) { if (n.triCount == 0) n.rightChild += offR; out.push_back(n); }
        out[0].rightChild = offR;
        for (const BvhStats* s : {&stL, &stR}) {
            st.maxDepth = std::max(st.maxDepth, s->maxDepth);
            st.splits += s->splits; st.degenerateSplits += s->degenerateSplits;
            st.leaves += s->leaves; st.leafDepthSum += s->leafDepthSum;
        }
        return out;
    }
};

}  // namespace

BvhTree::BvhTree(std::vector<Triangle> triangles, int maxTrianglesPerLeaf)
    : triangles_(std::move(triangles)), maxLeaf_(maxTrianglesPerLeaf) {}

void BvhTree::Build(int threads) {
    nodes_.clear();
    stats_ = BvhStats{};
    if (triangles_.empty()) return;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<Item> items(triangles_.size());
    for (size_t i = 0; i < triangles_.size(); ++i) {
        const Triangle& t = triangles_[i];
        items[i] = Item{{t.centroid.x, t.centroid.y, t.centroid.z}, {t.lo.x, t.lo.y, t.lo.z}, {t.hi.x, t.hi.y, t.hi.z}, (uint32_t)i};
    }
    Builder b{items.data(), maxLeaf_};
    const int n = (int)items.size();
    if (threads <= 1) {
        nodes_.reserve((size_t)n * 2);
        b.Emit(0, n, 1, nodes_, stats_);
    } else {
        int levels = 0;
        while ((1 << levels) < threads * 2) ++levels;
        nodes_ = b.EmitParallel(0, n, 1, levels, stats_);
    }
    std::vector<Triangle> ordered;
    ordered.reserve(triangles_.size());
    for (const Item& it : items) ordered.push_back(triangles_[it.id]);
    triangles_.swap(ordered);
    stats_.buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rta
