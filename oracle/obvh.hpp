// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of the reference's SAH BVH build and traversal.
//
// Reference: src/bvh.cpp:54-306 (BVHBuildTask::execute / execute_serially), :329-382 (build + compaction),
//            :384-402 (statistics), :404-463 (rayIntersect); include/nori/bvh.h:127-164 (32-byte node).
//
// Determinism contract (SURVEY §7 "BVH determinism"): the reference's parallel partition hands out output
// ranges with atomics, so its result depends on TBB scheduling for subtrees > GRAIN_SIZE primitives.  This
// restatement is the 1-thread schedule: chunks run in index order, i.e. a stable partition.  The serial phase
// uses the same libstdc++ std::sort calls in the same sequence (axis 0,1,2, then best axis) so that ties
// between equal centroids land exactly where the reference puts them.
//
// Nodes are emitted directly in pre-order.  The reference builds into a 2N array (left child = idx+1, right =
// idx + 2*left_count) and then removes unused entries preserving order (:360-375); left/right subtree index
// ranges are disjoint and ordered, so that compaction IS the pre-order numbering.
#pragma once
#include "omath.hpp"
#include <vector>
#include <algorithm>

namespace orc {

struct BvhNode {                    // 32 bytes, same information as include/nori/bvh.h:127-164
    uint32_t w0;                    // bit0 = leaf flag; bits 1..31 = leaf size | inner axis
    uint32_t w1;                    // leaf: start into indices | inner: right child
    float mn[3], mx[3];
    bool is_leaf() const { return w0 & 1u; }
    uint32_t size() const { return w0 >> 1; }
    uint32_t axis() const { return w0 >> 1; }
    uint32_t start() const { return w1; }
    uint32_t right() const { return w1; }
    BBox bbox() const { BBox b; b.mn = mk3(mn[0], mn[1], mn[2]); b.mx = mk3(mx[0], mx[1], mx[2]); return b; }
    void set_bbox(const BBox &b) { mn[0] = b.mn.x; mn[1] = b.mn.y; mn[2] = b.mn.z; mx[0] = b.mx.x; mx[1] = b.mx.y; mx[2] = b.mx.z; }
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> indices;
    float sah_cost = 0; uint32_t leaf_count = 0, max_leaf = 0, depth = 0;
};

// Prims must provide: uint32_t count() const; V3 centroid(uint32_t) const; BBox bbox(uint32_t) const;
template <typename Prims> class BvhBuilder {
public:
    enum { SERIAL_THRESHOLD = 32, BIN_COUNT = 16 };   // src/bvh.cpp:38,62-74 (TRAVERSAL_COST = INTERSECTION_COST = 1)
    BvhBuilder(const Prims &p, Bvh &out) : prims(p), bvh(out) {}

    void build(const BBox &root_bbox) {
        uint32_t n = prims.count();
        bvh.nodes.clear(); bvh.indices.clear();
        if (n == 0) return;
        cen.resize(n); box.resize(n);
        for (uint32_t i = 0; i < n; ++i) { cen[i] = prims.centroid(i); box[i] = prims.bbox(i); }
        bvh.indices.resize(n); temp.resize(n);
        for (uint32_t i = 0; i < n; ++i) bvh.indices[i] = i;
        bvh.nodes.reserve(2 * (size_t) n / 3 + 16);
        build_node(root_bbox, 0, n);
        std::pair<float, uint32_t> st = statistics(0, 1);
        bvh.sah_cost = st.first;
    }

private:
    const Prims &prims; Bvh &bvh;
    std::vector<V3> cen; std::vector<BBox> box; std::vector<uint32_t> temp;

    // returns index of the emitted node
    uint32_t build_node(const BBox &node_bbox, uint32_t s, uint32_t e) {
        uint32_t size = e - s;
        if (size < SERIAL_THRESHOLD) return build_serial(node_bbox, s, e);

        uint32_t *start = bvh.indices.data() + s;
        int axis = node_bbox.largest_axis();
        float mn = node_bbox.mn[axis], mx = node_bbox.mx[axis], inv_bin_size = BIN_COUNT / (mx - mn);

        uint32_t counts[BIN_COUNT]; BBox bbins[BIN_COUNT];
        for (int i = 0; i < BIN_COUNT; ++i) counts[i] = 0;
        for (uint32_t i = 0; i < size; ++i) {
            uint32_t f = start[i];
            float centroid = cen[f][axis];
            int index = std::min(std::max((int) ((centroid - mn) * inv_bin_size), 0), (BIN_COUNT - 1));
            counts[index]++;
            bbins[index].expand(box[f]);
        }
        BBox bbox_left[BIN_COUNT];
        bbox_left[0] = bbins[0];
        for (int i = 1; i < BIN_COUNT; ++i) { counts[i] += counts[i - 1]; bbox_left[i] = BBox::merge(bbox_left[i - 1], bbins[i]); }

        BBox bbox_right = bbins[BIN_COUNT - 1], best_bbox_right;
        int64_t best_index = -1;
        float best_cost = (float) 1 * size;
        float tri_factor = (float) 1 / node_bbox.surface_area();
        for (int i = BIN_COUNT - 2; i >= 0; --i) {
            uint32_t prims_left = counts[i], prims_right = size - counts[i];
            float sah_cost = 2.0f * 1 + tri_factor * (prims_left * bbox_left[i].surface_area() + prims_right * bbox_right.surface_area());
            if (sah_cost < best_cost) { best_cost = sah_cost; best_index = i; best_bbox_right = bbox_right; }
            bbox_right = BBox::merge(bbox_right, bbins[i]);
        }
        if (best_index == -1) return build_serial(node_bbox, s, e);

        uint32_t left_count = counts[best_index];
        // stable partition (1-thread schedule of src/bvh.cpp:188-214)
        uint32_t il = 0, ir = left_count; uint32_t *tmp = temp.data() + s;
        for (uint32_t i = 0; i < size; ++i) {
            uint32_t f = start[i];
            float centroid = cen[f][axis];
            int index = (int) ((centroid - mn) * inv_bin_size);
            if (index <= best_index) tmp[il++] = f; else tmp[ir++] = f;
        }
        std::copy(tmp, tmp + size, start);

        uint32_t me = (uint32_t) bvh.nodes.size();
        bvh.nodes.push_back(BvhNode());
        BBox lbox = bbox_left[best_index], rbox = best_bbox_right;
        build_node(lbox, s, s + left_count);
        uint32_t r = build_node(rbox, s + left_count, e);
        BvhNode &nd = bvh.nodes[me];
        nd.w0 = ((uint32_t) axis) << 1; nd.w1 = r; nd.set_bbox(node_bbox);
        return me;
    }

    uint32_t build_serial(const BBox &node_bbox_in, uint32_t s, uint32_t e) {
        uint32_t *start = bvh.indices.data() + s, *end = bvh.indices.data() + e;
        uint32_t size = e - s;
        BBox node_bbox = node_bbox_in;
        float best_cost = (float) 1 * size;
        int64_t best_index = -1, best_axis = -1;
        float *left_areas = (float *) (temp.data() + s);

        for (int axis = 0; axis < 3; ++axis) {
            std::sort(start, end, [&](uint32_t f1, uint32_t f2) { return cen[f1][axis] < cen[f2][axis]; });
            BBox bb;
            for (uint32_t i = 0; i < size; ++i) { bb.expand(box[start[i]]); left_areas[i] = (float) bb.surface_area(); }
            if (axis == 0) node_bbox = bb;
            bb.reset();
            float tri_factor = 1 / node_bbox.surface_area();
            for (uint32_t i = size - 1; i >= 1; --i) {
                bb.expand(box[start[i]]);
                float left_area = left_areas[i - 1];
                float right_area = bb.surface_area();
                uint32_t prims_left = i, prims_right = size - i;
                float sah_cost = 2.0f * 1 + tri_factor * (prims_left * left_area + prims_right * right_area);
                if (sah_cost < best_cost) { best_cost = sah_cost; best_index = i; best_axis = axis; }
            }
        }
        uint32_t me = (uint32_t) bvh.nodes.size();
        bvh.nodes.push_back(BvhNode());
        if (best_index == -1) {
            BvhNode &nd = bvh.nodes[me];
            nd.w0 = 1u | (size << 1); nd.w1 = s; nd.set_bbox(node_bbox);
            return me;
        }
        std::sort(start, end, [&](uint32_t f1, uint32_t f2) { return cen[f1][best_axis] < cen[f2][best_axis]; });
        uint32_t left_count = (uint32_t) best_index;
        // children boxes are (re)computed by their own serial step; the incoming value is irrelevant there
        build_serial(BBox(), s, s + left_count);
        uint32_t r = build_serial(BBox(), s + left_count, e);
        BvhNode &nd = bvh.nodes[me];
        nd.w0 = ((uint32_t) best_axis) << 1; nd.w1 = r; nd.set_bbox(node_bbox);
        return me;
    }

    // src/bvh.cpp:384-402
    std::pair<float, uint32_t> statistics(uint32_t idx, uint32_t depth) {
        const BvhNode &node = bvh.nodes[idx];
        bvh.depth = std::max(bvh.depth, depth);
        if (node.is_leaf()) {
            bvh.leaf_count++; bvh.max_leaf = std::max(bvh.max_leaf, node.size());
            return std::make_pair((float) 1 * node.size(), 1u);
        }
        std::pair<float, uint32_t> sl = statistics(idx + 1u, depth + 1), sr = statistics(node.right(), depth + 1);
        float saLeft = bvh.nodes[idx + 1u].bbox().surface_area();
        float saRight = bvh.nodes[node.right()].bbox().surface_area();
        float saCur = node.bbox().surface_area();
        float sah = 2 * 1 + (saLeft * sl.first + saRight * sr.first) / saCur;
        return std::make_pair(sah, sl.second + sr.second + 1u);
    }
};

} // namespace orc
