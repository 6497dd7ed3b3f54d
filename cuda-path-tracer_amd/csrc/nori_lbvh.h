// Device-side BVH construction: a linear BVH over Morton-sorted primitive records, emitted in the
// 64-byte two-box node layout of include/nori_hip.h.  NOT the reference's tree (src/bvh.cpp builds a SAH tree on the
// host; cuda-path-tracer_amd/host/bvh.cpp reproduces that one and stays the parity mode): closest hits are the same
// primitive except where two primitives tie in t, any-hit answers are identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/nori_hip.h"

namespace lbvh {

struct Result {
    uint32_t root;                 // child reference of the root
    uint32_t n_nodes;              // nodes in the (packed) node array
    uint32_t n_used, depth;        // = n_nodes; depth of the uncollapsed tree (levels of inner nodes)
    float root_min[3], root_max[3];
    double build_ms;               // device time of the whole build (HIP events)
    void *nodes4 = nullptr;        // the same tree as 128-byte four-box nodes (Trav::box_step4), hipMalloc'ed here, owned by the caller;
    uint32_t n_nodes4 = 0;         // null when the tree is one leaf.  Its root is node 0.
    uint32_t depth4 = 0;           // levels of four-box nodes (a traversal pushes at most three entries per level)
    void *nodes8 = nullptr;        // (want8) the same tree as 128-byte eight-box nodes with quantised boxes (Trav::box_step8), hipMalloc'ed here,
    uint32_t n_nodes8 = 0;         // owned by the caller; root = node 0
    uint32_t depth8 = 0;           // levels of eight-box nodes
    uint32_t stack8 = 0;           // most stack entries a traversal of the eight-box nodes can hold at once
    void *prims_al = nullptr, *shade_al = nullptr;   // (align_leaves) the records placed so that no leaf straddles a 128-byte line, hipMalloc'ed here, owned by the
    uint32_t n_positions = 0;      // caller, n_positions records each (unused positions are zero records that no leaf names); prims_out / shade_out are then unused
};

// prims_in / shade_in: n records in any order (bit 31 of shape_flags ignored).  prims_out / shade_out: n records in
// leaf order; nodes_out: max(n - 1, 1) nodes.  All device pointers.  Leaves hold at most `leaf_max` records; records
// longer than `big` times the scene on some axis go into a subtree of their own under the root (0: off).
// plain_morton: keys interleave x, y, z bit by bit; otherwise the bits are ordered by the size of the cell they halve and
// bits finer than the typical record on their axis come last (see plan_bits in nori_lbvh.hip).
// align_leaves (leaf_max <= 2 only): see Result::prims_al.
// want8: also convert the tree into quantised eight-box nodes (Result::nodes8).
// Returns 0 or a negative NORI_HIP_E* code with `err` set.
int build(hipStream_t stream, uint32_t n, const NoriHipPrim *prims_in, const NoriHipShadePrim *shade_in, NoriHipPrim *prims_out,
          NoriHipShadePrim *shade_out, NoriHipNode *nodes_out, uint32_t leaf_max, float big, uint32_t max_depth, bool plain_morton, bool want8, bool align_leaves, Result &res, std::string &err);

// the four-box nodes of a tree as 64-byte nodes with quantised boxes (layout: k_quantize4 in nori_lbvh.hip; Trav::box_step4q)
hipError_t quantize4(hipStream_t stream, const void *nodes4, uint32_t n, void **out);

// radix sort of n (key, value) pairs of 32 bits each on key bits [begin_bit, end_bit); `temp` of sort_pairs_temp_bytes(n) bytes
size_t sort_pairs_temp_bytes(uint32_t n);
hipError_t sort_pairs(hipStream_t stream, void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, int end_bit, int begin_bit = 0);

} // namespace lbvh
