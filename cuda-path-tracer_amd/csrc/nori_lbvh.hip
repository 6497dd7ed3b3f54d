// Linear BVH construction on the GPU (see nori_lbvh.h).
//
//   k_boxes     per record: its box (the reference's per-primitive boxes, src/mesh.cpp:170-175, src/sphere.cpp:38) and
//               the bounds of the box centres (block reduction, then six atomics per block)
//   k_morton    63-bit Morton code of the box centre (21 bits per axis), value = record index
//   hipcub      radix sort of (code, index)
//   k_hierarchy one thread per inner node: its range of sorted records and split (Karras 2012, "Maximizing
//               parallelism in the construction of BVHs, octrees and k-d trees"); equal codes are told apart by
//               their position, so keys are unique
//   k_refit_pass  bottom-up in passes: a node whose children are finished takes the union of their boxes
//   k_depth     depth of every leaf -> depth of the tree (the traversal stack holds ND_STACK entries)
//   k_reorder   records into sorted (leaf) order
//   k_emit      one thread per inner node: the 64-byte node with both children's boxes; subtrees of at most
//               `leaf_max` records become one leaf (their records are contiguous in sorted order)
//   hipcub scan + k_compact   the nodes in use, packed (the nodes inside collapsed subtrees are dropped)
#include "nori_lbvh.h"
#include <hipcub/hipcub.hpp>
#include <vector>
#include <algorithm>
#include <cmath>

namespace lbvh {
namespace {

#define LB_LEAF 0x80000000u

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f2o(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float o2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    union { uint32_t u; float f; } c; c.u = u; return c.f;
}

__device__ __forceinline__ void record_box(const NoriHipPrim &p, const NoriHipShadePrim &s, float mn[3], float mx[3]) {
    if (p.shape_flags & 0x40000000u) {                       // sphere: centre -+ radius (src/sphere.cpp:33-34)
        for (int a = 0; a < 3; ++a) { mn[a] = p.a[a] - p.e1[0]; mx[a] = p.a[a] + p.e1[0]; }
        return;
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = fminf(fminf(s.p0[a], s.p1[a]), s.p2[a]);
        mx[a] = fmaxf(fmaxf(s.p0[a], s.p1[a]), s.p2[a]);
    }
}

// log2 of a box extent in 1/256ths, clamped to [2^-40, 2^40]: summed in integers, so the sum (and with it the bit plan of the
// keys) does not depend on the order of the atomics
__device__ __forceinline__ long long log2_fix(float e) { return (long long) (log2f(fminf(fmaxf(e, 9.0949470e-13f), 1.0995116e12f)) * 256.0f); }

__global__ __launch_bounds__(256) void k_boxes(uint32_t n, const NoriHipPrim *prims, const NoriHipShadePrim *shade, float4 *pmin, float4 *pmax, uint32_t *cbounds,
                                               long long *log_extent /* 3: sum over the records of log2(extent) per axis, 1/256ths */) {
    __shared__ uint32_t red[6]; __shared__ long long lsum[3];
    if (threadIdx.x < 3) { red[threadIdx.x] = 0xffffffffu; lsum[threadIdx.x] = 0; } else if (threadIdx.x < 6) red[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    long long le[3] = {0, 0, 0};
    if (i < n) {
        float mn[3], mx[3]; record_box(prims[i], shade[i], mn, mx);
        pmin[i] = make_float4(mn[0], mn[1], mn[2], 0.f); pmax[i] = make_float4(mx[0], mx[1], mx[2], 0.f);
        for (int a = 0; a < 3; ++a) { float c = 0.5f * (mn[a] + mx[a]); atomicMin(&red[a], f2o(c)); atomicMax(&red[3 + a], f2o(c)); le[a] = log2_fix(mx[a] - mn[a]); }
    }
    for (int a = 0; a < 3; ++a) {
        long long v = le[a];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63u) == 0u) atomicAdd((unsigned long long *) &lsum[a], (unsigned long long) v);
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&cbounds[threadIdx.x], red[threadIdx.x]); atomicAdd((unsigned long long *) &log_extent[threadIdx.x], (unsigned long long) lsum[threadIdx.x]); }
    else if (threadIdx.x < 6) atomicMax(&cbounds[threadIdx.x], red[threadIdx.x]);
}

// Which bit of which axis' 21-bit cell coordinate goes where in the 63-bit key, most significant first (see plan_bits).
struct BitPlan { unsigned char axis[63], bit[63]; };

// Records whose box is longer than `big` times the extent of the centre bounds on some axis get the top key bit: they
// sort behind all others and the hierarchy splits them off at the root, into a subtree of their own.  (A room-sized
// wall triangle that shares a Morton cell with millimetre-sized ones would otherwise inflate every box above it.)
__global__ __launch_bounds__(256) void k_morton(uint32_t n, const float4 *pmin, const float4 *pmax, const uint32_t *cbounds, float big, BitPlan plan, unsigned long long *keys, uint32_t *vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = pmin[i], b = pmax[i];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)}, e[3] = {b.x - a.x, b.y - a.y, b.z - a.z};
    uint32_t q[3]; bool large = false;
    for (int k = 0; k < 3; ++k) {
        const float lo = o2f(cbounds[k]), hi = o2f(cbounds[3 + k]);
        float t = (hi > lo) ? (c[k] - lo) / (hi - lo) : 0.f;
        q[k] = (uint32_t) fminf(fmaxf(t * 2097152.0f, 0.f), 2097151.0f);
        large = large || (big > 0.f && e[k] > big * (hi - lo));
    }
    unsigned long long code = 0;
    for (int p = 0; p < 63; ++p) code = (code << 1) | (unsigned long long) ((q[plan.axis[p]] >> plan.bit[p]) & 1u);
    keys[i] = code | (large ? 0x8000000000000000ull : 0ull); vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; -1 outside the array
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    if (x == 0ull) return 64 + __clz((unsigned int) (i ^ j));
    return __clzll((long long) x);
}

__global__ __launch_bounds__(256) void k_hierarchy(int n, const unsigned long long *keys, uint32_t *left, uint32_t *right, uint32_t *first, uint32_t *last,
                                                   uint32_t *parent_inner, uint32_t *parent_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t lref = (lo == gamma) ? (LB_LEAF | (uint32_t) gamma) : (uint32_t) gamma;
    const uint32_t rref = (hi == gamma + 1) ? (LB_LEAF | (uint32_t) (gamma + 1)) : (uint32_t) (gamma + 1);
    left[i] = lref; right[i] = rref; first[i] = (uint32_t) lo; last[i] = (uint32_t) hi;
    if (lref & LB_LEAF) parent_leaf[gamma] = (uint32_t) i; else parent_inner[gamma] = (uint32_t) i;
    if (rref & LB_LEAF) parent_leaf[gamma + 1] = (uint32_t) i; else parent_inner[gamma + 1] = (uint32_t) i;
    if (i == 0) parent_inner[0] = 0xffffffffu;
}

// Boxes of the inner nodes, bottom-up in passes: in pass p a node whose children were both finished in earlier passes takes the
// union of their boxes and records p in `done`.  A child finished in the same pass still reads as unfinished (its mark is p, not
// below p), and what a pass reads of the passes before it is ordered by the kernel boundary, so plain loads and stores do.  The tree
// is `depth` levels high, so `depth` passes finish it.  (One thread per leaf walking up with an arrival counter per node - the usual
// form - needs device-scope 4-byte stores and fences at every step; on ten million records that took 56 ms.)
__global__ __launch_bounds__(256) void k_refit_pass(uint32_t n_inner, uint32_t pass, const uint32_t *vals, const float4 *pmin, const float4 *pmax, const uint32_t *left, const uint32_t *right,
                                                    float *nbox /* 8 floats per inner node: min xyz _, max xyz _ */, uint32_t *done) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || done[i]) return;
    const uint32_t l = left[i], r = right[i];
    if (!(l & LB_LEAF)) { const uint32_t d = done[l]; if (d == 0u || d >= pass) return; }
    if (!(r & LB_LEAF)) { const uint32_t d = done[r]; if (d == 0u || d >= pass) return; }
    float mn[3], mx[3];
    for (int side = 0; side < 2; ++side) {
        const uint32_t c = side ? r : l;
        float4 a, b;
        if (c & LB_LEAF) { a = pmin[vals[c & 0x7fffffffu]]; b = pmax[vals[c & 0x7fffffffu]]; }
        else { const float4 *q = reinterpret_cast<const float4 *>(nbox + 8 * (size_t) c); a = q[0]; b = q[1]; }
        const float cmn[3] = {a.x, a.y, a.z}, cmx[3] = {b.x, b.y, b.z};
        for (int k = 0; k < 3; ++k) { mn[k] = side ? fminf(mn[k], cmn[k]) : cmn[k]; mx[k] = side ? fmaxf(mx[k], cmx[k]) : cmx[k]; }
    }
    float4 *q = reinterpret_cast<float4 *>(nbox + 8 * (size_t) i);
    q[0] = make_float4(mn[0], mn[1], mn[2], 0.f); q[1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    done[i] = pass;
}

__global__ __launch_bounds__(256) void k_depth(uint32_t n, const uint32_t *parent_inner, const uint32_t *parent_leaf, uint32_t *max_depth) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0;
    if (i < n) { d = 1; for (uint32_t cur = parent_leaf[i]; cur != 0u; cur = parent_inner[cur]) ++d; }
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_down(d, off, 64); d = o > d ? o : d; }
    if ((threadIdx.x & 63u) == 0u) atomicMax(max_depth, d);
}

// Where sorted record i goes.  Plain: position i.  Aligned (leaves of one or two records): leaf number c takes the cell of two
// positions 8 (c / 3) + 3 (c % 3) - positions 0-1, 3-4 and 6-7 of every run of eight 48-byte records, none of which crosses a
// 128-byte line (positions 2 and 5 do, and stay empty): a leaf visit then costs one HBM line instead of 1.5 on average.
// leaf_idx = exclusive prefix sum of leaf_start.
__device__ __forceinline__ uint32_t place(uint32_t i, const uint32_t *leaf_start, const uint32_t *leaf_idx) {
    if (!leaf_idx) return i;
    const bool st = leaf_start[i] != 0u;
    const uint32_t c = st ? leaf_idx[i] : leaf_idx[i] - 1u;
    return 8u * (c / 3u) + 3u * (c % 3u) + (st ? 0u : 1u);
}
__global__ __launch_bounds__(256) void k_reorder(uint32_t n, const uint32_t *vals, const NoriHipPrim *pin, const NoriHipShadePrim *sin, NoriHipPrim *pout, NoriHipShadePrim *sout,
                                                 const uint32_t *leaf_start, const uint32_t *leaf_idx, const uint32_t *leaf_end) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = vals[i];
    NoriHipPrim p = pin[src]; p.shape_flags &= 0x7fffffffu;
    if (leaf_end && leaf_end[i]) p.shape_flags |= 0x80000000u;
    const uint32_t dst = place(i, leaf_start, leaf_idx);
    pout[dst] = p; sout[dst] = sin[src];
}
// leaf references of the emitted nodes from sorted positions to placed ones
__global__ __launch_bounds__(256) void k_place_refs(uint32_t n_inner, const uint32_t *used, NoriHipNode *nodes, const uint32_t *leaf_start, const uint32_t *leaf_idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !used[i]) return;
    NoriHipNode &w = nodes[i];
    if (w.left != NORI_HIP_NO_CHILD && (w.left & LB_LEAF)) w.left = LB_LEAF | place(w.left & 0x7fffffffu, leaf_start, leaf_idx);
    if (w.right != NORI_HIP_NO_CHILD && (w.right & LB_LEAF)) w.right = LB_LEAF | place(w.right & 0x7fffffffu, leaf_start, leaf_idx);
}

__global__ __launch_bounds__(256) void k_emit(uint32_t n, uint32_t leaf_max, const uint32_t *vals, const float4 *pmin, const float4 *pmax, const uint32_t *left,
                                              const uint32_t *right, const uint32_t *first, const uint32_t *last, const float *nbox, NoriHipNode *nodes,
                                              uint32_t *leaf_start, uint32_t *leaf_end, uint32_t *used) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    NoriHipNode w;
    if (i != 0u && last[i] - first[i] + 1u <= leaf_max) { used[i] = 0u; return; }   // inside a collapsed subtree: never referenced
    for (int side = 0; side < 2; ++side) {
        const uint32_t c = side ? right[i] : left[i];
        float *mn = side ? w.rmin : w.lmin, *mx = side ? w.rmax : w.lmax;
        uint32_t ref, count;
        if (c & LB_LEAF) {
            const uint32_t pos = c & 0x7fffffffu;
            const float4 a = pmin[vals[pos]], b = pmax[vals[pos]];
            mn[0] = a.x; mn[1] = a.y; mn[2] = a.z; mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
            ref = LB_LEAF | pos; count = 1;
            leaf_start[pos] = 1u; leaf_end[pos] = 1u;
        } else {
            const float *q = nbox + 8 * (size_t) c;
            for (int a = 0; a < 3; ++a) { mn[a] = q[a]; mx[a] = q[4 + a]; }
            const uint32_t sz = last[c] - first[c] + 1u;
            if (sz <= leaf_max) { ref = LB_LEAF | first[c]; count = sz; leaf_start[first[c]] = 1u; leaf_end[last[c]] = 1u; }
            else { ref = c; count = 0; }
        }
        if (side) { w.right = ref; w.rcount = count; } else { w.left = ref; w.lcount = count; }
    }
    nodes[i] = w;
    used[i] = 1u;
}

// the nodes in use, packed in their old order (so the root stays node 0 and neighbours in Morton order stay neighbours)
__global__ __launch_bounds__(256) void k_compact(uint32_t n_inner, const uint32_t *used, const uint32_t *new_index, const NoriHipNode *nodes, NoriHipNode *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !used[i]) return;
    NoriHipNode w = nodes[i];
    if (!(w.left & LB_LEAF)) w.left = new_index[w.left];
    if (!(w.right & LB_LEAF)) w.right = new_index[w.right];
    out[new_index[i]] = w;
}

__global__ void k_mark_last(NoriHipPrim *pout, uint32_t pos) { pout[pos].shape_flags |= 0x80000000u; }

// ---- four-box nodes (Trav::box_step4 in nori_device.h).  A four-box node starts from a binary inner node's two children and,
// while it has a free slot, opens the inner child with the largest box surface (in place, so the slots stay in the binary tree's
// depth-first order); children that are leaves stay.  The inner children left at the end become four-box nodes themselves.  Which
// binary nodes those are depends on the choices above them, so the tree is converted level by level: `queue` holds the binary
// nodes of one level, a count pass and a scan number their inner children, an emit pass writes the level's nodes and the next
// level's queue.  Node numbers: level after level, within a level in queue order (deterministic).
template <int N> struct Slots { float mn[N][3], mx[N][3]; uint32_t ref[N]; int n; };
__device__ __forceinline__ float box_area(const float *mn, const float *mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}
template <int N> __device__ __forceinline__ void open_greedy(const NoriHipNode *nodes, uint32_t b, Slots<N> &s) {
    const float inf = __builtin_huge_valf();
    for (int k = 0; k < N; ++k) { s.ref[k] = NORI_HIP_NO_CHILD; for (int a = 0; a < 3; ++a) { s.mn[k][a] = inf; s.mx[k][a] = -inf; } }
    s.n = 0;
    const NoriHipNode w = nodes[b];
    for (int side = 0; side < 2; ++side) {
        const uint32_t r = side ? w.right : w.left;
        if (r == NORI_HIP_NO_CHILD) continue;
        for (int a = 0; a < 3; ++a) { s.mn[s.n][a] = side ? w.rmin[a] : w.lmin[a]; s.mx[s.n][a] = side ? w.rmax[a] : w.lmax[a]; }
        s.ref[s.n++] = r;
    }
    while (s.n < N) {
        int best = -1; float best_area = -1.f;
        for (int k = 0; k < s.n; ++k) if (!(s.ref[k] & LB_LEAF)) { const float ar = box_area(s.mn[k], s.mx[k]); if (ar > best_area) { best_area = ar; best = k; } }
        if (best < 0) break;
        const NoriHipNode c = nodes[s.ref[best]];
        if (c.right == NORI_HIP_NO_CHILD) {                                // (a node with one child: take the child in its place)
            for (int a = 0; a < 3; ++a) { s.mn[best][a] = c.lmin[a]; s.mx[best][a] = c.lmax[a]; }
            s.ref[best] = c.left; continue;
        }
        for (int k = s.n; k > best + 1; --k) { s.ref[k] = s.ref[k - 1]; for (int a = 0; a < 3; ++a) { s.mn[k][a] = s.mn[k - 1][a]; s.mx[k][a] = s.mx[k - 1][a]; } }
        for (int a = 0; a < 3; ++a) { s.mn[best][a] = c.lmin[a]; s.mx[best][a] = c.lmax[a]; s.mn[best + 1][a] = c.rmin[a]; s.mx[best + 1][a] = c.rmax[a]; }
        s.ref[best] = c.left; s.ref[best + 1] = c.right; s.n++;
    }
}
template <int N> __global__ __launch_bounds__(256) void k_wide_count(uint32_t n_level, const uint32_t *queue, const NoriHipNode *nodes, uint32_t *count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_level) return;
    Slots<N> s; open_greedy<N>(nodes, queue[i], s);
    uint32_t c = 0; for (int k = 0; k < s.n; ++k) if (!(s.ref[k] & LB_LEAF)) ++c;
    count[i] = c;
}
__global__ __launch_bounds__(256) void k_wide_emit(uint32_t n_level, const uint32_t *queue, const NoriHipNode *nodes, const uint32_t *offset, uint32_t level_base, uint32_t next_base,
                                                   uint32_t *next_queue, float4 *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_level) return;
    Slots<4> s; open_greedy<4>(nodes, queue[i], s);
    uint32_t o = offset[i];
    for (int k = 0; k < s.n; ++k) if (!(s.ref[k] & LB_LEAF)) { next_queue[o] = s.ref[k]; s.ref[k] = next_base + o; ++o; }
    float4 *q = out + 8 * (size_t) (level_base + i);
    q[0] = make_float4(s.mn[0][0], s.mn[1][0], s.mn[2][0], s.mn[3][0]); q[1] = make_float4(s.mn[0][1], s.mn[1][1], s.mn[2][1], s.mn[3][1]); q[2] = make_float4(s.mn[0][2], s.mn[1][2], s.mn[2][2], s.mn[3][2]);
    q[3] = make_float4(s.mx[0][0], s.mx[1][0], s.mx[2][0], s.mx[3][0]); q[4] = make_float4(s.mx[0][1], s.mx[1][1], s.mx[2][1], s.mx[3][1]); q[5] = make_float4(s.mx[0][2], s.mx[1][2], s.mx[2][2], s.mx[3][2]);
    q[6] = make_float4(__uint_as_float(s.ref[0]), __uint_as_float(s.ref[1]), __uint_as_float(s.ref[2]), __uint_as_float(s.ref[3]));
    q[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- eight-box nodes with quantised boxes (Trav::box_step8 in nori_device.h), for trees that do not fit the caches: the same
// greedy opening up to eight slots, and the children's boxes stored as 8-bit offsets from the node's own box corner `p` in units of a
// power of two per axis (after Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs through compressed wide
// BVHs"): a 128-byte line decides eight descents instead of four.  Layout, 32 words:
//   0-2  p (float3)        3  exponent bytes ex | ey << 8 | ez << 16 (biased: the unit on axis a is the float with exponent field e_a),
//                             children in use << 24
//   4-5  low x of children 0..7, a byte each | 6-7 low y | 8-9 low z | 10-11 high x | 12-13 high y | 14-15 high z
//   16-23 child references   24-31 unused
// Quantisation only enlarges: low corners round down, high corners up, each checked against the value p + q * unit as a float
// evaluates it.  An unused slot has low = 255, high = 0 and reference NORI_HIP_NO_CHILD.
// `need` carries, per queued node, the stack entries a traversal may hold when it arrives there (a visit of a node with c children
// leaves at most c - 1 behind); the largest need + c - 1 over all nodes is what the traversal stack must hold.
__global__ __launch_bounds__(256) void k_wide_emit8(uint32_t n_level, const uint32_t *queue, const uint32_t *need, const NoriHipNode *nodes, const uint32_t *offset, uint32_t level_base,
                                                    uint32_t next_base, uint32_t *next_queue, uint32_t *next_need, uint4 *out, uint32_t *stack_max) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_level) return;
    Slots<8> s; open_greedy<8>(nodes, queue[i], s);
    const uint32_t held = need[i] + (uint32_t) (s.n > 0 ? s.n - 1 : 0);
    atomicMax(stack_max, held);
    uint32_t o = offset[i];
    for (int k = 0; k < s.n; ++k) if (!(s.ref[k] & LB_LEAF)) { next_queue[o] = s.ref[k]; next_need[o] = held; s.ref[k] = next_base + o; ++o; }
    float p[3], unit[3]; uint32_t eb[3];
    uint32_t qlo[3][8], qhi[3][8];
    for (int a = 0; a < 3; ++a) {
        float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
        for (int k = 0; k < s.n; ++k) { lo = fminf(lo, s.mn[k][a]); hi = fmaxf(hi, s.mx[k][a]); }
        p[a] = lo;
        // the smallest power of two u with 254 u >= extent (254, not 255: room for the upward correction of a high corner)
        const float ext = hi - lo;
        int e = 0; (void) frexpf(ext > 0.f ? ext / 254.0f : 0.f, &e);                 // ext / 254 = m 2^e, m in [0.5, 1)  =>  2^e >= ext / 254
        int be = e + 127; be = be < 1 ? 1 : (be > 254 ? 254 : be);
        eb[a] = (uint32_t) be; unit[a] = __uint_as_float((uint32_t) be << 23);
        for (int k = 0; k < 8; ++k) { qlo[a][k] = 255u; qhi[a][k] = 0u; }
        for (int k = 0; k < s.n; ++k) {
            float ql = floorf((s.mn[k][a] - lo) / unit[a]); ql = fminf(fmaxf(ql, 0.f), 255.f);
            while (ql > 0.f && lo + ql * unit[a] > s.mn[k][a]) ql -= 1.f;
            float qh = ceilf((s.mx[k][a] - lo) / unit[a]); qh = fminf(fmaxf(qh, 0.f), 255.f);
            while (qh < 255.f && lo + qh * unit[a] < s.mx[k][a]) qh += 1.f;
            qlo[a][k] = (uint32_t) ql; qhi[a][k] = (uint32_t) qh;
        }
    }
    auto pack = [](const uint32_t *q) { return q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24); };
    uint4 *q = out + 8 * (size_t) (level_base + i);
    q[0] = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), eb[0] | (eb[1] << 8) | (eb[2] << 16) | ((uint32_t) s.n << 24));
    q[1] = make_uint4(pack(qlo[0]), pack(qlo[0] + 4), pack(qlo[1]), pack(qlo[1] + 4));
    q[2] = make_uint4(pack(qlo[2]), pack(qlo[2] + 4), pack(qhi[0]), pack(qhi[0] + 4));
    q[3] = make_uint4(pack(qhi[1]), pack(qhi[1] + 4), pack(qhi[2]), pack(qhi[2] + 4));
    q[4] = make_uint4(s.ref[0], s.ref[1], s.ref[2], s.ref[3]);
    q[5] = make_uint4(s.ref[4], s.ref[5], s.ref[6], s.ref[7]);
    q[6] = make_uint4(0u, 0u, 0u, 0u); q[7] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- four-box nodes with quantised boxes, 64 bytes (Trav::box_step4q in nori_device.h): the four-box node above with its children's
// boxes stored as 8-bit offsets from the node's own low corner `p`, in units of 1/255 of the node's extent per axis (a float per
// axis: unlike the eight-box node there is room for one, and a unit that is not rounded up to a power of two grows the boxes less).
// Same numbering and child references as the node it is made from, so one thread converts one node.  Layout, 16 words:
//   0-2  p (float3)     3  unit x
//   4 low x of children 0..3, a byte each | 5 low y | 6 low z | 7 high x | 8 high y | 9 high z | 10 unit y | 11 unit z
//   12-15 child references (an unused slot: NORI_HIP_NO_CHILD, low 255, high 0)
// Half the bytes per visit, four 16-byte loads instead of seven, two nodes per HBM line; the boxes only ever grow (low corners
// round down, high corners up), so a ray reaches every record it reached before.
__global__ __launch_bounds__(256) void k_quantize4(uint32_t n, const float4 *nodes4, uint4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 *q = nodes4 + 8 * (size_t) i;
    const float4 mn4[3] = {q[0], q[1], q[2]}, mx4[3] = {q[3], q[4], q[5]}, r4 = q[6];
    const uint32_t ref[4] = {__float_as_uint(r4.x), __float_as_uint(r4.y), __float_as_uint(r4.z), __float_as_uint(r4.w)};
    float p[3], un[3]; uint32_t wlo[3], whi[3];
    for (int a = 0; a < 3; ++a) {
        const float mn[4] = {mn4[a].x, mn4[a].y, mn4[a].z, mn4[a].w}, mx[4] = {mx4[a].x, mx4[a].y, mx4[a].z, mx4[a].w};
        float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
        for (int k = 0; k < 4; ++k) if (ref[k] != NORI_HIP_NO_CHILD) { lo = fminf(lo, mn[k]); hi = fmaxf(hi, mx[k]); }
        p[a] = lo;
        const float ext = hi - lo;
        // (254, not 255: room for the upward correction of a high corner; a flat node gets the smallest normal number)
        float unit = ext / 254.0f; if (!(unit >= 1.17549435e-38f)) unit = 1.17549435e-38f;
        un[a] = unit;
        uint32_t ql4[4] = {255u, 255u, 255u, 255u}, qh4[4] = {0u, 0u, 0u, 0u};
        for (int k = 0; k < 4; ++k) if (ref[k] != NORI_HIP_NO_CHILD) {
            float ql = floorf((mn[k] - lo) / unit); ql = fminf(fmaxf(ql, 0.f), 255.f);
            while (ql > 0.f && lo + ql * unit > mn[k]) ql -= 1.f;
            float qh = ceilf((mx[k] - lo) / unit); qh = fminf(fmaxf(qh, 0.f), 255.f);
            while (qh < 255.f && lo + qh * unit < mx[k]) qh += 1.f;
            ql4[k] = (uint32_t) ql; qh4[k] = (uint32_t) qh;
        }
        wlo[a] = ql4[0] | (ql4[1] << 8) | (ql4[2] << 16) | (ql4[3] << 24); whi[a] = qh4[0] | (qh4[1] << 8) | (qh4[2] << 16) | (qh4[3] << 24);
    }
    uint4 *o = out + 4 * (size_t) i;
    o[0] = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(un[0]));
    o[1] = make_uint4(wlo[0], wlo[1], wlo[2], whi[0]);
    o[2] = make_uint4(whi[1], whi[2], __float_as_uint(un[1]), __float_as_uint(un[2]));
    o[3] = make_uint4(ref[0], ref[1], ref[2], ref[3]);
}

} // namespace

// nodes4: n four-box nodes (device memory).  *out: n 64-byte quantised nodes, hipMalloc'ed here, owned by the caller.
hipError_t quantize4(hipStream_t stream, const void *nodes4, uint32_t n, void **out) {
    *out = nullptr;
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc(out, (size_t) n * 64);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_quantize4, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, (const float4 *) nodes4, (uint4 *) *out);
    return hipGetLastError();
}

// The order of the key bits decides which axis the radix tree splits at each level.  A plain Morton code alternates x, y, z
// whatever the geometry; on a scene whose primitives are long on one axis (a height field of needles: millimetres wide, a
// tenth of the scene high) a third of the splits then cut along the axis on which every primitive overlaps its neighbours,
// both children keep the parent's extent on the other two axes, and a ray that enters one enters both.  Here the bits are laid
// out by the physical size of the cell they halve, largest first (so axes are split in proportion to the scene's shape), and
// the bits of an axis that cut finer than the typical primitive's extent on that axis (geometric mean over the records) are
// moved behind all the useful ones: they still order the records inside the finest useful cell, but split nothing before that.
static BitPlan plan_bits(const float lo[3], const float hi[3], const long long log_extent[3], uint32_t n) {
    struct Cand { double size; int axis, bit; bool useful; };
    std::vector<Cand> c;
    for (int a = 0; a < 3; ++a) {
        const double ext = (double) hi[a] - (double) lo[a];
        const double typical = std::exp2((double) log_extent[a] / 256.0 / (double) n);      // geometric mean of the records' extents
        for (int k = 0; k < 21; ++k) {
            const double cell = ext > 0 ? ext / std::exp2((double) (k + 1)) : 0.0;         // size of the halves this bit tells apart
            c.push_back({cell, a, 20 - k, cell >= 0.75 * typical && cell > 0});
        }
    }
    std::stable_sort(c.begin(), c.end(), [](const Cand &x, const Cand &y) { if (x.useful != y.useful) return x.useful; return x.size > y.size; });
    BitPlan p;
    for (int i = 0; i < 63; ++i) { p.axis[i] = (unsigned char) c[i].axis; p.bit[i] = (unsigned char) c[i].bit; }
    return p;
}

int build(hipStream_t stream, uint32_t n, const NoriHipPrim *prims_in, const NoriHipShadePrim *shade_in, NoriHipPrim *prims_out,
          NoriHipShadePrim *shade_out, NoriHipNode *nodes_out, uint32_t leaf_max, float big, uint32_t max_depth, bool plain_morton, bool want8, bool align_leaves, Result &res, std::string &err) {
    std::vector<void *> owned; hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { for (void *p : owned) (void) hipFree(p); owned.clear(); if (e0) (void) hipEventDestroy(e0); if (e1) (void) hipEventDestroy(e1); e0 = e1 = nullptr; };
#define LCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); cleanup(); return NORI_HIP_EDEVICE; } } while (0)
    auto alloc = [&](void **p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) owned.push_back(*p); return e; };
    if (n == 0 || leaf_max == 0) { err = "lbvh: nothing to build"; return NORI_HIP_EINVAL; }
    if (n >= 0x7fffffffu) { err = "lbvh: too many records"; return NORI_HIP_EINVAL; }
    const unsigned blocks = (n + 255u) / 256u;
    LCHK(hipEventCreate(&e0)); LCHK(hipEventCreate(&e1));
    float4 *pmin, *pmax; uint32_t *cb; long long *log_extent;
    LCHK(alloc((void **) &pmin, (size_t) n * 16)); LCHK(alloc((void **) &pmax, (size_t) n * 16)); LCHK(alloc((void **) &cb, 8 * 4)); LCHK(alloc((void **) &log_extent, 3 * 8));
    const uint32_t cb_init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};     // [6] = tree depth, [7] = nodes in use
    LCHK(hipMemcpyAsync(cb, cb_init, sizeof(cb_init), hipMemcpyHostToDevice, stream));
    LCHK(hipMemsetAsync(log_extent, 0, 3 * 8, stream));
    LCHK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(k_boxes, dim3(blocks), dim3(256), 0, stream, n, prims_in, shade_in, pmin, pmax, cb, log_extent);
    LCHK(hipGetLastError());
    res = Result();
    if (n <= leaf_max) {
        // one leaf: the root is a leaf reference, no inner node (include/nori_hip.h: NoriHipSceneDesc::root)
        std::vector<uint32_t> ident(n); for (uint32_t i = 0; i < n; ++i) ident[i] = i;
        uint32_t *vals; LCHK(alloc((void **) &vals, (size_t) n * 4));
        LCHK(hipMemcpyAsync(vals, ident.data(), (size_t) n * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_reorder, dim3(blocks), dim3(256), 0, stream, n, (const uint32_t *) vals, prims_in, shade_in, prims_out, shade_out, (const uint32_t *) nullptr, (const uint32_t *) nullptr, (const uint32_t *) nullptr);
        hipLaunchKernelGGL(k_mark_last, dim3(1), dim3(1), 0, stream, prims_out, n - 1);
        LCHK(hipGetLastError());
        std::vector<float4> hmn(n), hmx(n);
        LCHK(hipMemcpyAsync(hmn.data(), pmin, (size_t) n * 16, hipMemcpyDeviceToHost, stream));
        LCHK(hipMemcpyAsync(hmx.data(), pmax, (size_t) n * 16, hipMemcpyDeviceToHost, stream));
        LCHK(hipEventRecord(e1, stream));
        LCHK(hipStreamSynchronize(stream));
        for (int a = 0; a < 3; ++a) { res.root_min[a] = 3.402823466e38f; res.root_max[a] = -3.402823466e38f; }
        for (uint32_t i = 0; i < n; ++i) {
            const float mn[3] = {hmn[i].x, hmn[i].y, hmn[i].z}, mx[3] = {hmx[i].x, hmx[i].y, hmx[i].z};
            for (int a = 0; a < 3; ++a) { res.root_min[a] = std::min(res.root_min[a], mn[a]); res.root_max[a] = std::max(res.root_max[a], mx[a]); }
        }
        res.root = LB_LEAF | 0u; res.n_nodes = 0; res.n_used = 0; res.depth = 0;
        float ms = 0.f; LCHK(hipEventElapsedTime(&ms, e0, e1)); res.build_ms = ms;
        cleanup();
        return NORI_HIP_OK;
    }
    unsigned long long *keys, *keys_s; uint32_t *vals, *vals_s;
    LCHK(alloc((void **) &keys, (size_t) n * 8)); LCHK(alloc((void **) &keys_s, (size_t) n * 8));
    LCHK(alloc((void **) &vals, (size_t) n * 4)); LCHK(alloc((void **) &vals_s, (size_t) n * 4));
    BitPlan plan;
    {
        uint32_t h_b[8]; long long h_le[3];
        LCHK(hipMemcpyAsync(h_b, cb, sizeof(h_b), hipMemcpyDeviceToHost, stream));
        LCHK(hipMemcpyAsync(h_le, log_extent, sizeof(h_le), hipMemcpyDeviceToHost, stream));
        LCHK(hipStreamSynchronize(stream));
        float lo[3], hi[3]; for (int a = 0; a < 3; ++a) { lo[a] = o2f(h_b[a]); hi[a] = o2f(h_b[3 + a]); }
        if (plain_morton) { for (int i = 0; i < 63; ++i) { plan.axis[i] = (unsigned char) (i % 3); plan.bit[i] = (unsigned char) (20 - i / 3); } }
        else plan = plan_bits(lo, hi, h_le, n);
    }
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, stream, n, (const float4 *) pmin, (const float4 *) pmax, (const uint32_t *) cb, big, plan, keys, vals);
    LCHK(hipGetLastError());
    size_t tmp_bytes = 0; void *tmp = nullptr;
    LCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_s, vals, vals_s, (int) n, 0, 64, stream));
    LCHK(alloc(&tmp, tmp_bytes));
    LCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_s, vals, vals_s, (int) n, 0, 64, stream));
    uint32_t *left, *right, *first, *last, *parent_inner, *parent_leaf, *arrived; float *nbox;
    const size_t ni = (size_t) n - 1;
    LCHK(alloc((void **) &left, ni * 4)); LCHK(alloc((void **) &right, ni * 4)); LCHK(alloc((void **) &first, ni * 4)); LCHK(alloc((void **) &last, ni * 4));
    LCHK(alloc((void **) &parent_inner, ni * 4)); LCHK(alloc((void **) &parent_leaf, (size_t) n * 4)); LCHK(alloc((void **) &arrived, ni * 4));
    LCHK(alloc((void **) &nbox, ni * 32));
    LCHK(hipMemsetAsync(arrived, 0, ni * 4, stream));
    hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(256), 0, stream, (int) n, (const unsigned long long *) keys_s, left, right, first, last, parent_inner, parent_leaf);
    LCHK(hipGetLastError());
    hipLaunchKernelGGL(k_depth, dim3(blocks), dim3(256), 0, stream, n, (const uint32_t *) parent_inner, (const uint32_t *) parent_leaf, cb + 6);
    LCHK(hipGetLastError());
    uint32_t h_cb[8];
    LCHK(hipMemcpyAsync(h_cb, cb, sizeof(h_cb), hipMemcpyDeviceToHost, stream));
    LCHK(hipStreamSynchronize(stream));
    res.depth = h_cb[6];
    if (res.depth > max_depth) {
        err = "lbvh: tree depth " + std::to_string(res.depth) + " exceeds the traversal stack (" + std::to_string(max_depth) + ")";
        cleanup();
        return NORI_HIP_EINVAL;
    }
    for (uint32_t pass = 1; pass <= res.depth + 1u; ++pass)
        hipLaunchKernelGGL(k_refit_pass, dim3(blocks), dim3(256), 0, stream, (uint32_t) ni, pass, (const uint32_t *) vals_s, (const float4 *) pmin, (const float4 *) pmax, (const uint32_t *) left,
                           (const uint32_t *) right, nbox, arrived);
    LCHK(hipGetLastError());
    NoriHipNode *nodes_tmp; uint32_t *used, *new_index, *leaf_start, *leaf_end, *leaf_idx = nullptr;
    LCHK(alloc((void **) &nodes_tmp, ni * sizeof(NoriHipNode))); LCHK(alloc((void **) &used, ni * 4)); LCHK(alloc((void **) &new_index, ni * 4));
    LCHK(alloc((void **) &leaf_start, (size_t) n * 4)); LCHK(alloc((void **) &leaf_end, (size_t) n * 4));
    LCHK(hipMemsetAsync(leaf_start, 0, (size_t) n * 4, stream)); LCHK(hipMemsetAsync(leaf_end, 0, (size_t) n * 4, stream));
    hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(256), 0, stream, n, leaf_max, (const uint32_t *) vals_s, (const float4 *) pmin, (const float4 *) pmax, (const uint32_t *) left,
                       (const uint32_t *) right, (const uint32_t *) first, (const uint32_t *) last, (const float *) nbox, nodes_tmp, leaf_start, leaf_end, used);
    LCHK(hipGetLastError());
    size_t scan_bytes = 0; void *scan_tmp = nullptr;
    LCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, leaf_start, new_index, (int) n, stream));       // (the larger of the scans made with this scratch)
    LCHK(alloc(&scan_tmp, scan_bytes));
    NoriHipPrim *p_dst = prims_out; NoriHipShadePrim *s_dst = shade_out;
    if (align_leaves && leaf_max <= 2u) {
        LCHK(alloc((void **) &leaf_idx, (size_t) n * 4));
        LCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, leaf_start, leaf_idx, (int) n, stream));
        uint32_t h2[2];
        LCHK(hipMemcpyAsync(&h2[0], leaf_start + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        LCHK(hipMemcpyAsync(&h2[1], leaf_idx + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        LCHK(hipStreamSynchronize(stream));
        const uint64_t n_leaves = (uint64_t) h2[0] + h2[1], n_pos = ((n_leaves + 2u) / 3u) * 8u;
        if (n_pos >= 0x7fffffffull) { err = "lbvh: too many records to place leaf by leaf"; cleanup(); return NORI_HIP_EINVAL; }
        void *pa = nullptr, *sa = nullptr;
        LCHK(hipMalloc(&pa, (size_t) n_pos * sizeof(NoriHipPrim))); res.prims_al = pa;                           // owned by the caller from here on
        LCHK(hipMalloc(&sa, (size_t) n_pos * sizeof(NoriHipShadePrim))); res.shade_al = sa;
        LCHK(hipMemsetAsync(pa, 0, (size_t) n_pos * sizeof(NoriHipPrim), stream)); LCHK(hipMemsetAsync(sa, 0, (size_t) n_pos * sizeof(NoriHipShadePrim), stream));
        res.n_positions = (uint32_t) n_pos;
        p_dst = (NoriHipPrim *) pa; s_dst = (NoriHipShadePrim *) sa;
        hipLaunchKernelGGL(k_place_refs, dim3(blocks), dim3(256), 0, stream, (uint32_t) ni, (const uint32_t *) used, nodes_tmp, (const uint32_t *) leaf_start, (const uint32_t *) leaf_idx);
        LCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_reorder, dim3(blocks), dim3(256), 0, stream, n, (const uint32_t *) vals_s, prims_in, shade_in, p_dst, s_dst, (const uint32_t *) leaf_start, (const uint32_t *) leaf_idx,
                       (const uint32_t *) leaf_end);
    LCHK(hipGetLastError());
    LCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, used, new_index, (int) ni, stream));
    hipLaunchKernelGGL(k_compact, dim3(blocks), dim3(256), 0, stream, (uint32_t) ni, (const uint32_t *) used, (const uint32_t *) new_index, (const NoriHipNode *) nodes_tmp, nodes_out);
    LCHK(hipGetLastError());
    // the same tree as four-box nodes, level by level (on the emitted nodes before they are packed: `nodes_tmp`, root 0)
    {
        uint32_t *qa, *qb, *cnt, *off; float4 *wide_tmp;
        LCHK(alloc((void **) &qa, ni * 4)); LCHK(alloc((void **) &qb, ni * 4)); LCHK(alloc((void **) &cnt, ni * 4)); LCHK(alloc((void **) &off, ni * 4));
        LCHK(alloc((void **) &wide_tmp, ni * 128));                               // at most one four-box node per binary node
        const uint32_t zero = 0u;
        LCHK(hipMemcpyAsync(qa, &zero, 4, hipMemcpyHostToDevice, stream));
        uint32_t n_level = 1, base = 0, levels = 0;
        while (n_level > 0) {
            const unsigned lb = (n_level + 255u) / 256u;
            hipLaunchKernelGGL(k_wide_count<4>, dim3(lb), dim3(256), 0, stream, n_level, (const uint32_t *) qa, (const NoriHipNode *) nodes_tmp, cnt);
            LCHK(hipGetLastError());
            LCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, cnt, off, (int) n_level, stream));
            hipLaunchKernelGGL(k_wide_emit, dim3(lb), dim3(256), 0, stream, n_level, (const uint32_t *) qa, (const NoriHipNode *) nodes_tmp, (const uint32_t *) off, base, base + n_level, qb, wide_tmp);
            LCHK(hipGetLastError());
            uint32_t h2[2];
            LCHK(hipMemcpyAsync(&h2[0], cnt + (n_level - 1), 4, hipMemcpyDeviceToHost, stream));
            LCHK(hipMemcpyAsync(&h2[1], off + (n_level - 1), 4, hipMemcpyDeviceToHost, stream));
            LCHK(hipStreamSynchronize(stream));
            base += n_level; n_level = h2[0] + h2[1]; ++levels;
            std::swap(qa, qb);
            if (base + (size_t) n_level > ni) { err = "lbvh: four-box conversion overran its node bound"; cleanup(); return NORI_HIP_EDEVICE; }
        }
        res.n_nodes4 = base; res.depth4 = levels;
        void *n4 = nullptr;
        LCHK(hipMalloc(&n4, (size_t) std::max<uint32_t>(res.n_nodes4, 1u) * 128));       // owned by the caller from here on
        res.nodes4 = n4;
        LCHK(hipMemcpyAsync(n4, wide_tmp, (size_t) res.n_nodes4 * 128, hipMemcpyDeviceToDevice, stream));
        if (want8) {
            // ... and as quantised eight-box nodes, the same way (the level queues, the scan's scratch and wide_tmp are reused)
            uint32_t *na, *nb, *smax;
            LCHK(alloc((void **) &na, ni * 4)); LCHK(alloc((void **) &nb, ni * 4)); LCHK(alloc((void **) &smax, 4));
            LCHK(hipMemsetAsync(qa, 0, 4, stream)); LCHK(hipMemsetAsync(na, 0, 4, stream)); LCHK(hipMemsetAsync(smax, 0, 4, stream));
            n_level = 1; base = 0; levels = 0;
            while (n_level > 0) {
                const unsigned lb = (n_level + 255u) / 256u;
                hipLaunchKernelGGL(k_wide_count<8>, dim3(lb), dim3(256), 0, stream, n_level, (const uint32_t *) qa, (const NoriHipNode *) nodes_tmp, cnt);
                LCHK(hipGetLastError());
                LCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, cnt, off, (int) n_level, stream));
                hipLaunchKernelGGL(k_wide_emit8, dim3(lb), dim3(256), 0, stream, n_level, (const uint32_t *) qa, (const uint32_t *) na, (const NoriHipNode *) nodes_tmp, (const uint32_t *) off, base,
                                   base + n_level, qb, nb, reinterpret_cast<uint4 *>(wide_tmp), smax);
                LCHK(hipGetLastError());
                uint32_t h2[2];
                LCHK(hipMemcpyAsync(&h2[0], cnt + (n_level - 1), 4, hipMemcpyDeviceToHost, stream));
                LCHK(hipMemcpyAsync(&h2[1], off + (n_level - 1), 4, hipMemcpyDeviceToHost, stream));
                LCHK(hipStreamSynchronize(stream));
                base += n_level; n_level = h2[0] + h2[1]; ++levels;
                std::swap(qa, qb); std::swap(na, nb);
                if (base + (size_t) n_level > ni) { err = "lbvh: eight-box conversion overran its node bound"; cleanup(); return NORI_HIP_EDEVICE; }
            }
            res.n_nodes8 = base; res.depth8 = levels;
            LCHK(hipMemcpyAsync(&res.stack8, smax, 4, hipMemcpyDeviceToHost, stream));
            void *n8 = nullptr;
            LCHK(hipMalloc(&n8, (size_t) std::max<uint32_t>(res.n_nodes8, 1u) * 128));   // owned by the caller from here on
            res.nodes8 = n8;
            LCHK(hipMemcpyAsync(n8, wide_tmp, (size_t) res.n_nodes8 * 128, hipMemcpyDeviceToDevice, stream));
        }
    }
    uint32_t h_last[2];
    LCHK(hipMemcpyAsync(&h_last[0], used + (ni - 1), 4, hipMemcpyDeviceToHost, stream));
    LCHK(hipMemcpyAsync(&h_last[1], new_index + (ni - 1), 4, hipMemcpyDeviceToHost, stream));
    float h_root[8];
    LCHK(hipMemcpyAsync(h_root, nbox, 32, hipMemcpyDeviceToHost, stream));
    LCHK(hipMemcpyAsync(h_cb, cb, sizeof(h_cb), hipMemcpyDeviceToHost, stream));
    LCHK(hipEventRecord(e1, stream));
    LCHK(hipStreamSynchronize(stream));
    for (int a = 0; a < 3; ++a) { res.root_min[a] = h_root[a]; res.root_max[a] = h_root[4 + a]; }
    res.root = 0u; res.n_used = h_last[0] + h_last[1]; res.n_nodes = res.n_used;
    float ms = 0.f; LCHK(hipEventElapsedTime(&ms, e0, e1)); res.build_ms = ms;
    cleanup();
    return NORI_HIP_OK;
#undef LCHK
}

// radix sort of (32-bit key, 32-bit value) pairs on `stream` (hipCUB); the render loop orders its rays with it
size_t sort_pairs_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void) hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr, (const uint32_t *) nullptr, (uint32_t *) nullptr, (int) n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}
hipError_t sort_pairs(hipStream_t stream, void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, int end_bit, int begin_bit) {
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int) n, begin_bit, end_bit, stream);
}

} // namespace lbvh
