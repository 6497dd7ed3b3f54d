// SAH BVH build with the reference's topology and leaf order, and the flattening into the C ABI's tables.
//
// Reference: src/bvh.cpp:54-306 (binned SAH over 16 bins on the largest axis while >= 32 primitives, exact sweep
// SAH on all three axes below; costs 1/1), :329-382 (build, compaction), include/nori/bvh.h:105-109 (findShape).
//
// What is kept: every decision the reference makes (bin index arithmetic, cost comparisons, tie behaviour of
// std::sort, which box a node stores) so that the node sequence and the index array are bit-identical to the
// reference built on one TBB thread (the multi-threaded reference build is itself nondeterministic).
// What is different: the tree is emitted directly in its final pre-order numbering (no 2N scratch array and
// compaction pass), per-primitive centroids/boxes are tabulated once in SoA form instead of being re-derived
// through findShape + virtual calls at every comparison, and disjoint subtrees are built concurrently — each into
// its own node vector that is spliced in afterwards — which cannot change the result because a subtree only
// touches its own index range.
#include "scene.h"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <future>
#include <thread>

namespace nori {

BVH::~BVH() { if (m_owns) for (auto s : m_shapes) delete s; }

void BVH::addShape(Shape *shape) {
    m_shapes.push_back(shape);
    m_shapeOffset.push_back(m_shapeOffset.back() + shape->getPrimitiveCount());
    m_bbox.expandBy(shape->getBoundingBox());
}
uint32_t BVH::findShape(uint32_t &idx) const {
    auto it = std::lower_bound(m_shapeOffset.begin(), m_shapeOffset.end(), idx + 1) - 1;
    idx -= *it;
    return (uint32_t) (it - m_shapeOffset.begin());
}

namespace {
typedef std::vector<BVH::Node32> NodeVec;
enum { SERIAL_THRESHOLD = 32, BIN_COUNT = 16, SPAWN_MIN = 1 << 15 };

struct Builder {
    std::vector<float> c[3];                 // centroids, one array per axis
    std::vector<BoundingBox3f> box;
    uint32_t *indices = nullptr; uint32_t *temp = nullptr;

    static void store(BVH::Node32 &n, const BoundingBox3f &b) { for (int i = 0; i < 3; ++i) { n.bmin[i] = b.min[i]; n.bmax[i] = b.max[i]; } }

    // exact SAH sweep (execute_serially, src/bvh.cpp:236-305)
    uint32_t sweep(NodeVec &out, uint32_t s, uint32_t e) {
        uint32_t *first = indices + s, *last = indices + e;
        const uint32_t size = e - s;
        float *leftAreas = reinterpret_cast<float *>(temp + s);
        BoundingBox3f nodeBox;
        float bestCost = (float) 1 * size; int64_t bestIndex = -1, bestAxis = -1;
        for (int axis = 0; axis < 3; ++axis) {
            const float *ca = c[axis].data();
            std::sort(first, last, [ca](uint32_t a, uint32_t b) { return ca[a] < ca[b]; });
            BoundingBox3f acc;
            for (uint32_t i = 0; i < size; ++i) { acc.expandBy(box[first[i]]); leftAreas[i] = acc.getSurfaceArea(); }
            if (axis == 0) nodeBox = acc;
            acc.reset();
            const float triFactor = 1 / nodeBox.getSurfaceArea();
            for (uint32_t i = size - 1; i >= 1; --i) {
                acc.expandBy(box[first[i]]);
                const float leftArea = leftAreas[i - 1], rightArea = acc.getSurfaceArea();
                const uint32_t primsLeft = i, primsRight = size - i;
                const float cost = 2.0f * 1 + triFactor * (primsLeft * leftArea + primsRight * rightArea);
                if (cost < bestCost) { bestCost = cost; bestIndex = i; bestAxis = axis; }
            }
        }
        const uint32_t me = (uint32_t) out.size();
        out.emplace_back();
        if (bestIndex == -1) { out[me].w0 = 1u | (size << 1); out[me].w1 = s; store(out[me], nodeBox); return me; }
        const float *ca = c[bestAxis].data();
        std::sort(first, last, [ca](uint32_t a, uint32_t b) { return ca[a] < ca[b]; });
        const uint32_t leftCount = (uint32_t) bestIndex;
        sweep(out, s, s + leftCount);
        const uint32_t r = sweep(out, s + leftCount, e);
        out[me].w0 = ((uint32_t) bestAxis) << 1; out[me].w1 = r; store(out[me], nodeBox);
        return me;
    }

    // binned SAH (BVHBuildTask::execute, src/bvh.cpp:100-233); chunks of the reference's parallel loops taken in order
    uint32_t split(NodeVec &out, const BoundingBox3f &nodeBox, uint32_t s, uint32_t e, int threads) {
        const uint32_t size = e - s;
        if (size < SERIAL_THRESHOLD) return sweep(out, s, e);
        uint32_t *first = indices + s;
        const int axis = nodeBox.getLargestAxis();
        const float lo = nodeBox.min[axis], hi = nodeBox.max[axis], invBin = BIN_COUNT / (hi - lo);
        const float *ca = c[axis].data();
        uint32_t counts[BIN_COUNT] = {0}; BoundingBox3f bins[BIN_COUNT];
        for (uint32_t i = 0; i < size; ++i) {
            const uint32_t f = first[i];
            const int b = std::min(std::max((int) ((ca[f] - lo) * invBin), 0), (int) BIN_COUNT - 1);
            counts[b]++; bins[b].expandBy(box[f]);
        }
        BoundingBox3f leftBox[BIN_COUNT]; leftBox[0] = bins[0];
        for (int i = 1; i < BIN_COUNT; ++i) { counts[i] += counts[i - 1]; leftBox[i] = leftBox[i - 1]; leftBox[i].expandBy(bins[i]); }
        BoundingBox3f rightBox = bins[BIN_COUNT - 1], bestRight;
        int64_t bestIndex = -1; float bestCost = (float) 1 * size;
        const float triFactor = (float) 1 / nodeBox.getSurfaceArea();
        for (int i = BIN_COUNT - 2; i >= 0; --i) {
            const uint32_t primsLeft = counts[i], primsRight = size - counts[i];
            const float cost = 2.0f * 1 + triFactor * (primsLeft * leftBox[i].getSurfaceArea() + primsRight * rightBox.getSurfaceArea());
            if (cost < bestCost) { bestCost = cost; bestIndex = i; bestRight = rightBox; }
            rightBox.expandBy(bins[i]);
        }
        if (bestIndex == -1) return sweep(out, s, e);
        const uint32_t leftCount = counts[bestIndex];
        uint32_t *tmp = temp + s; uint32_t il = 0, ir = leftCount;
        for (uint32_t i = 0; i < size; ++i) {                 // stable: the partition test uses the UNclamped bin index
            const uint32_t f = first[i];
            if ((int) ((ca[f] - lo) * invBin) <= bestIndex) tmp[il++] = f; else tmp[ir++] = f;
        }
        std::memcpy(first, tmp, size * sizeof(uint32_t));

        const uint32_t me = (uint32_t) out.size();
        out.emplace_back();
        const BoundingBox3f lb = leftBox[bestIndex], rb = bestRight;
        uint32_t r;
        if (threads >= 2 && size >= SPAWN_MIN) {
            // right subtree on another thread, into its own vector; indices in it are local and shifted when spliced
            const int tr = threads / 2, tl = threads - tr;
            NodeVec rightNodes;
            auto fut = std::async(std::launch::async, [&]() { split(rightNodes, rb, s + leftCount, e, tr); });
            split(out, lb, s, s + leftCount, tl);
            fut.get();
            r = (uint32_t) out.size();
            out.reserve(out.size() + rightNodes.size());
            for (BVH::Node32 n : rightNodes) { if (!(n.w0 & 1u)) n.w1 += r; out.push_back(n); }
        } else {
            split(out, lb, s, s + leftCount, 1);
            r = split(out, rb, s + leftCount, e, 1);
        }
        out[me].w0 = ((uint32_t) axis) << 1; out[me].w1 = r; store(out[me], nodeBox);
        return me;
    }
};
} // namespace

void BVH::build(int threads) {
    const uint32_t n = getPrimitiveCount();
    m_nodes.clear(); m_indices.clear();
    if (n == 0) return;
    auto t0 = std::chrono::steady_clock::now();
    if (threads <= 0) { threads = (int) std::thread::hardware_concurrency(); if (threads <= 0) threads = 1; if (threads > 32) threads = 32; }
    Builder b;
    for (int a = 0; a < 3; ++a) b.c[a].resize(n);
    b.box.resize(n);
    {   // tabulate per-primitive centroid and box (BVH::getCentroid / getBoundingBox, include/nori/bvh.h:112-121)
        auto work = [&](uint32_t lo, uint32_t hi) {
            size_t s = std::upper_bound(m_shapeOffset.begin(), m_shapeOffset.end(), lo) - m_shapeOffset.begin() - 1;
            for (uint32_t i = lo; i < hi; ++i) {
                while (i >= m_shapeOffset[s + 1]) ++s;
                const Shape *sh = m_shapes[s]; const uint32_t local = i - m_shapeOffset[s];
                Point3f cen = sh->getCentroid(local);
                b.c[0][i] = cen.x; b.c[1][i] = cen.y; b.c[2][i] = cen.z; b.box[i] = sh->getBoundingBox(local);
            }
        };
        if (n < 100000 || threads == 1) work(0, n);
        else { std::vector<std::thread> th; for (int t = 0; t < threads; ++t) th.emplace_back(work, (uint32_t) ((uint64_t) n * t / threads), (uint32_t) ((uint64_t) n * (t + 1) / threads)); for (auto &t : th) t.join(); }
    }
    // On-disk cache of the tree ($NORI_HIP_BVH_CACHE = a directory): the reference-order SAH build is a function
    // of the per-primitive boxes and centroids just tabulated (and of the scene box), so their 64-bit hash names the entry; a hit
    // replaces 14 s of building for ten million triangles by a read.  An entry is used only if its key, counts and sizes agree and
    // every index is in range; nori_hip_upload_scene checks the structure of whatever tree it is given before a kernel walks it.
    std::string cachePath; uint64_t cacheKey = 0;
    if (const char *cdir = getenv("NORI_HIP_BVH_CACHE")) if (*cdir) {
        // (hashed in fixed chunks of 65536 primitives, combined in order: the key does not depend on the number of threads)
        const uint32_t n_chunks = (n + 65535u) / 65536u;
        std::vector<uint64_t> part((size_t) n_chunks, 0);
        auto hashChunks = [&](uint32_t c0, uint32_t c1) {
            for (uint32_t c = c0; c < c1; ++c) {
                uint64_t h = 0x9E3779B97F4A7C15ull ^ c;
                auto mix = [&](const void *p, size_t bytes) { const uint32_t *w = (const uint32_t *) p; for (size_t k = 0; k < bytes / 4; ++k) { h ^= w[k]; h *= 0x100000001B3ull; h ^= h >> 29; } };
                const uint32_t hi = std::min<uint64_t>(n, ((uint64_t) c + 1) * 65536u);
                for (uint32_t i = c * 65536u; i < hi; ++i) { mix(&b.box[i], sizeof(BoundingBox3f)); const float c3[3] = {b.c[0][i], b.c[1][i], b.c[2][i]}; mix(c3, 12); }
                part[c] = h;
            }
        };
        { std::vector<std::thread> th; for (int t = 0; t < threads; ++t) th.emplace_back(hashChunks, (uint32_t) ((uint64_t) n_chunks * t / threads), (uint32_t) ((uint64_t) n_chunks * (t + 1) / threads)); for (auto &t : th) t.join(); }
        uint64_t h = 0xCBF29CE484222325ull ^ n;
        for (uint64_t v : part) { h ^= v; h *= 0x100000001B3ull; h ^= h >> 31; }
        const float bb[6] = {m_bbox.min[0], m_bbox.min[1], m_bbox.min[2], m_bbox.max[0], m_bbox.max[1], m_bbox.max[2]};
        for (float f : bb) { uint32_t w; std::memcpy(&w, &f, 4); h ^= w; h *= 0x100000001B3ull; h ^= h >> 31; }
        cacheKey = h;
        char name[64]; snprintf(name, sizeof(name), "/bvh-%016llx.noricache", (unsigned long long) h);
        cachePath = std::string(cdir) + name;
        struct Hdr { char magic[8]; uint64_t key; uint32_t n, nNodes; } hd;
        FILE *f = fopen(cachePath.c_str(), "rb");
        if (f) {
            bool ok = fread(&hd, sizeof(hd), 1, f) == 1 && std::memcmp(hd.magic, "NORIBVH1", 8) == 0 && hd.key == h && hd.n == n && hd.nNodes >= 1 && hd.nNodes <= 2u * n;
            if (ok) {
                m_nodes.resize(hd.nNodes); m_indices.resize(n);
                ok = fread(m_nodes.data(), sizeof(Node32), hd.nNodes, f) == hd.nNodes && fread(m_indices.data(), 4, n, f) == n && fgetc(f) == EOF;
                for (uint32_t i = 0; ok && i < n; ++i) ok = m_indices[i] < n;
            }
            fclose(f);
            if (ok) { m_buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); return; }
            m_nodes.clear(); m_indices.clear();
        }
    }
    m_indices.resize(n);
    for (uint32_t i = 0; i < n; ++i) m_indices[i] = i;
    std::vector<uint32_t> temp(n);
    b.indices = m_indices.data(); b.temp = temp.data();
    m_nodes.reserve((size_t) n);
    b.split(m_nodes, m_bbox, 0, n, threads);
    m_nodes.shrink_to_fit();
    m_buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!cachePath.empty()) {
        struct Hdr { char magic[8]; uint64_t key; uint32_t n, nNodes; } hd;
        std::memcpy(hd.magic, "NORIBVH1", 8); hd.key = cacheKey; hd.n = n; hd.nNodes = (uint32_t) m_nodes.size();
        const std::string tmp = cachePath + ".tmp";
        if (FILE *f = fopen(tmp.c_str(), "wb")) {                      // (a directory that cannot be written: no entry, no error)
            const bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1 && fwrite(m_nodes.data(), sizeof(Node32), m_nodes.size(), f) == m_nodes.size() && fwrite(m_indices.data(), 4, n, f) == n;
            if (fclose(f) == 0 && ok) std::rename(tmp.c_str(), cachePath.c_str()); else std::remove(tmp.c_str());
        }
    }
}

// ---- flattening -------------------------------------------------------------------------------------------
void flattenScene(const Scene *scene, FlatScene &out) {
    const BVH *bvh = scene->getBVH();
    const auto &nodes = bvh->getNodes(); const auto &indices = bvh->getIndices();
    const auto &shapes = bvh->getShapes(); const auto &offs = bvh->getShapeOffsets();
    const uint32_t nPrims = bvh->getPrimitiveCount();
    if (nPrims == 0) throw NoriException("flattenScene: the scene has no primitives");
    if (shapes.size() >= (1u << 24)) throw NoriException("flattenScene: too many shapes (24-bit shape index)");
    out = FlatScene();

    // shapes, emitters, area CDFs
    const auto &lights = scene->getLights();
    out.shapes.resize(shapes.size());
    // images: one table entry per distinct plugin object; returns the 1-based index the shape records hold
    std::vector<const ImageData *> seen;
    auto addImage = [&](const ImageData *img) -> uint32_t {
        if (!img) return 0u;
        for (size_t k = 0; k < seen.size(); ++k) if (seen[k] == img) return (uint32_t) k + 1;
        NoriHipTexture t = img->meta; t.offset = out.tex_data.size();
        out.tex_data.insert(out.tex_data.end(), img->texels.begin(), img->texels.end());
        if (out.textures.size() >= 65535) throw NoriException("flattenScene: too many images (16-bit image index)");
        out.textures.push_back(t); seen.push_back(img);
        return (uint32_t) out.textures.size();
    };
    for (size_t s = 0; s < shapes.size(); ++s) {
        NoriHipShape &d = out.shapes[s]; std::memset(&d, 0, sizeof(d));
        const Shape *sh = shapes[s];
        d.type = sh->isSphere() ? NORI_HIP_SHAPE_SPHERE : NORI_HIP_SHAPE_MESH;
        d.alpha = 0.1f; d.int_ior = 1.5046f; d.ext_ior = 1.000277f; d.ks = 0.5f;
        sh->getBSDF()->describe(d);
        d.emitter = -1;
        d.albedo_tex = (uint16_t) addImage(sh->getBSDF()->albedoImage());
        d.normal_tex = (uint16_t) addImage(sh->getNormalModifier() ? sh->getNormalModifier()->image() : nullptr);
        if (sh->isSphere()) { const Sphere *sp = static_cast<const Sphere *>(sh); Point3f c = sp->getCenter(); d.center[0] = c.x; d.center[1] = c.y; d.center[2] = c.z; d.radius = sp->getRadius(); }
        else { const Mesh *m = static_cast<const Mesh *>(sh); d.flags = (m->getVertexNormals().empty() ? 0u : 1u) | (m->getVertexTexCoords().empty() ? 0u : 2u); }
    }
    auto shadePrim = [](const Mesh *m, uint32_t tri, NoriHipShadePrim &sp) {
        std::memset(&sp, 0, sizeof(sp));
        const auto &F = m->getIndices(); const auto &V = m->getVertexPositions(); const auto &N = m->getVertexNormals(); const auto &UV = m->getVertexTexCoords();
        float *P[3] = {sp.p0, sp.p1, sp.p2}, *NN[3] = {sp.n0, sp.n1, sp.n2}, *T[3] = {sp.uv0, sp.uv1, sp.uv2};
        for (int k = 0; k < 3; ++k) {
            uint32_t v = F[3 * tri + k];
            for (int a = 0; a < 3; ++a) P[k][a] = V[3 * v + a];
            if (!N.empty()) for (int a = 0; a < 3; ++a) NN[k][a] = N[3 * v + a];
            if (!UV.empty()) for (int a = 0; a < 2; ++a) T[k][a] = UV[2 * v + a];
        }
    };
    out.emitters.resize(lights.size());
    for (size_t e = 0; e < lights.size(); ++e) {
        NoriHipEmitter &d = out.emitters[e]; std::memset(&d, 0, sizeof(d));
        const Emitter *em = lights[e];
        if (!em->getShape()) throw NoriException("There is no shape attached to this Area light!");
        Color3f r = em->getRadiance(); d.radiance[0] = r.x; d.radiance[1] = r.y; d.radiance[2] = r.z;
        size_t si = std::find(shapes.begin(), shapes.end(), em->getShape()) - shapes.begin();
        if (si == shapes.size()) throw NoriException("flattenScene: emitter is attached to a shape that is not part of the scene");
        d.shape = (int32_t) si; out.shapes[si].emitter = (int32_t) e;
        if (!shapes[si]->isSphere()) {
            const Mesh *m = static_cast<const Mesh *>(shapes[si]);
            d.cdf_offset = (uint32_t) out.cdf.size(); d.n_tris = m->getPrimitiveCount(); d.light_prim_offset = (uint32_t) out.light_prims.size();
            d.area_pdf = m->getAreaNormalization();
            out.cdf.insert(out.cdf.end(), m->getAreaCdf().begin(), m->getAreaCdf().end());
            for (uint32_t t = 0; t < d.n_tris; ++t) { out.light_prims.emplace_back(); shadePrim(m, t, out.light_prims.back()); }
        }
    }

    // primitive records in leaf (m_indices) order; in scene order when the tree is left to the device (no host build)
    const bool built = bvh->isBuilt();
    out.prims.resize(nPrims); out.shade.resize(nPrims);
    for (uint32_t pos = 0; pos < nPrims; ++pos) {
        const uint32_t global = built ? indices[pos] : pos;
        uint32_t local = global; const uint32_t s = bvh->findShape(local);
        NoriHipPrim &p = out.prims[pos]; std::memset(&p, 0, sizeof(p));
        p.prim = global; p.shape_flags = s;
        if (shapes[s]->isSphere()) {
            const Sphere *sp = static_cast<const Sphere *>(shapes[s]); Point3f c = sp->getCenter();
            p.a[0] = c.x; p.a[1] = c.y; p.a[2] = c.z; p.e1[0] = sp->getRadius(); p.shape_flags |= 0x40000000u;
            std::memset(&out.shade[pos], 0, sizeof(NoriHipShadePrim));
        } else {
            const Mesh *m = static_cast<const Mesh *>(shapes[s]);
            shadePrim(m, local, out.shade[pos]);
            const NoriHipShadePrim &q = out.shade[pos];
            for (int a = 0; a < 3; ++a) { p.a[a] = q.p0[a]; p.e1[a] = q.p1[a] - q.p0[a]; p.e2[a] = q.p2[a] - q.p0[a]; }
        }
    }
    (void) offs;

    // 32-byte pre-order nodes -> 64-byte two-box inner nodes; leaves live on as child references + record flags
    std::vector<uint32_t> innerId(nodes.size(), 0); uint32_t nInner = 0;
    for (size_t i = 0; i < nodes.size(); ++i) if (!(nodes[i].w0 & 1u)) innerId[i] = nInner++;
    auto childRef = [&](uint32_t idx, uint32_t &count) -> uint32_t {
        const BVH::Node32 &n = nodes[idx];
        if (n.w0 & 1u) { count = n.w0 >> 1; return NORI_HIP_LEAF_BIT | n.w1; }
        count = 0; return innerId[idx];
    };
    for (const BVH::Node32 &n : nodes) if (n.w0 & 1u) { uint32_t sz = n.w0 >> 1; if (sz == 0) throw NoriException("flattenScene: empty leaf"); out.prims[n.w1 + sz - 1].shape_flags |= 0x80000000u; }
    out.nodes.resize(nInner);
    for (size_t i = 0; i < nodes.size(); ++i) {
        const BVH::Node32 &n = nodes[i];
        if (n.w0 & 1u) continue;
        NoriHipNode &w = out.nodes[innerId[i]];
        const BVH::Node32 &l = nodes[i + 1], &r = nodes[n.w1];
        for (int a = 0; a < 3; ++a) { w.lmin[a] = l.bmin[a]; w.lmax[a] = l.bmax[a]; w.rmin[a] = r.bmin[a]; w.rmax[a] = r.bmax[a]; }
        w.left = childRef((uint32_t) i + 1, w.lcount); w.right = childRef(n.w1, w.rcount);
    }
    NoriHipSceneDesc &d = out.desc; std::memset(&d, 0, sizeof(d));
    if (built) {
        uint32_t rootCount; d.root = childRef(0, rootCount);
        for (int a = 0; a < 3; ++a) { d.root_min[a] = nodes[0].bmin[a]; d.root_max[a] = nodes[0].bmax[a]; }
    } else {
        d.root = NORI_HIP_NO_CHILD;                          // for nori_hip_upload_scene_build
        for (int a = 0; a < 3; ++a) { d.root_min[a] = bvh->getBoundingBox().min[a]; d.root_max[a] = bvh->getBoundingBox().max[a]; }
    }
    d.n_nodes = (uint32_t) out.nodes.size(); d.nodes = out.nodes.empty() ? nullptr : out.nodes.data();
    d.n_prims = nPrims; d.prims = out.prims.data(); d.shade_prims = out.shade.data();
    d.n_shapes = (uint32_t) out.shapes.size(); d.shapes = out.shapes.data();
    d.n_emitters = (uint32_t) out.emitters.size(); d.emitters = out.emitters.empty() ? nullptr : out.emitters.data();
    d.n_cdf = (uint32_t) out.cdf.size(); d.cdf = out.cdf.empty() ? nullptr : out.cdf.data();
    d.n_light_prims = (uint32_t) out.light_prims.size(); d.light_prims = out.light_prims.empty() ? nullptr : out.light_prims.data();
    d.n_textures = (uint32_t) out.textures.size(); d.textures = out.textures.empty() ? nullptr : out.textures.data();
    d.n_tex_bytes = out.tex_data.size(); d.tex_data = out.tex_data.empty() ? nullptr : out.tex_data.data();
    scene->getCamera()->describe(d.camera);
    if (const EnvironmentMap *env = scene->getEnvMap()) {
        env->describe(d.env);
        d.env.texture = addImage(env->image());
        d.n_textures = (uint32_t) out.textures.size(); d.textures = out.textures.data();
        d.n_tex_bytes = out.tex_data.size(); d.tex_data = out.tex_data.data();
    }
}

} // namespace nori
