// ORACLE — TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference).
//
// C API over the REFERENCE's own, unmodified sources (compiled where they lie under /root/reference by
// oracle/Makefile target _ref), used to pin the CPU restatement (oracle/*.hpp) and to generate tests/golden/.
//
// One reference file on the path cannot be built here: src/bvh.cpp needs the TBB runtime library
// (tbb::task, parallel_reduce, parallel_for), and ext/tbb only builds through its own cmake system with a
// generated version_string.ver — treated as unbuildable (DESIGN.md "Oracle").  nori::BVH's member functions
// are therefore defined in this file on top of the restated builder (oracle/obvh.hpp); the traversal loop
// below calls the reference's own TBoundingBox::rayIntersect (include/nori/bbox.h:336-363),
// Shape::rayIntersect (src/mesh.cpp:84-121, src/sphere.cpp:42-113) and setHitInformation.
// src/block.cpp is left out for the same reason (tbb::mutex's inline lock() references
// tbb::internal::handle_perror from the TBB runtime); the film semantics are pinned by SURVEY §8 A13 and the
// reference's recorded cbox image mean instead.  Without ImageBlock the reference's Independent sampler cannot
// be re-seeded (its only seeding entry is prepare(const ImageBlock&)), so SeededSampler below wraps the
// reference's pcg32 with the same next2D expression; it is checked against a factory-made "independent"
// sampler in its default state (ref_sampler_check).  Everything else — parser, OBJ loader, camera, BSDFs,
// emitters, warps, PathMisIntegrator::Li — is reference code.
#include <nori/bvh.h>
#include <nori/scene.h>
#include <nori/parser.h>
#include <nori/camera.h>
#include <nori/sampler.h>
#include <nori/integrator.h>
#include <nori/bsdf.h>
#include <nori/emitter.h>
#include <nori/mesh.h>
#include <nori/rfilter.h>
#include <filesystem/resolver.h>
#include <pcg32.h>
#include <cstring>
#include "../obvh.hpp"

NORI_NAMESPACE_BEGIN

namespace {
thread_local uint32_t g_last_prim = 0;     // global primitive id (m_indices value) of the last closest hit
struct ShapePrims {
    const std::vector<Shape *> &shapes; const std::vector<uint32_t> &offs; std::vector<uint32_t> owner;
    ShapePrims(const std::vector<Shape *> &s, const std::vector<uint32_t> &o) : shapes(s), offs(o) {
        owner.resize(o.back());
        for (size_t k = 0; k < s.size(); ++k) for (uint32_t i = o[k]; i < o[k + 1]; ++i) owner[i] = (uint32_t) k;
    }
    uint32_t count() const { return offs.back(); }
    orc::V3 centroid(uint32_t i) const { Point3f c = shapes[owner[i]]->getCentroid(i - offs[owner[i]]); return orc::mk3(c.x(), c.y(), c.z()); }
    orc::BBox bbox(uint32_t i) const {
        BoundingBox3f b = shapes[owner[i]]->getBoundingBox(i - offs[owner[i]]);
        orc::BBox r; r.mn = orc::mk3(b.min.x(), b.min.y(), b.min.z()); r.mx = orc::mk3(b.max.x(), b.max.y(), b.max.z()); return r;
    }
};
}

void BVH::addShape(Shape *shape) {
    m_shapes.push_back(shape);
    m_shapeOffset.push_back(m_shapeOffset.back() + shape->getPrimitiveCount());
    m_bbox.expandBy(shape->getBoundingBox());
}
void BVH::clear() {
    for (auto s : m_shapes) delete s;
    m_shapes.clear(); m_shapeOffset.clear(); m_shapeOffset.push_back(0u); m_nodes.clear(); m_indices.clear(); m_bbox.reset();
}
void BVH::build() {
    if (getPrimitiveCount() == 0) return;
    ShapePrims sp(m_shapes, m_shapeOffset);
    orc::Bvh out; orc::BvhBuilder<ShapePrims> b(sp, out);
    orc::BBox rb; rb.mn = orc::mk3(m_bbox.min.x(), m_bbox.min.y(), m_bbox.min.z()); rb.mx = orc::mk3(m_bbox.max.x(), m_bbox.max.y(), m_bbox.max.z());
    b.build(rb);
    static_assert(sizeof(BVHNode) == sizeof(orc::BvhNode), "node layouts must agree");
    m_nodes.resize(out.nodes.size());
    for (size_t i = 0; i < out.nodes.size(); ++i) {
        const orc::BvhNode &s = out.nodes[i]; BVHNode &d = m_nodes[i];
        d.data = 0;
        if (s.is_leaf()) { d.leaf.flag = 1; d.leaf.size = s.size(); d.leaf.start = s.start(); }
        else { d.inner.flag = 0; d.inner.axis = s.axis(); d.inner.rightChild = s.right(); }
        d.bbox.min = Point3f(s.mn[0], s.mn[1], s.mn[2]); d.bbox.max = Point3f(s.mx[0], s.mx[1], s.mx[2]);
    }
    m_indices = out.indices;
}
std::pair<float, uint32_t> BVH::statistics(uint32_t) const { return std::make_pair(0.f, (uint32_t) m_nodes.size()); }

bool BVH::rayIntersect(const Ray3f &_ray, Intersection &its, bool shadowRay, const Shape *mask) const {
    its.t = std::numeric_limits<float>::infinity();
    Ray3f ray(_ray);
    if (ray.mint == Epsilon) ray.mint = std::max(ray.mint, ray.mint * ray.o.array().abs().maxCoeff());
    if (m_nodes.empty() || ray.maxt < ray.mint) return false;
    std::vector<uint32_t> todo; todo.reserve(64);
    uint32_t cur = 0, hitPrim = 0; bool found = false;
    for (;;) {
        const BVHNode &node = m_nodes[cur];
        bool descend = false;
        if (node.bbox.rayIntersect(ray)) {
            if (node.isInner()) { todo.push_back(node.inner.rightChild); ++cur; descend = true; }
            else {
                for (uint32_t i = node.start(); i < node.end(); ++i) {
                    uint32_t idx = m_indices[i], local = idx;
                    const Shape *shape = m_shapes[findShape(local)];
                    if (mask && mask != shape) continue;
                    float u, v, t;
                    if (shape->rayIntersect(local, ray, u, v, t)) {
                        if (shadowRay) return true;
                        found = true; ray.maxt = its.t = t; its.uv = Point2f(u, v); its.mesh = shape; hitPrim = local;
                        g_last_prim = idx;
                    }
                }
            }
        }
        if (descend) continue;
        if (todo.empty()) break;
        cur = todo.back(); todo.pop_back();
    }
    if (found) its.mesh->setHitInformation(hitPrim, ray, its);
    return found;
}
NORI_NAMESPACE_END

using namespace nori;

namespace {
thread_local std::string g_err;
struct RefHandle { NoriObject *root = nullptr; Scene *scene = nullptr; };
class SeededSampler : public Sampler {
public:
    SeededSampler() { m_sampleCount = 1; }
    SeededSampler(uint64_t state, uint64_t seq) { m_sampleCount = 1; m_random.seed(state, seq); }
    std::unique_ptr<Sampler> clone() const { std::unique_ptr<SeededSampler> c(new SeededSampler()); c->m_random = m_random; return std::move(c); }
    void prepare(const ImageBlock &) { }
    void generate() { }
    void advance() { }
    float next1D() { return m_random.nextFloat(); }
    Point2f next2D() { return Point2f(m_random.nextFloat(), m_random.nextFloat()); }     // as src/independent.cpp:62-67
    std::string toString() const { return "SeededSampler[]"; }
private:
    pcg32 m_random;
};
Sampler *make_sampler(const Scene *, int state, int seq) { return new SeededSampler((uint64_t) state, (uint64_t) seq); }
}

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

void *ref_load(const char *xml, const char *resolve_dir) {
    try {
        getFileResolver()->prepend(filesystem::path(resolve_dir));
        RefHandle *h = new RefHandle();
        h->root = loadFromXML(xml);
        if (h->root->getClassType() == NoriObject::EScene) h->scene = static_cast<Scene *>(h->root);
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void ref_scene_info(void *hp, int *out) {
    Scene *sc = ((RefHandle *) hp)->scene;
    out[0] = (int) sc->getShapes().size(); out[1] = (int) sc->getBVH()->getPrimitiveCount(); out[2] = (int) sc->getLights().size();
    out[3] = sc->getCamera()->getOutputSize().x(); out[4] = sc->getCamera()->getOutputSize().y(); out[5] = (int) sc->getSampler()->getSampleCount();
}
void ref_mesh_info(void *hp, int shape, int *out /* nV, nF, hasN, hasUV */) {
    const Mesh *m = static_cast<const Mesh *>(((RefHandle *) hp)->scene->getShapes()[shape]);
    out[0] = (int) m->getVertexPositions().cols(); out[1] = (int) m->getIndices().cols();
    out[2] = m->getVertexNormals().size() > 0; out[3] = m->getVertexTexCoords().size() > 0;
}
void ref_mesh_copy(void *hp, int shape, float *V, float *N, float *UV, uint32_t *F) {
    const Mesh *m = static_cast<const Mesh *>(((RefHandle *) hp)->scene->getShapes()[shape]);
    std::memcpy(V, m->getVertexPositions().data(), sizeof(float) * m->getVertexPositions().size());
    if (N && m->getVertexNormals().size() > 0) std::memcpy(N, m->getVertexNormals().data(), sizeof(float) * m->getVertexNormals().size());
    if (UV && m->getVertexTexCoords().size() > 0) std::memcpy(UV, m->getVertexTexCoords().data(), sizeof(float) * m->getVertexTexCoords().size());
    std::memcpy(F, m->getIndices().data(), sizeof(uint32_t) * m->getIndices().size());
}
void ref_camera_rays(void *hp, const float *ps, int n, float *rays) {
    const Camera *cam = ((RefHandle *) hp)->scene->getCamera();
    for (int i = 0; i < n; ++i) {
        Ray3f r; cam->sampleRay(r, Point2f(ps[2 * i], ps[2 * i + 1]), Point2f(0.f, 0.f));
        float *o = rays + 8 * (size_t) i;
        o[0] = r.o.x(); o[1] = r.o.y(); o[2] = r.o.z(); o[3] = r.mint; o[4] = r.d.x(); o[5] = r.d.y(); o[6] = r.d.z(); o[7] = r.maxt;
    }
}
void ref_trace(void *hp, const float *rays, int n, int any_hit, uint32_t *hits, float *full) {
    const Scene *sc = ((RefHandle *) hp)->scene;
    for (int i = 0; i < n; ++i) {
        const float *q = rays + 8 * (size_t) i;
        Ray3f r(Point3f(q[0], q[1], q[2]), Vector3f(q[4], q[5], q[6]), q[3], q[7]);
        uint32_t *h4 = hits + 4 * (size_t) i;
        if (any_hit) { h4[0] = sc->rayIntersect(r) ? 1u : 0xffffffffu; h4[1] = h4[2] = h4[3] = 0; continue; }
        Intersection its;
        if (!sc->rayIntersect(r, its)) { h4[0] = 0xffffffffu; h4[1] = h4[2] = h4[3] = 0; if (full) std::memset(full + 16 * (size_t) i, 0, 64); continue; }
        h4[0] = g_last_prim;
        float uvx = its.uv.x(), uvy = its.uv.y();
        std::memcpy(&h4[1], &its.t, 4); std::memcpy(&h4[2], &uvx, 4); std::memcpy(&h4[3], &uvy, 4);
        if (full) {
            float *f = full + 16 * (size_t) i;
            f[0] = its.p.x(); f[1] = its.p.y(); f[2] = its.p.z(); f[3] = uvx; f[4] = uvy;
            f[5] = its.shFrame.n.x(); f[6] = its.shFrame.n.y(); f[7] = its.shFrame.n.z();
            f[8] = its.geoFrame.n.x(); f[9] = its.geoFrame.n.y(); f[10] = its.geoFrame.n.z();
            int si = 0; for (auto s : sc->getShapes()) { if (s == its.mesh) break; ++si; }
            f[11] = (float) si; f[12] = f[13] = f[14] = f[15] = 0;
        }
    }
}
// src/render.cpp:85-98 for ONE camera sample of pixel (px,py) with the sampler seeded (state, seq)
void ref_li_samples(void *hp, int n, const int *px, const int *py, const int *states, const int *seqs, float *rgb, float *pixel_samples) {
    const Scene *sc = ((RefHandle *) hp)->scene;
    const Camera *cam = sc->getCamera(); const Integrator *integ = sc->getIntegrator();
    for (int i = 0; i < n; ++i) {
        std::unique_ptr<Sampler> s(make_sampler(sc, states[i], seqs[i]));
        Point2f pixelSample = Point2f((float) px[i], (float) py[i]) + s->next2D();
        Point2f apertureSample = s->next2D();
        Ray3f ray; Color3f value = cam->sampleRay(ray, pixelSample, apertureSample);
        value *= integ->Li(sc, s.get(), ray);
        rgb[3 * i] = value.r(); rgb[3 * i + 1] = value.g(); rgb[3 * i + 2] = value.b();
        if (pixel_samples) { pixel_samples[2 * i] = pixelSample.x(); pixel_samples[2 * i + 1] = pixelSample.y(); }
    }
}
void ref_li_rays(void *hp, int n, const float *rays, const int *states, const int *seqs, float *rgb) {
    const Scene *sc = ((RefHandle *) hp)->scene; const Integrator *integ = sc->getIntegrator();
    for (int i = 0; i < n; ++i) {
        std::unique_ptr<Sampler> s(make_sampler(sc, states[i], seqs[i])); s->next2D(); s->next2D();
        const float *q = rays + 8 * (size_t) i;
        Ray3f r(Point3f(q[0], q[1], q[2]), Vector3f(q[4], q[5], q[6]), q[3], q[7]);
        Color3f value = integ->Li(sc, s.get(), r);
        rgb[3 * i] = value.r(); rgb[3 * i + 1] = value.g(); rgb[3 * i + 2] = value.b();
    }
}
// next1D, next2D, next2D after seed(state, seq): pins the x/y order of Independent::next2D under this compiler
void ref_sampler_draws(int state, int seq, float *out5) {
    std::unique_ptr<Sampler> s(make_sampler(nullptr, state, seq));
    out5[0] = s->next1D(); Point2f a = s->next2D(); out5[1] = a.x(); out5[2] = a.y(); Point2f b = s->next2D(); out5[3] = b.x(); out5[4] = b.y();
}
// src/ttest.cpp:167-185 on one scene, sampler seeded (state, seq)
void ref_ttest_scene(void *hp, int n, int state, int seq, double *mean_out, double *var_out) {
    const Scene *scene = ((RefHandle *) hp)->scene;
    const Integrator *integrator = scene->getIntegrator(); const Camera *camera = scene->getCamera();
    std::unique_ptr<Sampler> sampler(make_sampler(scene, state, seq));
    double mean = 0, variance = 0;
    for (int k = 0; k < n; ++k) {
        Ray3f ray;
        Point2f pixelSample = (sampler->next2D().array() * camera->getOutputSize().cast<float>().array()).matrix();
        Color3f value = camera->sampleRay(ray, pixelSample, sampler->next2D());
        value *= integrator->Li(scene, sampler.get(), ray);
        double result = (double) value.getLuminance();
        double delta = result - mean; mean += delta / (double) (k + 1); variance += delta * (result - mean);
    }
    variance /= n - 1; *mean_out = mean; *var_out = variance;
}
// BSDF plugins through the factory.  type: 0 diffuse 1 microfacet 2 dielectric 3 mirror; params: rgb, alpha, intIOR, extIOR
void ref_bsdf_sample(int type, const float *params, int n, const float *wi, const float *sample, float *out) {
    PropertyList pl; const char *name = "diffuse";
    if (type == 0) { pl.setColor("albedo", Color3f(params[0], params[1], params[2])); }
    else if (type == 1) { name = "microfacet"; pl.setColor("kd", Color3f(params[0], params[1], params[2])); pl.setFloat("alpha", params[3]); pl.setFloat("intIOR", params[4]); pl.setFloat("extIOR", params[5]); }
    else if (type == 2) { name = "dielectric"; pl.setFloat("intIOR", params[4]); pl.setFloat("extIOR", params[5]); }
    else if (type == 4) { name = "subsurface"; pl.setColor("albedo", Color3f(params[0], params[1], params[2])); pl.setFloat("intIOR", params[4]); pl.setFloat("extIOR", params[5]); }
    else name = "mirror";
    BSDF *b = static_cast<BSDF *>(NoriObjectFactory::createInstance(name, pl));
    b->activate();
    Intersection its;
    for (int i = 0; i < n; ++i) {
        BSDFQueryRecord r(Vector3f(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
        r.measure = ESolidAngle; r.uv = Point2f(0.f, 0.f); r.wo = Vector3f(0.f, 0.f, 0.f);
        Color3f w = b->sample(r, its, Point2f(sample[2 * i], sample[2 * i + 1]), type == 4 ? Point2f(sample[2 * i], sample[2 * i + 1]) : Point2f(0.f, 0.f));
        float *o = out + 12 * (size_t) i;
        bool wo_valid = !((type == 0 || type == 3) && r.wi.z() <= 0);
        o[0] = w.r(); o[1] = w.g(); o[2] = w.b(); o[3] = r.wo.x(); o[4] = r.wo.y(); o[5] = r.wo.z();
        float pdf = wo_valid ? b->pdf(r) : 0.f; Color3f ev = wo_valid ? b->eval(r) : Color3f(0.f);
        o[6] = pdf; o[7] = ev.r(); o[8] = ev.g(); o[9] = ev.b(); o[10] = r.isDelta ? 1.f : 0.f; o[11] = wo_valid ? 1.f : 0.f;
    }
    delete b;
}
// Subsurface::sampleDx alone (src/subsurface.cpp:199-249): same layout as orc_ss_sample_dx
void ref_ss_sample_dx(const float *dmfp, int n, const float *p, const float *nrm, const uint64_t *states, const uint64_t *seqs, float *out) {
    PropertyList pl; pl.setColor("dmfp", Color3f(dmfp[0], dmfp[1], dmfp[2]));
    BSDF *b = static_cast<BSDF *>(NoriObjectFactory::createInstance("subsurface", pl));
    b->activate();
    for (int i = 0; i < n; ++i) {
        SeededSampler s(states[i], seqs[i]);
        Intersection its; its.p = Point3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        its.shFrame = Frame(Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2])); its.geoFrame = its.shFrame;
        BSDFQueryRecord r(Vector3f(0.f, 0.f, 1.f));
        b->sampleDx(r, its, s);
        float *o = out + 10 * (size_t) i;
        o[0] = r.probeRay.o.x(); o[1] = r.probeRay.o.y(); o[2] = r.probeRay.o.z(); o[3] = r.probeRay.d.x(); o[4] = r.probeRay.d.y(); o[5] = r.probeRay.d.z();
        o[6] = r.probeRay.mint; o[7] = r.probeRay.maxt; o[8] = r.dx.x(); o[9] = r.dx.y();
    }
    delete b;
}
// 1 if SeededSampler in its default state draws exactly what the reference's factory-made "independent" sampler draws
int ref_sampler_check() {
    std::unique_ptr<Sampler> a(static_cast<Sampler *>(NoriObjectFactory::createInstance("independent", PropertyList())));
    SeededSampler b;
    for (int i = 0; i < 64; ++i) {
        if (a->next1D() != b.next1D()) return 0;
        Point2f x = a->next2D(), y = b.next2D();
        if (x.x() != y.x() || x.y() != y.y()) return 0;
    }
    return 1;
}
void ref_pcg32(uint64_t state, uint64_t seq, int n, uint32_t *out) { pcg32 r; r.seed(state, seq); for (int i = 0; i < n; ++i) out[i] = r.nextUInt(); }

} // extern "C"
