// C entry points of libnori_host.so (include/nori_host.h).
#include "render.h"
#include "../../include/nori_host.h"
#include <cstring>

using namespace nori;

struct NoriHostScene {
    std::unique_ptr<NoriObject> root; std::vector<Scene *> scenes; StudentsTTest *test = nullptr; ChiSquareTest *chi2 = nullptr;
    std::vector<std::unique_ptr<FlatScene>> flats;
    // programmatic scenes
    Scene *building = nullptr;
};
static thread_local std::string g_err;
#define GUARD(...) try { __VA_ARGS__ } catch (const std::exception &e) { g_err = e.what(); return -1; }
static Scene *sceneOf(NoriHostScene *h, int i) { if (!h || i < 0 || i >= (int) h->scenes.size()) throw NoriException("scene index out of range"); return h->scenes[i]; }
static GpuIntegrator *gpuOf(NoriHostScene *h, int i) {
    GpuIntegrator *g = dynamic_cast<GpuIntegrator *>(sceneOf(h, i)->getIntegrator());
    if (!g) throw NoriException("the scene's integrator is not the HIP path_mis integrator"); return g;
}

extern "C" {

const char *nori_host_last_error(void) { return g_err.c_str(); }

NoriHostScene *nori_host_load(const char *xml_path, int width, int height, int spp) {
    try {
        std::string f(xml_path); size_t slash = f.find_last_of('/');
        getFileResolver()->prepend(slash == std::string::npos ? std::string(".") : f.substr(0, slash));
        bool saved = StudentsTTest::s_runOnActivate, saved2 = ChiSquareTest::s_runOnActivate; StudentsTTest::s_runOnActivate = false; ChiSquareTest::s_runOnActivate = false;
        std::unique_ptr<NoriHostScene> h(new NoriHostScene());
        try { h->root.reset(loadFromXML(f)); } catch (...) { StudentsTTest::s_runOnActivate = saved; ChiSquareTest::s_runOnActivate = saved2; throw; }
        StudentsTTest::s_runOnActivate = saved; ChiSquareTest::s_runOnActivate = saved2;
        if (h->root->getClassType() == NoriObject::EScene) {
            Scene *s = static_cast<Scene *>(h->root.get()); h->scenes.push_back(s);
            if (width > 0 || height > 0) s->getCamera()->setOutputSize(width > 0 ? width : s->getCamera()->getWidth(), height > 0 ? height : s->getCamera()->getHeight());
            if (spp > 0) s->getSampler()->setSampleCount((size_t) spp);
        } else if (h->root->getClassType() == NoriObject::ETest) {
            h->test = dynamic_cast<StudentsTTest *>(h->root.get());
            h->chi2 = dynamic_cast<ChiSquareTest *>(h->root.get());
            if (!h->test && !h->chi2) throw NoriException("unsupported test type");
            if (h->test) h->scenes = h->test->scenes();
        } else throw NoriException("root element is neither a scene nor a test");
        h->flats.resize(h->scenes.size());
        return h.release();
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void nori_host_free(NoriHostScene *h) { delete h; }
int nori_host_is_test(NoriHostScene *h) { return h && h->test ? 1 : 0; }
int nori_host_scene_count(NoriHostScene *h) { return h ? (int) h->scenes.size() : 0; }

NoriHostScene *nori_host_scene_begin(void) {
    try { NoriHostScene *h = new NoriHostScene(); h->building = static_cast<Scene *>(NoriObjectFactory::createInstance("scene", PropertyList())); h->root.reset(h->building); return h; }
    catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
int nori_host_scene_add_mesh(NoriHostScene *h, const char *name, const float *V, size_t nV, const uint32_t *F, size_t nF, const float *N, const float *UV,
                             const char *bsdf_type, const float *kd, float alpha, float int_ior, float ext_ior, const float *radiance) {
    GUARD(
        if (!h || !h->building) throw NoriException("no scene under construction");
        for (size_t i = 0; i < 3 * nF; ++i) if (F[i] >= nV) throw NoriException("face index out of range");
        Mesh *m = static_cast<Mesh *>(NoriObjectFactory::createInstance("array_mesh", PropertyList()));
        std::unique_ptr<Mesh> guard(m);
        arrayMeshSet(m, name ? name : "array", V, nV, F, nF, N, UV);
        PropertyList bp; std::string bt = bsdf_type ? bsdf_type : "diffuse";
        if (bt == "diffuse") { if (kd) bp.setColor("albedo", Color3f(kd[0], kd[1], kd[2])); }
        else if (bt == "microfacet") { if (kd) bp.setColor("kd", Color3f(kd[0], kd[1], kd[2])); bp.setFloat("alpha", alpha); bp.setFloat("intIOR", int_ior); bp.setFloat("extIOR", ext_ior); }
        else if (bt == "dielectric") { bp.setFloat("intIOR", int_ior); bp.setFloat("extIOR", ext_ior); }
        NoriObject *b = NoriObjectFactory::createInstance(bt, bp); b->activate(); m->addChild(b);
        if (radiance) { PropertyList ep; ep.setColor("radiance", Color3f(radiance[0], radiance[1], radiance[2])); NoriObject *e = NoriObjectFactory::createInstance("area", ep); e->activate(); m->addChild(e); }
        m->activate();
        h->building->addChild(guard.release());
        return 0;
    )
}
int nori_host_scene_set_camera(NoriHostScene *h, const float *o, const float *t, const float *up, float fov, int width, int height) {
    GUARD(
        if (!h || !h->building) throw NoriException("no scene under construction");
        Vector3f origin(o[0], o[1], o[2]), target(t[0], t[1], t[2]), upv(up[0], up[1], up[2]);
        Vector3f dir = normalized(target - origin), left = normalized(cross(normalized(upv), dir)), newUp = normalized(cross(dir, left));
        float m[4][4] = {{left.x, newUp.x, dir.x, origin.x}, {left.y, newUp.y, dir.y, origin.y}, {left.z, newUp.z, dir.z, origin.z}, {0, 0, 0, 1}};
        PropertyList p; p.setTransform("toWorld", Transform(m)); p.setFloat("fov", fov); p.setInteger("width", width); p.setInteger("height", height);
        NoriObject *c = NoriObjectFactory::createInstance("perspective", p); c->activate(); h->building->addChild(c);
        return 0;
    )
}
int nori_host_scene_end(NoriHostScene *h, int spp, int build_threads) {
    GUARD(
        if (!h || !h->building) throw NoriException("no scene under construction");
        PropertyList sp; sp.setInteger("sampleCount", spp > 0 ? spp : 1);
        NoriObject *s = NoriObjectFactory::createInstance("independent", sp); s->activate(); h->building->addChild(s);
        NoriObject *in = NoriObjectFactory::createInstance("path_mis", PropertyList()); in->activate(); h->building->addChild(in);
        (void) build_threads;
        h->building->activate();
        h->scenes.push_back(h->building); h->flats.resize(1); h->building = nullptr;
        return 0;
    )
}

int nori_host_info(NoriHostScene *h, int scene, int *info) {
    GUARD(
        Scene *s = sceneOf(h, scene);
        size_t inner = 0; for (auto &n : s->getBVH()->getNodes()) if (!(n.w0 & 1u)) ++inner;
        info[0] = (int) s->getShapes().size(); info[1] = (int) s->getBVH()->getPrimitiveCount(); info[2] = (int) s->getLights().size();
        info[3] = s->getCamera()->getWidth(); info[4] = s->getCamera()->getHeight(); info[5] = (int) s->getSampler()->getSampleCount();
        info[6] = (int) s->getBVH()->getNodes().size(); info[7] = (int) inner;
        return 0;
    )
}
double nori_host_bvh_build_seconds(NoriHostScene *h, int scene) { try { return sceneOf(h, scene)->getBVH()->getBuildSeconds(); } catch (...) { return -1; } }
int nori_host_bvh32(NoriHostScene *h, int scene, const void **nodes32, uint32_t *n_nodes, const uint32_t **indices, uint32_t *n_indices) {
    GUARD(
        const BVH *b = sceneOf(h, scene)->getBVH();
        static_assert(sizeof(BVH::Node32) == 32, "Node32 must be 32 bytes");
        *nodes32 = b->getNodes().data(); *n_nodes = (uint32_t) b->getNodes().size(); *indices = b->getIndices().data(); *n_indices = (uint32_t) b->getIndices().size();
        return 0;
    )
}
int nori_host_mesh(NoriHostScene *h, int scene, int shape, int *counts, const float **V, const float **N, const float **UV, const uint32_t **F, const float **cdf) {
    GUARD(
        Scene *s = sceneOf(h, scene);
        if (shape < 0 || shape >= (int) s->getShapes().size()) throw NoriException("shape index out of range");
        const Mesh *m = dynamic_cast<const Mesh *>(s->getShapes()[shape]);
        if (!m) { counts[0] = counts[1] = counts[2] = counts[3] = 0; return 1; }
        counts[0] = (int) (m->getVertexPositions().size() / 3); counts[1] = (int) m->getPrimitiveCount();
        counts[2] = !m->getVertexNormals().empty(); counts[3] = !m->getVertexTexCoords().empty();
        *V = m->getVertexPositions().data(); *N = counts[2] ? m->getVertexNormals().data() : nullptr; *UV = counts[3] ? m->getVertexTexCoords().data() : nullptr;
        *F = m->getIndices().data(); *cdf = m->getAreaCdf().data();
        return 0;
    )
}
const NoriHipSceneDesc *nori_host_flatten(NoriHostScene *h, int scene) {
    try {
        Scene *s = sceneOf(h, scene);
        if (!h->flats[scene]) { h->flats[scene].reset(new FlatScene()); flattenScene(s, *h->flats[scene]); }
        return &h->flats[scene]->desc;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
int nori_host_preprocess(NoriHostScene *h, int scene, int device) {
    GUARD(
        Scene *s = sceneOf(h, scene);
        if (device >= 0) { char b[16]; snprintf(b, sizeof(b), "%d", device); setenv("NORI_HIP_DEVICE", b, 1); }
        if (!gpuOf(h, scene)->context()) {
            // the plugin reads its device when constructed; rebuild it if an explicit device was requested after load
            s->getIntegrator()->preprocess(s);
        }
        return 0;
    )
}
NoriHipContext *nori_host_context(NoriHostScene *h, int scene) { try { return gpuOf(h, scene)->context(); } catch (const std::exception &e) { g_err = e.what(); return nullptr; } }
int nori_host_render(NoriHostScene *h, int scene, int spp_begin, int spp_end) { GUARD( Scene *s = sceneOf(h, scene); s->getIntegrator()->render(s, spp_begin, spp_end); return 0; ) }

int nori_host_test_meta(NoriHostScene *h, int scene, float *reference, int *sample_count, float *significance, int *is_path_mis) {
    GUARD(
        if (!h || !h->test) throw NoriException("not a test file");
        sceneOf(h, scene);
        *reference = h->test->references().at(scene); *sample_count = h->test->sampleCount(); *significance = h->test->significanceLevel();
        *is_path_mis = dynamic_cast<GpuIntegrator *>(h->scenes[scene]->getIntegrator()) != nullptr;
        return 0;
    )
}
int nori_host_test_run(NoriHostScene *h, int scene, double *out) {
    GUARD(
        if (!h || !h->test) throw NoriException("not a test file");
        sceneOf(h, scene);
        TTestResult r = h->test->runScene((size_t) scene);
        out[0] = r.mean; out[1] = r.variance; out[2] = r.reference; out[3] = r.t; out[4] = r.pvalue; out[5] = r.alpha; out[6] = r.passed ? 1.0 : 0.0;
        return 0;
    )
}
int nori_host_li(NoriHostScene *h, int scene, size_t n, const float *rays, float *rgb) {
    GUARD(
        Scene *s = sceneOf(h, scene);
        for (size_t i = 0; i < n; ++i) {
            const float *q = rays + 8 * i;
            Color3f c = s->getIntegrator()->Li(s, s->getSampler(), Ray3f(Point3f(q[0], q[1], q[2]), Vector3f(q[4], q[5], q[6]), q[3], q[7]));
            rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
        }
        return 0;
    )
}
int nori_host_bsdf_tests_run(NoriHostScene *h, double *out, int max_tests) {
    GUARD(
        if (!h || (!h->chi2 && !(h->test && h->test->bsdfMode()))) throw NoriException("not a BSDF test file (chi2test, or ttest with <bsdf> children)");
        int n = 0;
        if (h->chi2) {
            for (const Chi2Result &r : h->chi2->run()) { if (n < max_tests) { double *o = out + 6 * n; o[0] = r.chsq; o[1] = r.dof; o[2] = r.pvalue; o[3] = r.alpha; o[4] = r.passed ? 1 : 0; o[5] = r.pooledCells; } ++n; }
        } else {
            for (const TTestResult &r : h->test->runBsdfs()) { if (n < max_tests) { double *o = out + 6 * n; o[0] = r.mean; o[1] = r.variance; o[2] = r.pvalue; o[3] = r.alpha; o[4] = r.passed ? 1 : 0; o[5] = r.reference; } ++n; }
        }
        return n;
    )
}
int nori_host_render_file(const char *xml_path, char *out_path, size_t out_len) {
    GUARD( std::string o = renderScene(xml_path); if (out_path && out_len) { strncpy(out_path, o.c_str(), out_len - 1); out_path[out_len - 1] = 0; } return 0; )
}
int nori_host_write_image(const char *path, const float *rgb, int w, int h) {
    GUARD(
        std::string p(path); std::vector<float> v(rgb, rgb + (size_t) w * h * 3);
        const std::string ext = p.size() > 4 ? p.substr(p.size() - 4) : std::string();
        if (ext == ".png") writePNG(p, v, w, h); else if (ext == ".exr") writeEXR(p, v, w, h); else writePFM(p, v, w, h);
        return 0;
    )
}

} // extern "C"
