// ORACLE — TEST INFRASTRUCTURE ONLY.  C API over the CPU restatement for ctypes (tests/, bench.py cpu_baseline leg,
// __graft_entry__.smoke()).  Never linked into, loaded by or called from the product library.
#include "orender.hpp"
#include <cstdio>
#include <chrono>

using namespace orc;

namespace {
struct Handle {
    Scene scene; Accel accel; Film film; std::string err;
    TestSuite suite; std::vector<Accel> suite_accels; bool is_suite = false;
};
thread_local std::string g_err;
}

extern "C" {

const char *orc_last_error() { return g_err.c_str(); }

void *orc_load_scene(const char *xml_path, int width, int height, int spp) {
    try {
        Handle *h = new Handle();
        h->scene = load_scene_xml(xml_path);
        if (width > 0) h->scene.camera.width = width;
        if (height > 0) h->scene.camera.height = height;
        if (width > 0 || height > 0) h->scene.camera.activate();
        if (spp > 0) h->scene.sample_count = spp;
        h->accel.build(h->scene);
        h->film.init(h->scene.camera.width, h->scene.camera.height, h->scene.camera.rfilter_radius);
        h->film.set_filter(h->scene.camera);
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void *orc_load_test(const char *xml_path) {
    try {
        Handle *h = new Handle();
        h->suite = load_test_xml(xml_path); h->is_suite = true;
        h->suite_accels.resize(h->suite.scenes.size());
        for (size_t i = 0; i < h->suite.scenes.size(); ++i) h->suite_accels[i].build(h->suite.scenes[i]);
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void orc_free(void *hp) { delete (Handle *) hp; }

// ---- test-suite access
int orc_test_count(void *hp) { return (int) ((Handle *) hp)->suite.scenes.size(); }
int orc_test_sample_count(void *hp) { return ((Handle *) hp)->suite.sample_count; }
float orc_test_significance(void *hp) { return ((Handle *) hp)->suite.significance; }
float orc_test_reference(void *hp, int i) { return ((Handle *) hp)->suite.references[i]; }
int orc_test_is_path_mis(void *hp, int i) { return ((Handle *) hp)->suite.scenes[i].integrator == "path_mis"; }
// 0 path_mis, 1 path_mats, 2 direct_mis, -1 anything else (not restated)
int orc_test_integrator(void *hp, int i) { const std::string &n = ((Handle *) hp)->suite.scenes[i].integrator; return n == "path_mis" ? 0 : n == "path_mats" ? 1 : n == "direct_mis" ? 2 : -1; }
// Runs scenes [0, n) in order with ONE default-seeded sampler carried across scenes, as src/ttest.cpp:152-156 does;
// every scene through the integrator it names (path_mis or path_mats).
void orc_test_run(void *hp, int only_path_mis, double *means, double *variances) {
    Handle *h = (Handle *) hp; Sampler smp;
    for (size_t i = 0; i < h->suite.scenes.size(); ++i) {
        if (only_path_mis && h->suite.scenes[i].integrator != "path_mis") { means[i] = variances[i] = 0; continue; }
        ttest_scene(h->suite_accels[i], smp, h->suite.sample_count, means[i], variances[i]);
    }
}
// Same, but scene i alone with a sampler seeded (seed_state, seed_seq)
void orc_test_run_one(void *hp, int i, uint64_t seed_state, uint64_t seed_seq, int n, double *mean, double *variance) {
    Handle *h = (Handle *) hp; Sampler smp; smp.rng.seed(seed_state, seed_seq);
    ttest_scene(h->suite_accels[i], smp, n, *mean, *variance);
}

static Handle *sub(void *hp, int scene_index, const Scene **sc, const Accel **ac) {
    Handle *h = (Handle *) hp;
    if (h->is_suite) { *sc = &h->suite.scenes[scene_index]; *ac = &h->suite_accels[scene_index]; }
    else { *sc = &h->scene; *ac = &h->accel; }
    return h;
}

// ---- scene queries (scene_index is ignored for plain scenes)
void orc_scene_info(void *hp, int si, int *out /*[8]: shapes, prims, emitters, width, height, spp, nodes, border*/) {
    const Scene *sc; const Accel *ac; Handle *h = sub(hp, si, &sc, &ac);
    out[0] = (int) sc->shapes.size(); out[1] = (int) sc->prim_total(); out[2] = (int) sc->emitters.size();
    out[3] = sc->camera.width; out[4] = sc->camera.height; out[5] = sc->sample_count; out[6] = (int) ac->bvh.nodes.size();
    out[7] = h->film.border;
}
void orc_bvh_stats(void *hp, int si, float *sah, int *out /*[4]: nodes, leaves, max_leaf, depth*/) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    *sah = ac->bvh.sah_cost; out[0] = (int) ac->bvh.nodes.size(); out[1] = (int) ac->bvh.leaf_count; out[2] = (int) ac->bvh.max_leaf; out[3] = (int) ac->bvh.depth;
}
void orc_bvh_copy(void *hp, int si, void *nodes32 /* nodes*32 bytes */, uint32_t *indices) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    std::memcpy(nodes32, ac->bvh.nodes.data(), ac->bvh.nodes.size() * sizeof(BvhNode));
    std::memcpy(indices, ac->bvh.indices.data(), ac->bvh.indices.size() * sizeof(uint32_t));
}
void orc_shape_info(void *hp, int si, int shape, int *out /*[8]: type, nV, nF, hasN, hasUV, emitter, bsdf_type, prim_offset*/) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    const Shape &s = sc->shapes[shape];
    out[0] = s.type; out[1] = (int) s.V.size(); out[2] = (int) s.prim_count(); out[3] = !s.N.empty(); out[4] = !s.UV.empty();
    out[5] = s.emitter; out[6] = s.bsdf.type; out[7] = (int) sc->shape_offset[shape];
}
void orc_shape_copy(void *hp, int si, int shape, float *V, float *N, float *UV, uint32_t *F, float *cdf) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    const Shape &s = sc->shapes[shape];
    if (V) std::memcpy(V, s.V.data(), s.V.size() * sizeof(V3));
    if (N && !s.N.empty()) std::memcpy(N, s.N.data(), s.N.size() * sizeof(V3));
    if (UV && !s.UV.empty()) std::memcpy(UV, s.UV.data(), s.UV.size() * sizeof(V2));
    if (F) std::memcpy(F, s.F.data(), s.F.size() * sizeof(uint32_t));
    if (cdf && !s.cdf.empty()) std::memcpy(cdf, s.cdf.data(), s.cdf.size() * sizeof(float));
}
// images: info = kind, lookup (1 bilinear), cols, rows, then scale xy and delta xy as float bits; texels in the device layout
// (4 bytes per texel: RGBA, or the two int16 gradients of a bump map)
int orc_image_count(void *hp, int si) { const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac); return (int) sc->images.size(); }
void orc_image_get(void *hp, int si, int k, int *info /*[4]*/, float *xform /*[4]*/, uint8_t *texels /* may be null */) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    const Image &im = sc->images[k];
    info[0] = im.kind; info[1] = im.bilinear ? 1 : 0; info[2] = im.cols; info[3] = im.rows;
    xform[0] = im.scale.x; xform[1] = im.scale.y; xform[2] = im.delta.x; xform[3] = im.delta.y;
    if (!texels) return;
    const size_t N = (size_t) im.cols * im.rows;
    if (im.kind == IMG_CHECKER_COLOR || im.kind == IMG_CHECKER_FLOAT) std::memcpy(texels, im.values, N * 4);       // the constants, as the C ABI carries them
    else if (im.kind == IMG_BUMP) for (size_t i = 0; i < N; ++i) { std::memcpy(texels + 4 * i, &im.gx[i], 2); std::memcpy(texels + 4 * i + 2, &im.gy[i], 2); }
    else std::memcpy(texels, im.rgba.data(), N * 4);
}
// albedo image / normal modifier of a shape (-1: none) and its bsdf's subsurface dmfp
void orc_shape_images(void *hp, int si, int shape, int *out /*[2]*/, float *dmfp /*[3]*/) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    const Shape &s = sc->shapes[shape];
    out[0] = s.bsdf.albedo_tex; out[1] = s.normal_tex; dmfp[0] = s.bsdf.dmfp.x; dmfp[1] = s.bsdf.dmfp.y; dmfp[2] = s.bsdf.dmfp.z;
}
// texture / modifier evaluation alone: colour of image k at uv, or the modified normal for shading normal nrm
void orc_image_eval(void *hp, int si, int k, int n, const float *uv, const float *nrm, float *out /* n x 3 */) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    const Image &im = sc->images[k];
    for (int i = 0; i < n; ++i) {
        V2 q = mk2(uv[2 * i], uv[2 * i + 1]);
        V3 r = (im.kind == IMG_COLOR || im.kind == IMG_CHECKER_COLOR) ? image_color(im, q) : image_normal(im, q, Frame(mk3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2])));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}

// camera matrices, row-major 4x4: sampleToCamera, cameraToWorld; + [invW, invH, near, far]
void orc_camera(void *hp, int si, float *s2c, float *c2w, float *misc) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    std::memcpy(s2c, sc->camera.sampleToCamera.mat.m, 64); std::memcpy(c2w, sc->camera.cameraToWorld.mat.m, 64);
    misc[0] = sc->camera.invW; misc[1] = sc->camera.invH; misc[2] = sc->camera.nearClip; misc[3] = sc->camera.farClip;
}

// ---- rays: 8 floats each (o.xyz, mint, d.xyz, maxt)
void orc_camera_rays(void *hp, int si, const float *pixel_samples /* n x 2 */, int n, float *rays) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    for (int i = 0; i < n; ++i) {
        Ray r; sc->camera.sample_ray(r, mk2(pixel_samples[2 * i], pixel_samples[2 * i + 1]));
        float *o = rays + 8 * (size_t) i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.mint; o[4] = r.d.x; o[5] = r.d.y; o[6] = r.d.z; o[7] = r.maxt;
    }
}
// hits: prim (uint32, 0xffffffff = miss), then t,u,v as floats → 4 x 32-bit per ray.  Optional full records:
// full[16] = p.xyz, uv.xy(after setHitInformation), sh.n.xyz, geo.n.xyz, shape, 4 pad
void orc_trace(void *hp, int si, const float *rays, int n, int any_hit, uint32_t *hits, float *full, uint64_t *counters /*[2] nodes,tris or null*/) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    TraceCounters tc;
    for (int i = 0; i < n; ++i) {
        const float *q = rays + 8 * (size_t) i;
        Ray r; r.o = mk3(q[0], q[1], q[2]); r.mint = q[3]; r.d = mk3(q[4], q[5], q[6]); r.maxt = q[7]; r.update();
        Intersection its; its.uv = mk2(0, 0); its.prim = 0;
        bool hit = ac->ray_intersect(r, its, any_hit != 0, -1, counters ? &tc : nullptr);
        uint32_t *h4 = hits + 4 * (size_t) i;
        if (any_hit) { h4[0] = hit ? 1u : 0xffffffffu; h4[1] = h4[2] = h4[3] = 0; continue; }
        if (!hit) { h4[0] = 0xffffffffu; h4[1] = h4[2] = h4[3] = 0; if (full) std::memset(full + 16 * (size_t) i, 0, 64); continue; }
        h4[0] = its.prim;
        // barycentric u,v of the winning triangle are overwritten by texture uv in setHitInformation when the mesh has UVs;
        // report t plus the final uv
        std::memcpy(&h4[1], &its.t, 4); std::memcpy(&h4[2], &its.uv.x, 4); std::memcpy(&h4[3], &its.uv.y, 4);
        if (full) {
            float *f = full + 16 * (size_t) i;
            f[0] = its.p.x; f[1] = its.p.y; f[2] = its.p.z; f[3] = its.uv.x; f[4] = its.uv.y;
            f[5] = its.sh.n.x; f[6] = its.sh.n.y; f[7] = its.sh.n.z; f[8] = its.geo.n.x; f[9] = its.geo.n.y; f[10] = its.geo.n.z;
            f[11] = (float) its.shape; f[12] = f[13] = f[14] = f[15] = 0;
        }
    }
    if (counters) { counters[0] = tc.nodes; counters[1] = tc.tris; }
}

// ---- radiance of individual camera samples with per-sample seeding: sample i uses pcg32.seed(states[i], seqs[i]),
// draws the pixel jitter + aperture as src/render.cpp:85-86 does for pixel (px[i], py[i]), then Li.
void orc_li_samples(void *hp, int si, int n, const int *px, const int *py, const uint64_t *states, const uint64_t *seqs, float *rgb, float *pixel_samples) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    for (int i = 0; i < n; ++i) {
        Sampler s; s.rng.seed(states[i], seqs[i]);
        V2 j = s.next2D(); V2 ps = mk2((float) px[i] + j.x, (float) py[i] + j.y); (void) s.next2D();
        Ray ray; sc->camera.sample_ray(ray, ps);
        V3 v = scene_li(*ac, s, ray);
        rgb[3 * i] = v.x; rgb[3 * i + 1] = v.y; rgb[3 * i + 2] = v.z;
        if (pixel_samples) { pixel_samples[2 * i] = ps.x; pixel_samples[2 * i + 1] = ps.y; }
    }
}

// ---- rendering.  mode 0 = reference block streams, 1 = per-sample streams.  Accumulates into the handle's film.
// 1: upstream Nori's filter-weighted splat (src/block.cpp:113-128), 0: this fork's single-pixel add (default)
void orc_film_mode(void *hp, int splat) { ((Handle *) hp)->film.splat = splat != 0; }
void orc_film_filter_table(void *hp, float *table33, float *radius) { Handle *h = (Handle *) hp; std::memcpy(table33, h->film.table, 33 * 4); *radius = h->film.radius; }
void orc_film_clear(void *hp) { Handle *h = (Handle *) hp; std::fill(h->film.px.begin(), h->film.px.end(), 0.f); }
double orc_render(void *hp, int spp_begin, int spp_end, int mode, int threads, uint64_t *stats /*[5]: closest, shadow, bounces, nodes, tris or null*/) {
    Handle *h = (Handle *) hp; PathStats ps;
    auto t0 = std::chrono::steady_clock::now();
    render(h->accel, h->film, spp_begin, spp_end, mode, threads, &ps);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) { stats[0] = ps.closest; stats[1] = ps.shadow; stats[2] = ps.bounces; stats[3] = ps.tc.nodes; stats[4] = ps.tc.tris; }
    return sec;
}
// film: (h+2b) x (w+2b) x 4 floats;  bitmap: h x w x 3
void orc_film_copy(void *hp, float *film) { Handle *h = (Handle *) hp; std::memcpy(film, h->film.px.data(), h->film.px.size() * 4); }
void orc_bitmap(void *hp, float *rgb) { Handle *h = (Handle *) hp; auto b = h->film.to_bitmap(); std::memcpy(rgb, b.data(), b.size() * 4); }
// film semantics only: put n samples (pos xy, rgb, block offset) into a fresh film of (w,h) and return it
void orc_film_put(int w, int h, float radius, int n, const float *pos, const float *rgb, float *film_out) {
    Film f; f.init(w, h, radius);
    for (int i = 0; i < n; ++i) {
        int px = (int) pos[2 * i], py = (int) pos[2 * i + 1];
        f.put(mk2(pos[2 * i], pos[2 * i + 1]), mk3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]), (px / 32) * 32, (py / 32) * 32);
    }
    std::memcpy(film_out, f.px.data(), f.px.size() * 4);
}

// ---- pcg32 KAT + sampler order
void orc_pcg32(uint64_t state, uint64_t seq, int n, uint32_t *out) { Pcg32 r; r.seed(state, seq); for (int i = 0; i < n; ++i) out[i] = r.next_uint(); }
void orc_pcg32_floats(uint64_t state, uint64_t seq, int n, float *out) { Pcg32 r; r.seed(state, seq); for (int i = 0; i < n; ++i) out[i] = r.next_float(); }

// ---- BSDF unit access: type, params[8] = albedo/kd rgb, alpha, intIOR, extIOR ; returns weight rgb, wo, pdf, eval rgb, isDelta
void orc_bsdf_sample(int type, const float *params, int n, const float *wi, const float *sample, float *out /* n x 12 */) {
    Bsdf b; b.type = type; b.albedo = mk3(params[0], params[1], params[2]); b.kd = b.albedo; b.alpha = params[3]; b.intIOR = params[4]; b.extIOR = params[5];
    b.ks = 1 - maxcoeff(b.kd);
    for (int i = 0; i < n; ++i) {
        BsdfRec r; r.wi = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]); r.uv = mk2(0, 0); r.wo = mk3(0.f); r.albedo = b.albedo;
        V3 w = bsdf_sample(b, r, mk2(sample[2 * i], sample[2 * i + 1]), mk2(sample[2 * i], sample[2 * i + 1]));
        float *o = out + 12 * (size_t) i;
        o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = r.wo.x; o[4] = r.wo.y; o[5] = r.wo.z;
        float pdf = r.wo_valid ? bsdf_pdf(b, r) : 0.f; V3 ev = r.wo_valid ? bsdf_eval(b, r) : mk3(0.f);
        o[6] = pdf; o[7] = ev.x; o[8] = ev.y; o[9] = ev.z; o[10] = r.isDelta ? 1.f : 0.f; o[11] = r.wo_valid ? 1.f : 0.f;
    }
}

// ---- Subsurface::sampleDx alone: hit at p with shading normal nrm, sampler seeded (state, seq); out = probe o, d, mint, maxt, dx (10 floats)
void orc_ss_sample_dx(const float *dmfp, int n, const float *p, const float *nrm, const uint64_t *states, const uint64_t *seqs, float *out) {
    Bsdf b; b.type = BSDF_SUBSURFACE; b.dmfp = mk3(dmfp[0], dmfp[1], dmfp[2]);
    for (int i = 0; i < n; ++i) {
        Sampler s; s.rng.seed(states[i], seqs[i]);
        Frame sh(mk3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
        Ray pr; V2 dx; ss_sample_dx(b, mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), sh, s, pr, dx);
        float *o = out + 10 * (size_t) i;
        o[0] = pr.o.x; o[1] = pr.o.y; o[2] = pr.o.z; o[3] = pr.d.x; o[4] = pr.d.y; o[5] = pr.d.z; o[6] = pr.mint; o[7] = pr.maxt; o[8] = dx.x; o[9] = dx.y;
    }
}

// ---- NL-means filter pass on an sRGB image (row-major h x w x 3)
void orc_nl_means(const float *img, float *out, int w, int h, int r, int f, float sigma, float hh) { nl_means(img, out, w, h, r, f, sigma, hh); }

} // extern "C"

extern "C" {
// Li for explicit rays: sample i uses pcg32.seed(states[i], seqs[i]), skips the 4 camera draws, then runs Li.
void orc_li_rays(void *hp, int si, int n, const float *rays, const uint64_t *states, const uint64_t *seqs, float *rgb) {
    const Scene *sc; const Accel *ac; sub(hp, si, &sc, &ac);
    for (int i = 0; i < n; ++i) {
        Sampler s; s.rng.seed(states[i], seqs[i]); (void) s.next2D(); (void) s.next2D();
        const float *q = rays + 8 * (size_t) i;
        Ray r; r.o = mk3(q[0], q[1], q[2]); r.mint = q[3]; r.d = mk3(q[4], q[5], q[6]); r.maxt = q[7]; r.update();
        V3 v = scene_li(*ac, s, r);
        rgb[3 * i] = v.x; rgb[3 * i + 1] = v.y; rgb[3 * i + 2] = v.z;
    }
}
}
