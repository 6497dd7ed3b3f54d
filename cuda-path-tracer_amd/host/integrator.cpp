// The path_mis integrator plugin backed by libnori_hip.so, the t-test plugin, and the headless render driver.
// Reference: src/pathMisIntegrator.cpp:175 (NORI_REGISTER_CLASS(PathMisIntegrator, "path_mis")),
//            include/nori/integrator.h:35-62, src/render.cpp:103-214, src/ttest.cpp:58-220,
//            ext/hypothesis/hypothesis.h:313-346 (students_t_test).
#include "scene.h"
#include "render.h"
#include <cstring>
#include <iostream>
#include <fstream>
#include <atomic>
#include <mutex>

namespace nori {

Color3f Integrator::Li(const Scene *, Sampler *, const Ray3f &) const {
    throw NoriException::format("%s: Li() is not implemented by this integrator", toString().c_str());
}

static void hipCheck(int rc, const char *what) {
    if (rc != NORI_HIP_OK) throw NoriException::format("%s failed: %s", what, nori_hip_last_error());
}

/// GPU path tracer with multiple importance sampling.  Registered as "path_mis" (so existing scene files select
/// it unchanged) and as "path_mis_hip".  Properties: device (int, default 0 or $NORI_HIP_DEVICE); devices (string, default
/// $NORI_HIP_DEVICES: "all" or a comma-separated list such as "0,1,2,3" - the integrator then drives one context per listed GPU
/// through nori_hip_group_*, splits every render call among them and sums the films with RCCL; `device` is ignored);
/// partition (string, "spp" (default) or "tiles", default $NORI_HIP_PARTITION: how a group splits a render call);
/// bvh (string, "host" = the reference's SAH tree built by Scene::activate (default) or "device" = the tree built on the GPU at
/// preprocess(), default $NORI_HIP_BVH); film (string, "reference" = this fork's single-pixel ImageBlock::put (default) or
/// "splat" = upstream Nori's reconstruction-filter splat, default $NORI_HIP_FILM).
class PathMisHip : public GpuIntegrator {
public:
    explicit PathMisHip(const PropertyList &props, int kind = NORI_HIP_INTEGRATOR_PATH_MIS) : m_kind(kind) {
        const char *env = getenv("NORI_HIP_DEVICE");
        m_device = props.getInteger("device", env ? atoi(env) : 0);
        const char *denv = getenv("NORI_HIP_DEVICES");
        const std::string devs = props.getString("devices", denv ? denv : "");
        if (devs == "all") m_allDevices = true;
        else if (!devs.empty()) {
            for (const std::string &t : tokenize(devs, ", ")) {
                char *end = nullptr; long v = strtol(t.c_str(), &end, 10);
                if (end == t.c_str() || *end || v < 0) throw NoriException::format("path_mis: bad entry \"%s\" in devices", t.c_str());
                m_devices.push_back((int) v);
            }
        }
        const char *penv = getenv("NORI_HIP_PARTITION");
        const std::string part = props.getString("partition", penv ? penv : "spp");
        if (part != "spp" && part != "tiles") throw NoriException::format("path_mis: unknown partition \"%s\" (spp, tiles)", part.c_str());
        m_partition = part == "tiles" ? NORI_HIP_PARTITION_TILES : NORI_HIP_PARTITION_SPP;
        const char *benv = getenv("NORI_HIP_BVH");
        std::string bvh = props.getString("bvh", benv ? benv : "host");
        if (bvh != "host" && bvh != "device") throw NoriException::format("path_mis: unknown bvh mode \"%s\" (host, device)", bvh.c_str());
        m_deviceBvh = bvh == "device";
        const char *fenv = getenv("NORI_HIP_FILM");
        std::string film = props.getString("film", fenv ? fenv : "reference");
        if (film != "reference" && film != "splat") throw NoriException::format("path_mis: unknown film mode \"%s\" (reference, splat)", film.c_str());
        m_splat = film == "splat";
    }
    bool buildsOwnBVH() const override { return m_deviceBvh; }
    ~PathMisHip() override { if (m_group) nori_hip_group_destroy(m_group); else if (m_ctx) nori_hip_destroy(m_ctx); }
    void preprocess(const Scene *scene) override {
        flattenScene(scene, m_flat);
        if (m_flat.desc.n_emitters == 0 && !m_flat.desc.env.texture && m_kind == NORI_HIP_INTEGRATOR_PATH_MIS) throw NoriException("path_mis: the scene has no emitters");
        const bool grouped = m_allDevices || !m_devices.empty();
        if (grouped && !m_group) {
            if (m_allDevices) {
                int n = 0; hipCheck(nori_hip_device_count(&n), "nori_hip_device_count");
                m_devices.clear(); for (int i = 0; i < n; ++i) m_devices.push_back(i);
            }
            hipCheck(nori_hip_group_create(m_devices.data(), (int) m_devices.size(), &m_group), "nori_hip_group_create");
            m_ctx = nori_hip_group_context(m_group, 0);
        }
        if (!m_ctx) hipCheck(nori_hip_create(m_device, &m_ctx), "nori_hip_create");
        float table[33]; float radius = 0.f;
        if (m_splat) {
            // film = "splat": upstream Nori's filter-weighted accumulation; the table is ImageBlock::init's (src/block.cpp:54-64)
            const ReconstructionFilter *rf = scene->getCamera()->getReconstructionFilter();
            radius = rf->getRadius();
            for (int i = 0; i < 32; ++i) table[i] = rf->eval((radius * i) / 32);
            table[32] = 0.0f;
        }
        if (m_group) {
            hipCheck(nori_hip_group_set_integrator(m_group, m_kind), "nori_hip_group_set_integrator");
            hipCheck(nori_hip_group_set_film_filter(m_group, m_splat ? table : nullptr, radius), "nori_hip_group_set_film_filter");
            hipCheck(nori_hip_group_upload_scene(m_group, &m_flat.desc, m_deviceBvh ? 1 : 0), "nori_hip_group_upload_scene");
            return;
        }
        hipCheck(nori_hip_set_integrator(m_ctx, m_kind), "nori_hip_set_integrator");
        hipCheck(nori_hip_set_film_filter(m_ctx, m_splat ? table : nullptr, radius), "nori_hip_set_film_filter");
        if (m_deviceBvh) hipCheck(nori_hip_upload_scene_build(m_ctx, &m_flat.desc), "nori_hip_upload_scene_build");
        else hipCheck(nori_hip_upload_scene(m_ctx, &m_flat.desc), "nori_hip_upload_scene");
    }
    /// Accumulates samples [spp_begin, spp_end) of every pixel into the film (with a group: split among its devices, films summed).
    void render(const Scene *scene, int spp_begin, int spp_end) override {
        if (!m_ctx) preprocess(scene);
        const NoriHipCamera &c = m_flat.desc.camera;
        if (m_group) hipCheck(nori_hip_group_render(m_group, spp_begin, spp_end, m_partition), "nori_hip_group_render");
        else hipCheck(nori_hip_render(m_ctx, 0, 0, c.width, c.height, spp_begin, spp_end), "nori_hip_render");
    }
    void readFilm(std::vector<float> &film, int &rows, int &cols, int &border) override {
        need(); hipCheck(nori_hip_film_dims(m_ctx, &rows, &cols, &border), "nori_hip_film_dims");
        film.resize((size_t) rows * cols * 4); hipCheck(nori_hip_read_film(m_ctx, film.data()), "nori_hip_read_film");
    }
    void readBitmap(std::vector<float> &rgb) override {
        need(); const NoriHipCamera &c = m_flat.desc.camera;
        rgb.resize((size_t) c.width * c.height * 3); hipCheck(nori_hip_read_bitmap(m_ctx, rgb.data()), "nori_hip_read_bitmap");
    }
    /// Integrator::Li for one ray: PathMisIntegrator::Li (src/pathMisIntegrator.cpp:18-168) on the GPU through nori_hip_li_samples.
    /// The reference draws from the caller's Sampler; here call k of this integrator owns the stream pcg32.seed(k, 0x4c69), which has
    /// consumed the four draws of a camera sample (src/render.cpp:85-86) before the path starts.  preprocess() must have run.
    Color3f Li(const Scene *, Sampler *, const Ray3f &ray) const override {
        need();
        const uint64_t state = m_liCalls.fetch_add(1), seq = 0x4c69;
        NoriHipRay r; r.o[0] = ray.o.x; r.o[1] = ray.o.y; r.o[2] = ray.o.z; r.mint = ray.mint; r.d[0] = ray.d.x; r.d[1] = ray.d.y; r.d[2] = ray.d.z; r.maxt = ray.maxt;
        float rgb[3] = {0, 0, 0};
        {
            std::lock_guard<std::mutex> lock(m_liMutex);                  // a context serves one caller at a time
            hipCheck(nori_hip_li_samples(m_ctx, 1, nullptr, nullptr, &state, &seq, &r, rgb), "nori_hip_li_samples");
        }
        return Color3f(rgb[0], rgb[1], rgb[2]);
    }
    NoriHipContext *context() override { return m_ctx; }
    NoriHipGroup *group() { return m_group; }
    const FlatScene &flat() const override { return m_flat; }
    std::string toString() const override { return m_kind == NORI_HIP_INTEGRATOR_PATH_MATS ? "PathMatsIntegrator[backend=hip/gfx950]" : m_kind == NORI_HIP_INTEGRATOR_DIRECT_MIS ? "DirectMisIntegrator[backend=hip/gfx950]" : "PathMisIntegrator[backend=hip/gfx950]"; }
private:
    void need() const { if (!m_ctx) throw NoriException("path_mis: preprocess() has not run"); }
    int m_kind; int m_device = 0; bool m_deviceBvh = false, m_splat = false; NoriHipContext *m_ctx = nullptr; FlatScene m_flat;
    mutable std::atomic<uint64_t> m_liCalls{0}; mutable std::mutex m_liMutex;
    bool m_allDevices = false; std::vector<int> m_devices; int m_partition = NORI_HIP_PARTITION_SPP; NoriHipGroup *m_group = nullptr;
};
static NoriObject *PathMisHip_create(const PropertyList &l) { return new PathMisHip(l); }
// "path_mats" (src/pathMatsIntegrator.cpp:87): the same plugin with the other Li
static NoriObject *PathMatsHip_create(const PropertyList &l) { return new PathMisHip(l, NORI_HIP_INTEGRATOR_PATH_MATS); }
static struct PathMisHip_reg { PathMisHip_reg() {
    NoriObjectFactory::registerClass("path_mis", PathMisHip_create); NoriObjectFactory::registerClass("path_mis_hip", PathMisHip_create);
    NoriObjectFactory::registerClass("path_mats", PathMatsHip_create); NoriObjectFactory::registerClass("path_mats_hip", PathMatsHip_create);
    NoriObjectFactory::registerClass("direct_mis", [](const PropertyList &l) -> NoriObject * { return new PathMisHip(l, NORI_HIP_INTEGRATOR_DIRECT_MIS); });   // src/directMisIntegrator.cpp:147
} } PathMisHip_reg_;

/// The reference's test files also name integrators that are outside this build's scope.  They are
/// accepted by the parser so those files load unmodified, and refuse to render.
class OutOfScopeIntegrator : public Integrator {
public:
    OutOfScopeIntegrator(const PropertyList &, const char *name) : m_name(name) {}
    void render(const Scene *, int, int) override { fail(); }
    void readFilm(std::vector<float> &, int &, int &, int &) override { fail(); }
    void readBitmap(std::vector<float> &) override { fail(); }
    std::string toString() const override { return m_name + "[out of scope in this build]"; }
private:
    [[noreturn]] void fail() const { throw NoriException::format("integrator \"%s\" is outside the path_mis hot path and is not implemented by this build", m_name.c_str()); }
    std::string m_name;
};
static struct OutOfScope_reg { OutOfScope_reg() {
    for (const char *n : { "direct_ems", "direct_mats", "direct", "av", "normals", "photonmapper"})
        NoriObjectFactory::registerClass(n, [n](const PropertyList &l) -> NoriObject * { return new OutOfScopeIntegrator(l, n); });
} } OutOfScope_reg_;

// ---- Student's t-test -------------------------------------------------------------------------------------
// regularised incomplete beta by Lentz's continued fraction; enough for the CDF of Student's t
static double betacf(double a, double b, double x) {
    const double tiny = 1e-300; double qab = a + b, qap = a + 1, qam = a - 1, c = 1, d = 1 - qab * x / qap;
    if (std::fabs(d) < tiny) d = tiny; d = 1 / d; double h = d;
    for (int m = 1; m <= 500; ++m) {
        int m2 = 2 * m; double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1 + aa * d; if (std::fabs(d) < tiny) d = tiny; c = 1 + aa / c; if (std::fabs(c) < tiny) c = tiny; d = 1 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1 + aa * d; if (std::fabs(d) < tiny) d = tiny; c = 1 + aa / c; if (std::fabs(c) < tiny) c = tiny; d = 1 / d;
        double del = d * c; h *= del; if (std::fabs(del - 1) < 1e-15) break;
    }
    return h;
}
static double incbeta(double a, double b, double x) {
    if (x <= 0) return 0; if (x >= 1) return 1;
    double bt = std::exp(std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log(1 - x));
    if (x < (a + 1) / (a + b + 2)) return bt * betacf(a, b, x) / a;
    return 1 - bt * betacf(b, a, 1 - x) / b;
}
double studentsTCdf(double t, double dof) { double x = dof / (dof + t * t); double tail = 0.5 * incbeta(dof / 2, 0.5, x); return t >= 0 ? 1 - tail : tail; }

TTestResult studentsTTest(double mean, double variance, double reference, int sampleCount, double significanceLevel, int numTests) {
    TTestResult r; r.mean = mean; r.variance = variance; r.reference = reference;
    r.t = std::abs(mean - reference) * std::sqrt(sampleCount / std::max(variance, 1e-5));
    r.pvalue = 2 * (1 - studentsTCdf(r.t, sampleCount - 1));
    r.alpha = 1.0 - std::pow(1.0 - significanceLevel, 1.0 / numTests);          // Sidak correction
    r.passed = !(r.pvalue < r.alpha || !std::isfinite(r.pvalue));
    return r;
}

bool StudentsTTest::s_runOnActivate = true;
StudentsTTest::StudentsTTest(const PropertyList &p) {
    m_significanceLevel = p.getFloat("significanceLevel", 0.01f);
    for (auto &a : tokenize(p.getString("angles", ""))) m_angles.push_back(toFloat(a));
    for (auto &a : tokenize(p.getString("references", ""))) m_references.push_back(toFloat(a));
    m_sampleCount = p.getInteger("sampleCount", 100000);
}
StudentsTTest::~StudentsTTest() { for (auto s : m_scenes) delete s; for (auto b : m_bsdfs) delete b; }
void StudentsTTest::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
        case EBSDF: m_bsdfs.push_back(static_cast<BSDF *>(obj)); break;
        case EScene: m_scenes.push_back(static_cast<Scene *>(obj)); break;
        default: throw NoriException::format("StudentsTTest::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()).c_str());
    }
}
TTestResult StudentsTTest::runScene(size_t i) {
    Scene *scene = m_scenes.at(i);
    GpuIntegrator *gi = dynamic_cast<GpuIntegrator *>(scene->getIntegrator());
    if (!gi) throw NoriException("StudentsTTest: the scene's integrator has no GPU backend");
    if (!gi->context()) gi->preprocess(scene);
    // 100000 paths through the whole-image camera: pixelSample = next2D() * outputSize (src/ttest.cpp:171-173).
    // Path k owns the stream pcg32.seed(k, scene index) instead of sharing one sampler.
    const size_t n = (size_t) m_sampleCount;
    std::vector<uint64_t> st(n), sq(n, (uint64_t) i); for (size_t k = 0; k < n; ++k) st[k] = k;
    std::vector<float> rgb(3 * n);
    hipCheck(nori_hip_li_samples(gi->context(), n, nullptr, nullptr, st.data(), sq.data(), nullptr, rgb.data()), "nori_hip_li_samples");
    double mean = 0, variance = 0;
    for (size_t k = 0; k < n; ++k) {        // Knuth's online variance, as the reference (src/ttest.cpp:178-184)
        double result = (double) (rgb[3 * k] * 0.212671f + rgb[3 * k + 1] * 0.715160f + rgb[3 * k + 2] * 0.072169f);
        double delta = result - mean; mean += delta / (double) (k + 1); variance += delta * (result - mean);
    }
    variance /= m_sampleCount - 1;
    return studentsTTest(mean, variance, m_references.at(i), m_sampleCount, m_significanceLevel, (int) m_references.size());
}
void StudentsTTest::activate() {
    if (!m_bsdfs.empty() && !s_runOnActivate) return;
    if (m_bsdfs.empty() && m_references.size() != m_scenes.size()) throw NoriException("Specified a different number of scenes and reference values!");
    if (!s_runOnActivate) return;
    int passed = 0;
    if (!m_bsdfs.empty()) {
        m_results = runBsdfs();
        for (const TTestResult &r : m_results) {
            std::cout << "------------------------------------------------------\n"
                      << "Sample mean = " << r.mean << " (reference value = " << r.reference << ")\nSample variance = " << r.variance
                      << "\nt-statistic = " << r.t << " (d.o.f. = " << m_sampleCount - 1 << ")\n"
                      << (r.passed ? "Accepted" : "***** Rejected *****") << " the null hypothesis (p-value = " << r.pvalue << ", significance level = " << r.alpha << ")" << std::endl;
            if (r.passed) ++passed;
        }
        std::cout << "Passed " << passed << "/" << m_results.size() << " tests." << std::endl;
        return;
    }
    for (size_t i = 0; i < m_scenes.size(); ++i) {
        TTestResult r = runScene(i);
        m_results.push_back(r);
        std::cout << "------------------------------------------------------\n"
                  << "Sample mean = " << r.mean << " (reference value = " << r.reference << ")\nSample variance = " << r.variance
                  << "\nt-statistic = " << r.t << " (d.o.f. = " << m_sampleCount - 1 << ")\n"
                  << (r.passed ? "Accepted" : "***** Rejected *****") << " the null hypothesis (p-value = " << r.pvalue << ", significance level = " << r.alpha << ")" << std::endl;
        if (r.passed) ++passed;
    }
    std::cout << "Passed " << passed << "/" << m_scenes.size() << " tests." << std::endl;
}
std::string StudentsTTest::toString() const { char b[128]; snprintf(b, sizeof(b), "StudentsTTest[\n  significanceLevel = %f,\n  sampleCount= %i\n]", m_significanceLevel, m_sampleCount); return b; }
NORI_REGISTER_CLASS(StudentsTTest, "ttest")

// ---- headless render driver (src/render.cpp:103-214 without the GUI thread) --------------------------------
void writePFM(const std::string &path, const std::vector<float> &rgb, int w, int h) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw NoriException::format("Unable to write \"%s\"", path.c_str());
    f << "PF\n" << w << " " << h << "\n-1.0\n";
    for (int y = h - 1; y >= 0; --y) f.write(reinterpret_cast<const char *>(&rgb[(size_t) y * w * 3]), (std::streamsize) w * 3 * 4);
}

// 8-bit sRGB PNG (what Bitmap::savePNG writes next to the EXR, src/bitmap.cpp:109-123: toSRGB, clamp, * 255), encoded with
// stored (uncompressed) deflate blocks so that no compression library is needed
void writePNG(const std::string &path, const std::vector<float> &rgb, int w, int h) {
    if (w <= 0 || h <= 0 || rgb.size() < (size_t) w * h * 3) throw NoriException("writePNG: bad image size");
    static uint32_t crcTable[256]; static bool crcInit = false;
    if (!crcInit) { for (uint32_t n = 0; n < 256; ++n) { uint32_t c = n; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; crcTable[n] = c; } crcInit = true; }
    auto crc = [&](const std::vector<uint8_t> &d, size_t from) { uint32_t c = 0xffffffffu; for (size_t i = from; i < d.size(); ++i) c = crcTable[(c ^ d[i]) & 0xff] ^ (c >> 8); return c ^ 0xffffffffu; };
    auto be32 = [](std::vector<uint8_t> &d, uint32_t v) { d.push_back((uint8_t) (v >> 24)); d.push_back((uint8_t) (v >> 16)); d.push_back((uint8_t) (v >> 8)); d.push_back((uint8_t) v); };
    auto toSRGB = [](float v) { return v <= 0.0031308f ? 12.92f * v : (1.0f + 0.055f) * std::pow(v, 1.0f / 2.4f) - 0.055f; };      // src/common.cpp:190-206
    std::vector<uint8_t> raw; raw.reserve((size_t) h * (3 * w + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);                                                      // filter type "none"
        for (int x = 0; x < 3 * w; ++x) { float v = toSRGB(rgb[(size_t) y * 3 * w + x]) * 255.f; raw.push_back((uint8_t) (v != v || v < 0.f ? 0.f : v > 255.f ? 255.f : v)); }
    }
    std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);              // zlib header, then stored blocks of at most 65535 bytes
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size(); off += 65535) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t) n); z.push_back((uint8_t) (n >> 8)); z.push_back((uint8_t) ~n); z.push_back((uint8_t) (~n >> 8));
        for (size_t i = 0; i < n; ++i) { uint8_t c = raw[off + i]; z.push_back(c); a = (a + c) % 65521u; b = (b + a) % 65521u; }
    }
    be32(z, (b << 16) | a);                                                    // adler32
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        be32(out, (uint32_t) data.size()); size_t from = out.size();
        out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
        be32(out, crc(out, from));
    };
    std::vector<uint8_t> ihdr; be32(ihdr, (uint32_t) w); be32(ihdr, (uint32_t) h); ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) throw NoriException::format("Unable to write \"%s\"", path.c_str());
    f.write(reinterpret_cast<const char *>(out.data()), (std::streamsize) out.size());
}

// OpenEXR, the reference's native output (Bitmap::save, src/bitmap.cpp:82-107: channels R, G, B as 32-bit floats): a
// single-part scanline file without compression, written directly (the format is a header of named attributes, a table of
// line offsets, and per line the channels in alphabetical order)
void writeEXR(const std::string &path, const std::vector<float> &rgb, int w, int h) {
    if (w <= 0 || h <= 0 || rgb.size() < (size_t) w * h * 3) throw NoriException("writeEXR: bad image size");
    std::vector<uint8_t> o;
    auto u8 = [&](uint8_t v) { o.push_back(v); };
    auto i32 = [&](int32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t) ((uint32_t) v >> (8 * k))); };
    auto u64 = [&](uint64_t v) { for (int k = 0; k < 8; ++k) o.push_back((uint8_t) (v >> (8 * k))); };
    auto f32 = [&](float v) { uint32_t u; std::memcpy(&u, &v, 4); i32((int32_t) u); };
    auto str = [&](const char *s) { while (*s) o.push_back((uint8_t) *s++); o.push_back(0); };
    auto attr = [&](const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); };
    i32(20000630); i32(2);                                                     // magic, version 2, no flags
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char *c : {"B", "G", "R"}) { str(c); i32(2 /* FLOAT */); u8(0); u8(0); u8(0); u8(0); i32(1); i32(1); }
    u8(0);
    attr("compression", "compression", 1); u8(0);                              // NO_COMPRESSION
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32(w - 1); i32(h - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32(w - 1); i32(h - 1);
    attr("lineOrder", "lineOrder", 1); u8(0);                                  // INCREASING_Y
    attr("pixelAspectRatio", "float", 4); f32(1.0f);
    attr("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
    attr("screenWindowWidth", "float", 4); f32(1.0f);
    u8(0);                                                                     // end of header
    const uint64_t lineBytes = 8 + (uint64_t) w * 12, table = o.size();
    for (int y = 0; y < h; ++y) u64(table + (uint64_t) h * 8 + (uint64_t) y * lineBytes);
    o.reserve(o.size() + (size_t) h * lineBytes);
    for (int y = 0; y < h; ++y) {
        i32(y); i32(w * 12);
        for (int c = 2; c >= 0; --c) for (int x = 0; x < w; ++x) f32(rgb[((size_t) y * w + x) * 3 + c]);
    }
    std::ofstream f(path, std::ios::binary);
    if (!f) throw NoriException::format("Unable to write \"%s\"", path.c_str());
    f.write(reinterpret_cast<const char *>(o.data()), (std::streamsize) o.size());
}

std::string renderScene(const std::string &filename, int passSize) {
    size_t slash = filename.find_last_of('/');
    getFileResolver()->prepend(slash == std::string::npos ? std::string(".") : filename.substr(0, slash));
    std::unique_ptr<NoriObject> root(loadFromXML(filename));
    if (root->getClassType() != NoriObject::EScene) return std::string();
    Scene *scene = static_cast<Scene *>(root.get());
    scene->getIntegrator()->preprocess(scene);
    const int spp = (int) scene->getSampler()->getSampleCount();
    if (passSize <= 0) passSize = spp;
    for (int k = 0; k < spp; k += passSize) scene->getIntegrator()->render(scene, k, std::min(spp, k + passSize));
    std::vector<float> rgb; scene->getIntegrator()->readBitmap(rgb);
    std::string out = filename; size_t dot = out.find_last_of('.'); if (dot != std::string::npos) out.erase(dot); out += ".pfm";
    writePFM(out, rgb, scene->getCamera()->getWidth(), scene->getCamera()->getHeight());
    writePNG(out.substr(0, out.size() - 4) + ".png", rgb, scene->getCamera()->getWidth(), scene->getCamera()->getHeight());
    writeEXR(out.substr(0, out.size() - 4) + ".exr", rgb, scene->getCamera()->getWidth(), scene->getCamera()->getHeight());
    return out;
}

} // namespace nori
