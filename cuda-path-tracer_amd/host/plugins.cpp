// The plugins the path_mis hot path needs, registered under the reference's names.
// Reference: src/obj.cpp, src/mesh.cpp, src/sphere.cpp, src/shape.cpp, src/diffuse.cpp, src/microfacet.cpp,
//            src/dielectric.cpp, src/mirror.cpp, src/arealight.cpp, src/perspective.cpp, src/independent.cpp,
//            src/rfilter.cpp, src/scene.cpp, src/consttexture.cpp.
// The OBJ reader itself is host/objreader.cpp (mapped file, parsed in parallel pieces, vertex numbers by first use): its OUTPUT is
// what src/obj.cpp:32-132 produces - the parity tests compare the arrays - its text is not that function's.
#include "scene.h"
#include <cstring>
#include <cstdio>
#include <sys/stat.h>
#include <fstream>
#include <sstream>
#include <unordered_map>
#include <iostream>
#include <algorithm>

namespace nori {

// ---- Shape ------------------------------------------------------------------------------------------------
Shape::~Shape() { delete m_bsdf; delete m_normalModifier; }
void Shape::activate() {
    if (!m_bsdf) { m_bsdf = static_cast<BSDF *>(NoriObjectFactory::createInstance("diffuse", PropertyList())); m_bsdf->activate(); }
}
void Shape::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
        case EBSDF:
            if (m_bsdf) throw NoriException("Shape: tried to register multiple BSDF instances!");
            m_bsdf = static_cast<BSDF *>(obj); break;
        case EEmitter:
            if (m_emitter) throw NoriException("Shape: tried to register multiple Emitter instances!");
            m_emitter = static_cast<Emitter *>(obj); m_emitter->setShape(this); break;
        case ENormalModifier:                                // cuda/nori/shape.cpp:58-63
            if (m_normalModifier) throw NoriException("Shape: tried to register multiple Normal modifier instances!");
            m_normalModifier = static_cast<NormalModifier *>(obj); break;
        default:
            throw NoriException::format("Shape::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()).c_str());
    }
}

// ---- Mesh -------------------------------------------------------------------------------------------------
float Mesh::surfaceArea(uint32_t i) const {
    Point3f p0 = vertex(m_F[3 * i]), p1 = vertex(m_F[3 * i + 1]), p2 = vertex(m_F[3 * i + 2]);
    return 0.5f * norm(cross(p1 - p0, p2 - p0));
}
BoundingBox3f Mesh::getBoundingBox(uint32_t i) const {
    BoundingBox3f r(vertex(m_F[3 * i])); r.expandBy(vertex(m_F[3 * i + 1])); r.expandBy(vertex(m_F[3 * i + 2])); return r;
}
Point3f Mesh::getCentroid(uint32_t i) const {
    return (1.0f / 3.0f) * ((vertex(m_F[3 * i]) + vertex(m_F[3 * i + 1])) + vertex(m_F[3 * i + 2]));
}
void Mesh::activate() {
    Shape::activate();
    // DiscretePDF over triangle areas: running fp32 sum, then normalise and pin the last entry to 1 (include/nori/dpdf.h:55-57,97-109)
    uint32_t n = getPrimitiveCount();
    m_cdf.assign(1, 0.0f); m_cdf.reserve(n + 1);
    for (uint32_t i = 0; i < n; ++i) m_cdf.push_back(m_cdf.back() + surfaceArea(i));
    float sum = m_cdf.back();
    if (sum > 0) { m_cdfNorm = 1.0f / sum; for (size_t i = 1; i < m_cdf.size(); ++i) m_cdf[i] *= m_cdfNorm; m_cdf.back() = 1.0f; }
    else m_cdfNorm = 0.0f;
}
std::string Mesh::toString() const {
    char buf[512];
    snprintf(buf, sizeof(buf), "Mesh[\n  name = \"%s\",\n  vertexCount = %zu,\n  triangleCount = %zu,\n  bsdf = %s,\n  emitter = %s\n]", m_name.c_str(),
             m_V.size() / 3, m_F.size() / 3, m_bsdf ? indent(m_bsdf->toString()).c_str() : "null", m_emitter ? indent(m_emitter->toString()).c_str() : "null");
    return buf;
}

/// Wavefront OBJ mesh (src/obj.cpp:32-132): v/vt/vn/f, toWorld applied at load, quads split (0,1,2),(3,0,2),
/// vertices de-duplicated on the (p, uv, n) index triple in order of first use - read by readWavefrontOBJ (host/objreader.cpp).
class WavefrontOBJ : public Mesh {
public:
    explicit WavefrontOBJ(const PropertyList &propList) {
        std::string filename = getFileResolver()->resolve(propList.getString("filename"));
        { std::ifstream is(filename); if (is.fail()) throw NoriException::format("Unable to open OBJ file \"%s\"!", filename.c_str()); }
        Transform trafo = propList.getTransform("toWorld", Transform());
        // Binary cache of the parsed, transformed, de-duplicated arrays (text-parsing a 10 M-triangle OBJ
        // dominates start-up).  Opt-in: $NORI_HIP_MESH_CACHE=1 reads and writes "<file>.noricache" next to the OBJ; an entry
        // is used only if the OBJ's size and modification time and the toWorld matrix are the ones it was made from.
        const char *cenv = getenv("NORI_HIP_MESH_CACHE");
        const bool useCache = cenv && atoi(cenv) != 0;
        if (useCache && loadCache(filename, trafo)) { m_name = filename; return; }
        readWavefrontOBJ(filename, trafo, m_V, m_N, m_UV, m_F, m_bbox);      // host/objreader.cpp
        m_name = filename;
        if (useCache) saveCache(filename, trafo);
    }
private:
    struct CacheHeader { char magic[8]; uint64_t srcSize; int64_t srcMtimeNs; float trafo[16]; uint64_t nV, nF; uint32_t hasN, hasUV; float bbox[6]; };
    static bool sourceStamp(const std::string &filename, uint64_t &size, int64_t &mtimeNs) {
        struct stat st;
        if (stat(filename.c_str(), &st) != 0) return false;
        size = (uint64_t) st.st_size; mtimeNs = (int64_t) st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
        return true;
    }
    bool loadCache(const std::string &filename, const Transform &trafo) {
        std::ifstream f(filename + ".noricache", std::ios::binary);
        CacheHeader h; uint64_t size; int64_t mtime;
        if (!f || !f.read(reinterpret_cast<char *>(&h), sizeof(h)) || std::memcmp(h.magic, "NORIMSH1", 8) != 0) return false;
        if (!sourceStamp(filename, size, mtime) || size != h.srcSize || mtime != h.srcMtimeNs || std::memcmp(h.trafo, trafo.m, 64) != 0) return false;
        if (h.nV == 0 || h.nF == 0 || h.nV > (1ull << 32) || h.nF > (1ull << 32)) return false;
        std::vector<float> V(3 * h.nV), N(h.hasN ? 3 * h.nV : 0), UV(h.hasUV ? 2 * h.nV : 0); std::vector<uint32_t> F(3 * h.nF);
        auto rd = [&](void *p, size_t bytes) { return bytes == 0 || (bool) f.read(reinterpret_cast<char *>(p), (std::streamsize) bytes); };
        if (!rd(V.data(), V.size() * 4) || !rd(N.data(), N.size() * 4) || !rd(UV.data(), UV.size() * 4) || !rd(F.data(), F.size() * 4)) return false;
        for (uint32_t i : F) if (i >= h.nV) return false;
        m_V.swap(V); m_N.swap(N); m_UV.swap(UV); m_F.swap(F);
        m_bbox.min = Point3f(h.bbox[0], h.bbox[1], h.bbox[2]); m_bbox.max = Point3f(h.bbox[3], h.bbox[4], h.bbox[5]);
        return true;
    }
    void saveCache(const std::string &filename, const Transform &trafo) const {
        CacheHeader h; std::memset(&h, 0, sizeof(h)); std::memcpy(h.magic, "NORIMSH1", 8);
        if (!sourceStamp(filename, h.srcSize, h.srcMtimeNs)) return;
        std::memcpy(h.trafo, trafo.m, 64);
        h.nV = m_V.size() / 3; h.nF = m_F.size() / 3; h.hasN = !m_N.empty(); h.hasUV = !m_UV.empty();
        for (int a = 0; a < 3; ++a) { h.bbox[a] = m_bbox.min[a]; h.bbox[3 + a] = m_bbox.max[a]; }
        const std::string tmp = filename + ".noricache.tmp";
        {
            std::ofstream f(tmp, std::ios::binary);
            if (!f) return;                                  // read-only scene directory: no cache, no error
            f.write(reinterpret_cast<const char *>(&h), sizeof(h));
            f.write(reinterpret_cast<const char *>(m_V.data()), (std::streamsize) m_V.size() * 4);
            f.write(reinterpret_cast<const char *>(m_N.data()), (std::streamsize) m_N.size() * 4);
            f.write(reinterpret_cast<const char *>(m_UV.data()), (std::streamsize) m_UV.size() * 4);
            f.write(reinterpret_cast<const char *>(m_F.data()), (std::streamsize) m_F.size() * 4);
            if (!f) { std::remove(tmp.c_str()); return; }
        }
        std::rename(tmp.c_str(), (filename + ".noricache").c_str());       // atomic: readers never see a partial file
    }
};
NORI_REGISTER_CLASS(WavefrontOBJ, "obj")

/// Mesh handed over as arrays by the embedding program (procedural scenes, e.g. the 10M-triangle stress mesh)
class ArrayMesh : public Mesh {
public:
    explicit ArrayMesh(const PropertyList &) {}
    void set(const std::string &name, const float *V, size_t nV, const uint32_t *F, size_t nF, const float *N, const float *UV) {
        m_name = name; m_V.assign(V, V + 3 * nV); m_F.assign(F, F + 3 * nF);
        if (N) m_N.assign(N, N + 3 * nV); if (UV) m_UV.assign(UV, UV + 2 * nV);
        for (size_t i = 0; i < nV; ++i) m_bbox.expandBy(vertex((uint32_t) i));
    }
};
NORI_REGISTER_CLASS(ArrayMesh, "array_mesh")
void arrayMeshSet(Mesh *m, const std::string &name, const float *V, size_t nV, const uint32_t *F, size_t nF, const float *N, const float *UV) {
    static_cast<ArrayMesh *>(m)->set(name, V, nV, F, nF, N, UV);
}

// ---- Sphere -----------------------------------------------------------------------------------------------
Sphere::Sphere(const PropertyList &propList) {
    m_position = propList.getPoint3("center", Point3f()); m_radius = propList.getFloat("radius", 1.f);
    m_bbox.expandBy(m_position - Vector3f(m_radius)); m_bbox.expandBy(m_position + Vector3f(m_radius));
}
std::string Sphere::toString() const { char b[256]; snprintf(b, sizeof(b), "Sphere[\n  center = [%f, %f, %f],\n  radius = %f\n]", m_position.x, m_position.y, m_position.z, m_radius); return b; }
NORI_REGISTER_CLASS(Sphere, "sphere")

// ---- BSDFs ------------------------------------------------------------------------------------------------
static std::string colorStr(const Color3f &c) { char b[96]; snprintf(b, sizeof(b), "[%f, %f, %f]", c.x, c.y, c.z); return b; }

// ---- images ------------------------------------------------------------------------------------------------
static bool readPnm(const std::string &path, int &cols, int &rows, int &channels, std::vector<uint8_t> &data) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    char magic[2] = {0, 0}; f.read(magic, 2);
    if (magic[0] != 'P' || (magic[1] != '5' && magic[1] != '6')) return false;
    auto nextInt = [&]() -> int {                             // header integers, '#' comments skipped
        int c = f.get();
        for (;;) { while (c == ' ' || c == '\t' || c == '\r' || c == '\n') c = f.get(); if (c == '#') { while (c != '\n' && c != EOF) c = f.get(); } else break; }
        int v = 0; bool any = false;
        while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); any = true; c = f.get(); }
        if (!any) throw NoriException::format("Image %s: malformed PNM header", path.c_str());
        return v;                                             // the single whitespace after the last header field is consumed
    };
    cols = nextInt(); rows = nextInt(); int maxv = nextInt();
    if (cols <= 0 || rows <= 0 || maxv != 255) throw NoriException::format("Image %s: only 8-bit PNM images are supported", path.c_str());
    channels = magic[1] == '6' ? 3 : 1;
    data.resize((size_t) cols * rows * channels);
    f.read(reinterpret_cast<char *>(data.data()), (std::streamsize) data.size());
    if ((size_t) f.gcount() != data.size()) throw NoriException::format("Image %s: truncated pixel data", path.c_str());
    return true;
}
void loadImageFile(const std::string &filename, int &cols, int &rows, int &channels, std::vector<uint8_t> &data) {
    for (const std::string &cand : {filename, filename + ".ppm", filename + ".pgm"})
        if (readPnm(getFileResolver()->resolve(cand), cols, rows, channels, data)) return;
    throw NoriException::format("Image %s could not be found! (this build decodes binary PNM only: place a pre-decoded %s.ppm or .pgm next to it)", filename.c_str(), filename.c_str());
}
static void imageCommon(const PropertyList &p, ImageData &img, uint32_t kind) {
    std::memset(&img.meta, 0, sizeof(img.meta));
    Point2f delta = p.getPoint2("delta", Point2f(0.f, 0.f)), scale = p.getVector2("scale", Point2f(1.f, 1.f));
    img.meta.kind = kind; img.meta.delta[0] = delta.x; img.meta.delta[1] = delta.y; img.meta.scale[0] = scale.x; img.meta.scale[1] = scale.y;
    // "lookup": "bilinear" = the CUDA fork's tex2D, "nearest" = the CPU renderer's truncating lookup (colour images)
    std::string lk = p.getString("lookup", "bilinear");
    if (lk == "bilinear") img.meta.lookup = NORI_HIP_LOOKUP_BILINEAR; else if (lk == "nearest") img.meta.lookup = NORI_HIP_LOOKUP_NEAREST;
    else throw NoriException::format("Unknown texture lookup \"%s\"", lk.c_str());
}
static void rgbaTexels(const std::string &filename, ImageData &img) {
    int cols, rows, ch; std::vector<uint8_t> px; loadImageFile(filename, cols, rows, ch, px);
    img.meta.width = (uint32_t) cols; img.meta.height = (uint32_t) rows; img.texels.resize((size_t) cols * rows * 4);
    for (size_t i = 0; i < (size_t) cols * rows; ++i) {       // cv::cvtColor(BGR2RGBA): alpha = 255
        uint8_t r = ch == 3 ? px[3 * i] : px[i], g = ch == 3 ? px[3 * i + 1] : px[i], b = ch == 3 ? px[3 * i + 2] : px[i];
        img.texels[4 * i] = r; img.texels[4 * i + 1] = g; img.texels[4 * i + 2] = b; img.texels[4 * i + 3] = 255;
    }
}

/// constant_color texture (src/consttexture.cpp)
class ConstantColorTexture : public ColorTexture {
public:
    explicit ConstantColorTexture(const PropertyList &p) { m_value = p.getColor("value", Color3f(0.f)); }
    Color3f constantValue() const override { return m_value; }
    std::string toString() const override { return "ConstantTexture[ " + colorStr(m_value) + " ]"; }
private:
    Color3f m_value;
};
NORI_REGISTER_CLASS(ConstantColorTexture, "constant_color")

/// The procedural and scalar textures of the CPU renderer.  They cross the C ABI as NoriHipTexture entries without an image: the few
/// constants they need travel in tex_data, and the device evaluates them per lookup (include/nori_hip.h, NORI_HIP_TEX_CHECKER_* / _IMAGE_FLOAT).
static void constantsAsTexels(ImageData &img, const float *v, uint32_t n) {
    img.meta.width = n; img.meta.height = 1; img.texels.resize((size_t) n * 4); std::memcpy(img.texels.data(), v, (size_t) n * 4);
}
static void proceduralCommon(const PropertyList &p, ImageData &img, uint32_t kind) {
    std::memset(&img.meta, 0, sizeof(img.meta));
    Point2f delta = p.getPoint2("delta", Point2f(0.f, 0.f)), scale = p.getVector2("scale", Point2f(1.f, 1.f));
    img.meta.kind = kind; img.meta.lookup = NORI_HIP_LOOKUP_NEAREST;
    img.meta.delta[0] = delta.x; img.meta.delta[1] = delta.y; img.meta.scale[0] = scale.x; img.meta.scale[1] = scale.y;
}
/// checkerboard_color (src/checkerboard.cpp:73-79, eval :32-55)
class CheckerboardColor : public ColorTexture {
public:
    explicit CheckerboardColor(const PropertyList &p) {
        proceduralCommon(p, m_img, NORI_HIP_TEX_CHECKER_COLOR);
        const Color3f a = p.getColor("value1", Color3f(0.f)), b = p.getColor("value2", Color3f(1.f));
        const float v[6] = {a.x, a.y, a.z, b.x, b.y, b.z};
        constantsAsTexels(m_img, v, 6);
    }
    const ImageData *image() const override { return &m_img; }
    std::string toString() const override { return "Checkerboard[]"; }
private:
    ImageData m_img;
};
NORI_REGISTER_CLASS(CheckerboardColor, "checkerboard_color")
/// checkerboard_float (src/checkerboard.cpp:65-71) and constant_float (src/consttexture.cpp:38-41: one value everywhere)
class CheckerboardFloat : public ColorTexture {
public:
    explicit CheckerboardFloat(const PropertyList &p, bool constant = false) {
        proceduralCommon(p, m_img, NORI_HIP_TEX_CHECKER_FLOAT);
        float v[2] = {p.getFloat("value1", 0.f), p.getFloat("value2", 1.f)};
        if (constant) v[0] = v[1] = p.getFloat("value", 0.f);
        constantsAsTexels(m_img, v, 2);
        m_constant = constant;
    }
    bool scalar() const override { return true; }
    const ImageData *image() const override { return &m_img; }
    std::string toString() const override { return m_constant ? "ConstantTexture[]" : "Checkerboard[]"; }
private:
    ImageData m_img; bool m_constant;
};
class ConstantFloatTexture : public CheckerboardFloat { public: explicit ConstantFloatTexture(const PropertyList &p) : CheckerboardFloat(p, true) {} };
NORI_REGISTER_CLASS(CheckerboardFloat, "checkerboard_float")
NORI_REGISTER_CLASS(ConstantFloatTexture, "constant_float")
/// image_float (src/imagetexture.cpp:26-46, 81-95): a grey image (cv::imread(..., GRAYSCALE)), truncating lookup, value / 255
class ImageFloatTexture : public ColorTexture {
public:
    explicit ImageFloatTexture(const PropertyList &p) {
        proceduralCommon(p, m_img, NORI_HIP_TEX_IMAGE_FLOAT);
        int cols, rows, ch; std::vector<uint8_t> px; loadImageFile(p.getString("filename"), cols, rows, ch, px);
        m_img.meta.width = (uint32_t) cols; m_img.meta.height = (uint32_t) rows; m_img.texels.resize((size_t) cols * rows * 4);
        for (size_t i = 0; i < (size_t) cols * rows; ++i) {
            const uint8_t g = ch == 1 ? px[i] : (uint8_t) ((px[3 * i] * 4899 + px[3 * i + 1] * 9617 + px[3 * i + 2] * 1868 + 8192) >> 14);   // (fixed-point BT.601 luma, as for bump maps)
            m_img.texels[4 * i] = m_img.texels[4 * i + 1] = m_img.texels[4 * i + 2] = g; m_img.texels[4 * i + 3] = 255;
        }
    }
    bool scalar() const override { return true; }
    const ImageData *image() const override { return &m_img; }
    std::string toString() const override { return "ImageTexture[]"; }
private:
    ImageData m_img;
};
NORI_REGISTER_CLASS(ImageFloatTexture, "image_float")

/// image_color texture (cuda/nori/imagetexture.cpp.cu:29-52; src/imagetexture.cpp:100-111)
class ImageColorTexture : public ColorTexture {
public:
    explicit ImageColorTexture(const PropertyList &p) {
        imageCommon(p, m_img, NORI_HIP_TEX_COLOR);
        rgbaTexels(p.getString("filename"), m_img);
    }
    const ImageData *image() const override { return &m_img; }
    std::string toString() const override { return "ImageTexture[]"; }
private:
    ImageData m_img;
};
NORI_REGISTER_CLASS(ImageColorTexture, "image_color")

/// sphereMap (cuda/nori/spheremap.cpp.cu:31-42): a lat-long image around the scene; takes image_color's properties (filename, scale,
/// delta, lookup) plus brightness_factor and origin (the polar axis, default +y)
class SphereMap : public EnvironmentMap {
public:
    explicit SphereMap(const PropertyList &p) {
        imageCommon(p, m_img, NORI_HIP_TEX_COLOR);
        rgbaTexels(p.getString("filename"), m_img);
        m_bright = p.getFloat("brightness_factor", 1.0f);
        m_origin = p.getVector3("origin", Vector3f(0.f, 1.f, 0.f));
        const float len = std::sqrt(m_origin.x * m_origin.x + m_origin.y * m_origin.y + m_origin.z * m_origin.z);
        if (!(len > 0.f)) throw NoriException("sphereMap: origin must not be the zero vector");
        m_origin = Vector3f(m_origin.x / len, m_origin.y / len, m_origin.z / len);
    }
    const ImageData *image() const override { return &m_img; }
    void describe(NoriHipEnvMap &out) const override {
        out.brightness = m_bright;
        // Frame(n): coordinateSystem (src/common.cpp:274-283)
        const Vector3f a = m_origin; Vector3f c;
        if (std::fabs(a.x) > std::fabs(a.y)) { float inv = 1.0f / std::sqrt(a.x * a.x + a.z * a.z); c = Vector3f(a.z * inv, 0.0f, -a.x * inv); }
        else { float inv = 1.0f / std::sqrt(a.y * a.y + a.z * a.z); c = Vector3f(0.0f, a.z * inv, -a.y * inv); }
        const Vector3f s(c.y * a.z - c.z * a.y, c.z * a.x - c.x * a.z, c.x * a.y - c.y * a.x);
        const float f[9] = {s.x, s.y, s.z, c.x, c.y, c.z, a.x, a.y, a.z};
        for (int i = 0; i < 9; ++i) out.frame[i] = f[i];
    }
    std::string toString() const override { return "EnviromentLight[]"; }
private:
    ImageData m_img; float m_bright; Vector3f m_origin;
};
NORI_REGISTER_CLASS(SphereMap, "sphereMap")

/// bump_map.  With a `filename` property it is the CUDA fork's (cuda/nori/bumpMap.cpp.cu:31-115): grey image -> Sobel gradients in x
/// and y, 16-bit signed.  Without one it is the CPU renderer's (src/bumpMap.cpp:9-61): a Texture<float> child named "bump_map"
/// whose value is assigned to the shading normal as it is (n = (v, v, v): that is what the reference's adjustNormal does).
class BumpMap : public NormalModifier {
public:
    ~BumpMap() override { delete m_child; }
    void addChild(NoriObject *obj) override {
        if (obj->getClassType() != ETexture) throw NoriException::format("BumpMap::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()).c_str());
        if (obj->getIdName() != "bump_map") throw NoriException("The name of this texture does not match any field!");
        if (m_child || m_fromFile) throw NoriException("There is already an bump_map defined!");
        ColorTexture *t = static_cast<ColorTexture *>(obj);
        if (!t->scalar()) throw NoriException("BumpMap: the bump_map texture must be a float texture (constant_float, checkerboard_float, image_float)");
        m_child = t;
    }
    void activate() override { if (!m_child && !m_fromFile) throw NoriException("There is no texture with name bump_map given!"); }
    explicit BumpMap(const PropertyList &p) {
        if (!p.has("filename")) { std::memset(&m_img.meta, 0, sizeof(m_img.meta)); return; }      // the CPU renderer's form: the texture comes as a child
        m_fromFile = true;
        imageCommon(p, m_img, NORI_HIP_TEX_BUMP);
        const float height = p.getFloat("height", 5), gradDelta = p.getFloat("grad_delta", 0);
        int cols, rows, ch; std::vector<uint8_t> px; loadImageFile(p.getString("filename"), cols, rows, ch, px);
        std::vector<uint8_t> grey((size_t) cols * rows);
        for (size_t i = 0; i < grey.size(); ++i)             // cv::imread(..., GRAYSCALE) of a colour file: fixed-point BT.601 luma
            grey[i] = ch == 1 ? px[i] : (uint8_t) ((px[3 * i] * 4899 + px[3 * i + 1] * 9617 + px[3 * i + 2] * 1868 + 8192) >> 14);
        m_img.meta.width = (uint32_t) cols; m_img.meta.height = (uint32_t) rows; m_img.texels.resize(grey.size() * 4);
        // cv::Sobel(image, grad, CV_16S, dx, dy, 3, height, grad_delta, BORDER_DEFAULT): 3x3 kernels, reflect-101 border,
        // result = saturate_cast<short>(sum * scale + delta)
        auto at = [&](int r, int c) -> int {
            if (rows > 1) { if (r < 0) r = -r; if (r >= rows) r = 2 * rows - 2 - r; } else r = 0;
            if (cols > 1) { if (c < 0) c = -c; if (c >= cols) c = 2 * cols - 2 - c; } else c = 0;
            return grey[(size_t) r * cols + c];
        };
        auto sat = [](float v) -> int16_t { long q = lrintf(v); return (int16_t) (q < -32768 ? -32768 : q > 32767 ? 32767 : q); };
        for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) {
            int gx = (at(r - 1, c + 1) + 2 * at(r, c + 1) + at(r + 1, c + 1)) - (at(r - 1, c - 1) + 2 * at(r, c - 1) + at(r + 1, c - 1));
            int gy = (at(r + 1, c - 1) + 2 * at(r + 1, c) + at(r + 1, c + 1)) - (at(r - 1, c - 1) + 2 * at(r - 1, c) + at(r - 1, c + 1));
            int16_t sx = sat((float) gx * height + gradDelta), sy = sat((float) gy * height + gradDelta);
            std::memcpy(&m_img.texels[((size_t) r * cols + c) * 4], &sx, 2); std::memcpy(&m_img.texels[((size_t) r * cols + c) * 4 + 2], &sy, 2);
        }
    }
    const ImageData *image() const override { return m_child ? m_child->image() : &m_img; }
    std::string toString() const override { return "BumpMap[]"; }
private:
    ImageData m_img; bool m_fromFile = false; ColorTexture *m_child = nullptr;
};
NORI_REGISTER_CLASS(BumpMap, "bump_map")

/// normal_map (cuda/nori/normalMap.cpp.cu:28-97)
class NormalMap : public NormalModifier {
public:
    explicit NormalMap(const PropertyList &p) {
        imageCommon(p, m_img, NORI_HIP_TEX_NORMAL);
        rgbaTexels(p.getString("filename"), m_img);
    }
    const ImageData *image() const override { return &m_img; }
    std::string toString() const override { return "NormalMap[]"; }
private:
    ImageData m_img;
};
NORI_REGISTER_CLASS(NormalMap, "normal_map")

/// What Diffuse and Subsurface share: the "albedo" slot, a colour property or a texture child
/// (src/diffuse.cpp:31-72, src/subsurface.cpp:32-88)
class AlbedoSlot {
public:
    ~AlbedoSlot() { delete m_tex; }
    void fromProps(const PropertyList &p) { if (p.has("albedo")) { m_value = p.getColor("albedo"); m_has = true; } }
    void addChild(NoriObject *obj, const char *cls) {
        if (obj->getClassType() != NoriObject::ETexture) throw NoriException::format("%s::addChild(<%s>) is not supported!", cls, NoriObject::classTypeName(obj->getClassType()).c_str());
        if (obj->getIdName() != "albedo") throw NoriException("The name of this texture does not match any field!");
        if (m_has) throw NoriException("There is already an albedo defined!");
        if (static_cast<ColorTexture *>(obj)->scalar()) throw NoriException::format("%s: the albedo texture must be a colour texture", cls);
        m_tex = static_cast<ColorTexture *>(obj); m_has = true;
        if (!m_tex->image()) m_value = m_tex->constantValue();
    }
    void activate() { if (!m_has) { m_value = Color3f(0.5f); m_has = true; } }
    void describe(NoriHipShape &o) const { o.albedo[0] = m_value.x; o.albedo[1] = m_value.y; o.albedo[2] = m_value.z; }
    const ImageData *image() const { return m_tex ? m_tex->image() : nullptr; }
    std::string toString() const { return image() ? std::string("ImageTexture[]") : "ConstantTexture[ " + colorStr(m_value) + " ]"; }
private:
    Color3f m_value; bool m_has = false; ColorTexture *m_tex = nullptr;
};

class Diffuse : public BSDF {
public:
    explicit Diffuse(const PropertyList &propList) { m_albedo.fromProps(propList); }
    void addChild(NoriObject *obj) override { m_albedo.addChild(obj, "Diffuse"); }
    void activate() override { m_albedo.activate(); }
    bool isDelta() const override { return false; }
    void describe(NoriHipShape &o) const override { o.bsdf = NORI_HIP_BSDF_DIFFUSE; m_albedo.describe(o); }
    const ImageData *albedoImage() const override { return m_albedo.image(); }
    std::string toString() const override { return "Diffuse[\n  albedo = " + m_albedo.toString() + "\n]"; }
private:
    AlbedoSlot m_albedo;
};
NORI_REGISTER_CLASS(Diffuse, "diffuse")

/// subsurface (src/subsurface.cpp:29-88): diffuse transmission whose exit point is found with a probe ray
class Subsurface : public BSDF {
public:
    explicit Subsurface(const PropertyList &p) {
        m_albedo.fromProps(p);
        m_intIOR = p.getFloat("intIOR", 1.3046f); m_extIOR = p.getFloat("extIOR", 1.000277f);
        m_sigmaT = p.getColor("sigma_t", Color3f(0.01f)); m_dmfp = p.getColor("dmfp", Color3f(0.1f));
    }
    void addChild(NoriObject *obj) override { m_albedo.addChild(obj, "Subsurface"); }
    void activate() override { m_albedo.activate(); }
    bool isDelta() const override { return false; }
    void describe(NoriHipShape &o) const override {
        o.bsdf = NORI_HIP_BSDF_SUBSURFACE; m_albedo.describe(o); o.int_ior = m_intIOR; o.ext_ior = m_extIOR;
        auto colorNorm = [](const Color3f &c) { return std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z); };     // src/subsurface.cpp:94-97
        o.ss_cs = colorNorm(Color3f(1.0f)); o.ss_cd = colorNorm(m_dmfp);                                     // s = 1 (:91)
    }
    const ImageData *albedoImage() const override { return m_albedo.image(); }
    std::string toString() const override { return "Subsurface[\n  albedo = " + m_albedo.toString() + "\n]"; }
private:
    AlbedoSlot m_albedo; float m_intIOR, m_extIOR; Color3f m_sigmaT, m_dmfp;
};
NORI_REGISTER_CLASS(Subsurface, "subsurface")

class Microfacet : public BSDF {
public:
    explicit Microfacet(const PropertyList &p) {
        m_alpha = p.getFloat("alpha", 0.1f); m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f);
        m_kd = p.getColor("kd", Color3f(0.5f)); m_ks = 1 - std::max(m_kd.x, std::max(m_kd.y, m_kd.z));
    }
    bool isDelta() const override { return false; }
    void describe(NoriHipShape &o) const override {
        o.bsdf = NORI_HIP_BSDF_MICROFACET; o.albedo[0] = m_kd.x; o.albedo[1] = m_kd.y; o.albedo[2] = m_kd.z;
        o.alpha = m_alpha; o.int_ior = m_intIOR; o.ext_ior = m_extIOR; o.ks = m_ks;
    }
    std::string toString() const override { char b[256]; snprintf(b, sizeof(b), "Microfacet[\n  alpha = %f,\n  intIOR = %f,\n  extIOR = %f,\n  kd = %s,\n  ks = %f\n]", m_alpha, m_intIOR, m_extIOR, colorStr(m_kd).c_str(), m_ks); return b; }
private:
    float m_alpha, m_intIOR, m_extIOR, m_ks; Color3f m_kd;
};
NORI_REGISTER_CLASS(Microfacet, "microfacet")

class Dielectric : public BSDF {
public:
    explicit Dielectric(const PropertyList &p) { m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); }
    bool isDelta() const override { return true; }
    void describe(NoriHipShape &o) const override { o.bsdf = NORI_HIP_BSDF_DIELECTRIC; o.int_ior = m_intIOR; o.ext_ior = m_extIOR; }
    std::string toString() const override { char b[128]; snprintf(b, sizeof(b), "Dielectric[\n  intIOR = %f,\n  extIOR = %f\n]", m_intIOR, m_extIOR); return b; }
private:
    float m_intIOR, m_extIOR;
};
NORI_REGISTER_CLASS(Dielectric, "dielectric")

class Mirror : public BSDF {
public:
    explicit Mirror(const PropertyList &) {}
    bool isDelta() const override { return true; }
    void describe(NoriHipShape &o) const override { o.bsdf = NORI_HIP_BSDF_MIRROR; }
    std::string toString() const override { return "Mirror[]"; }
};
NORI_REGISTER_CLASS(Mirror, "mirror")

// ---- area emitter -----------------------------------------------------------------------------------------
class AreaEmitter : public Emitter {
public:
    explicit AreaEmitter(const PropertyList &p) { m_radiance = p.getColor("radiance"); }
    Color3f getRadiance() const override { return m_radiance; }
    std::string toString() const override { return "AreaLight[\n  radiance = " + colorStr(m_radiance) + ",\n]"; }
private:
    Color3f m_radiance;
};
NORI_REGISTER_CLASS(AreaEmitter, "area")

// ---- reconstruction filters (src/rfilter.cpp:43-135): the radius sizes the film's border; eval() feeds the tabulated
// filter of the optional splat mode (nori_hip_set_film_filter) ------------------------------------------------------
class GaussianFilter : public ReconstructionFilter {
public:
    explicit GaussianFilter(const PropertyList &p) { m_radius = p.getFloat("radius", 2.0f); m_stddev = p.getFloat("stddev", 0.5f); }
    float eval(float x) const override {
        float alpha = -1.0f / (2.0f * m_stddev * m_stddev);
        return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * m_radius * m_radius));
    }
    std::string toString() const override { char b[96]; snprintf(b, sizeof(b), "GaussianFilter[radius=%f, stddev=%f]", m_radius, m_stddev); return b; }
private:
    float m_stddev;
};
class MitchellNetravaliFilter : public ReconstructionFilter {
public:
    explicit MitchellNetravaliFilter(const PropertyList &p) { m_radius = p.getFloat("radius", 2.0f); m_B = p.getFloat("B", 1.0f / 3.0f); m_C = p.getFloat("C", 1.0f / 3.0f); }
    float eval(float x) const override {
        x = std::abs(2.0f * x / m_radius);
        float x2 = x * x, x3 = x2 * x;
        if (x < 1) return 1.0f / 6.0f * ((12 - 9 * m_B - 6 * m_C) * x3 + (-18 + 12 * m_B + 6 * m_C) * x2 + (6 - 2 * m_B));
        else if (x < 2) return 1.0f / 6.0f * ((-m_B - 6 * m_C) * x3 + (6 * m_B + 30 * m_C) * x2 + (-12 * m_B - 48 * m_C) * x + (8 * m_B + 24 * m_C));
        return 0.0f;
    }
    std::string toString() const override { char b[128]; snprintf(b, sizeof(b), "MitchellNetravaliFilter[radius=%f, B=%f, C=%f]", m_radius, m_B, m_C); return b; }
private:
    float m_B, m_C;
};
class TentFilter : public ReconstructionFilter {
public:
    explicit TentFilter(const PropertyList &) { m_radius = 1.0f; }
    float eval(float x) const override { return std::max(0.0f, 1.0f - std::abs(x)); }
    std::string toString() const override { return "TentFilter[]"; }
};
class BoxFilter : public ReconstructionFilter {
public:
    explicit BoxFilter(const PropertyList &) { m_radius = 0.5f; }
    float eval(float) const override { return 1.0f; }
    std::string toString() const override { return "BoxFilter[]"; }
};
NORI_REGISTER_CLASS(GaussianFilter, "gaussian")
NORI_REGISTER_CLASS(MitchellNetravaliFilter, "mitchell")
NORI_REGISTER_CLASS(TentFilter, "tent")
NORI_REGISTER_CLASS(BoxFilter, "box")

// ---- camera -----------------------------------------------------------------------------------------------
class PerspectiveCamera : public Camera {
public:
    explicit PerspectiveCamera(const PropertyList &p) {
        m_width = p.getInteger("width", 1280); m_height = p.getInteger("height", 720);
        m_cameraToWorld = p.getTransform("toWorld", Transform());
        m_fov = p.getFloat("fov", 30.0f); m_nearClip = p.getFloat("nearClip", 1e-4f); m_farClip = p.getFloat("farClip", 1e4f);
    }
    ~PerspectiveCamera() override { delete m_rfilter; }
    void addChild(NoriObject *obj) override {
        if (obj->getClassType() != EReconstructionFilter) throw NoriException::format("Camera::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()).c_str());
        if (m_rfilter) throw NoriException("Camera: tried to register multiple reconstruction filters!");
        m_rfilter = static_cast<ReconstructionFilter *>(obj);
    }
    void activate() override {
        // src/perspective.cpp:53-80: sampleToCamera = (Diag(0.5, -0.5*aspect, 1) * Translate(1, -1/aspect, 0) * perspective)^-1
        float aspect = m_width / (float) m_height;
        float recip = 1.0f / (m_farClip - m_nearClip), cot = 1.0f / std::tan(degToRad(m_fov / 2.0f));
        float persp[4][4] = {{cot, 0, 0, 0}, {0, cot, 0, 0}, {0, 0, m_farClip * recip, -m_nearClip * m_farClip * recip}, {0, 0, 1, 0}};
        float sx = 0.5f, sy = -0.5f * aspect, sz = 1.0f, tx = 1.0f, ty = -1.0f / aspect, tz = 0.0f;
        float st[4][4] = {{sx, 0, 0, sx * tx}, {0, sy, 0, sy * ty}, {0, 0, sz, sz * tz}, {0, 0, 0, 1}};
        float prod[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = st[i][0] * persp[0][j]; for (int k = 1; k < 4; ++k) s += st[i][k] * persp[k][j]; prod[i][j] = s; }
        m_sampleToCamera = Transform(prod).inverse();
        if (!m_rfilter) { m_rfilter = static_cast<ReconstructionFilter *>(NoriObjectFactory::createInstance("gaussian", PropertyList())); m_rfilter->activate(); }
    }
    void describe(NoriHipCamera &o) const override {
        std::memcpy(o.sample_to_camera, m_sampleToCamera.m, 64); std::memcpy(o.camera_to_world, m_cameraToWorld.m, 64);
        o.width = m_width; o.height = m_height; o.near_clip = m_nearClip; o.far_clip = m_farClip; o.filter_radius = m_rfilter->getRadius();
    }
    std::string toString() const override { char b[256]; snprintf(b, sizeof(b), "PerspectiveCamera[\n  outputSize = [%d, %d],\n  fov = %f,\n  clip = [%f, %f],\n  rfilter = %s\n]", m_width, m_height, m_fov, m_nearClip, m_farClip, m_rfilter ? m_rfilter->toString().c_str() : "null"); return b; }
private:
    Transform m_sampleToCamera, m_cameraToWorld; float m_fov, m_nearClip, m_farClip;
};
NORI_REGISTER_CLASS(PerspectiveCamera, "perspective")

// ---- sampler ----------------------------------------------------------------------------------------------
/// "independent": pcg32 streams.  On the GPU path every camera sample owns its stream:
/// pcg32.seed(initstate = y*width + x, initseq = sample index); the draw order within a sample is the reference's.
class Independent : public Sampler {
public:
    explicit Independent(const PropertyList &p) { m_sampleCount = (size_t) p.getInteger("sampleCount", 1); }
    std::string toString() const override { char b[64]; snprintf(b, sizeof(b), "Independent[sampleCount=%zu]", m_sampleCount); return b; }
};
NORI_REGISTER_CLASS(Independent, "independent")

// ---- scene ------------------------------------------------------------------------------------------------
Scene::Scene(const PropertyList &) { m_bvh = new BVH(); }
Scene::~Scene() { delete m_bvh; delete m_sampler; delete m_camera; delete m_integrator; delete m_envmap; for (auto e : m_emitters) delete e; }
void Scene::activate() {
    if (!m_integrator) throw NoriException("No integrator was specified!");
    if (!m_integrator->buildsOwnBVH()) m_bvh->build();      // src/scene.cpp:42-43; otherwise the device builds a tree at preprocess()
    if (!m_camera) throw NoriException("No camera was specified!");
    if (!m_sampler) { m_sampler = static_cast<Sampler *>(NoriObjectFactory::createInstance("independent", PropertyList())); m_sampler->activate(); }
}
void Scene::addChild(NoriObject *obj) {
    switch (obj->getClassType()) {
        case EMesh: { Shape *mesh = static_cast<Shape *>(obj); m_bvh->addShape(mesh); m_shapes.push_back(mesh); if (mesh->isEmitter()) m_emitters.push_back(mesh->getEmitter()); } break;
        case EEmitter: m_emitters.push_back(static_cast<Emitter *>(obj)); break;
        case ESampler: if (m_sampler) throw NoriException("There can only be one sampler per scene!"); m_sampler = static_cast<Sampler *>(obj); break;
        case ECamera: if (m_camera) throw NoriException("There can only be one camera per scene!"); m_camera = static_cast<Camera *>(obj); break;
        case EIntegrator: if (m_integrator) throw NoriException("There can only be one integrator per scene!"); m_integrator = static_cast<Integrator *>(obj); break;
        case EEnvMap: if (m_envmap) throw NoriException("There can only be one Enviroment map per scene"); m_envmap = static_cast<EnvironmentMap *>(obj); break;   // cuda/nori/scene.cpp:116-120
        default: throw NoriException::format("Scene::addChild(<%s>) is not supported!", classTypeName(obj->getClassType()).c_str());
    }
}
std::string Scene::toString() const {
    std::string shapes, lights;
    for (size_t i = 0; i < m_shapes.size(); ++i) { shapes += "  " + indent(m_shapes[i]->toString(), 2); if (i + 1 < m_shapes.size()) shapes += ","; shapes += "\n"; }
    for (size_t i = 0; i < m_emitters.size(); ++i) { lights += "  " + indent(m_emitters[i]->toString(), 2); if (i + 1 < m_emitters.size()) lights += ","; lights += "\n"; }
    return "Scene[\n  integrator = " + indent(m_integrator->toString()) + ",\n  sampler = " + indent(m_sampler->toString()) + "\n  camera = " + indent(m_camera->toString()) +
           ",\n  shapes = {\n  " + indent(shapes, 2) + "  }\n  emitters = {\n  " + indent(lights, 2) + "  }\n]";
}
NORI_REGISTER_CLASS(Scene, "scene")

} // namespace nori
