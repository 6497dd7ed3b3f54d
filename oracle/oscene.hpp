// ORACLE — TEST INFRASTRUCTURE ONLY.
// Scene description + loader restating the reference's XML parser, OBJ loader, mesh activation and camera set-up.
//
// Reference: src/parser.cpp:28-340 (tags, transform composition order, property parsing),
//            src/obj.cpp:32-132 (OBJ → indexed mesh, quad split, (p,uv,n) de-duplication),
//            src/mesh.cpp:30-38,75-81,170-182 (area CDF, surfaceArea, per-triangle bbox / centroid),
//            src/sphere.cpp:30-40, src/perspective.cpp:34-88, src/scene.cpp:62-99 (child registration order),
//            src/diffuse.cpp:31-72, src/microfacet.cpp:28-50, src/dielectric.cpp:28-34, src/arealight.cpp:29-31,
//            src/independent.cpp:35-37, src/ttest.cpp:60-77, src/subsurface.cpp:32-50,
//            cuda/nori/imagetexture.cpp.cu:29-52, cuda/nori/bumpMap.cpp.cu:31-115, cuda/nori/normalMap.cpp.cu:28-97 (images:
//            "parity unpinned" — the reference decodes with OpenCV, which is not available; see oracle/README.md).
#pragma once
#include "omath.hpp"
#include "oxml.hpp"
#include <map>
#include <unordered_map>
#include <cstring>
#include <cstdlib>

namespace orc {

// ---- string helpers (src/common.cpp:88-143) --------------------------------------------------------------
inline std::vector<std::string> tokenize(const std::string &s, const std::string &delim = ", ", bool includeEmpty = false) {
    std::string::size_type lastPos = 0, pos = s.find_first_of(delim, lastPos);
    std::vector<std::string> tokens;
    while (lastPos != std::string::npos) {
        if (pos != lastPos || includeEmpty) tokens.push_back(s.substr(lastPos, pos - lastPos));
        lastPos = pos;
        if (lastPos != std::string::npos) { lastPos += 1; pos = s.find_first_of(delim, lastPos); }
    }
    return tokens;
}
inline float to_float(const std::string &str) {
    char *end = nullptr; float r = strtof(str.c_str(), &end);
    if (*end != '\0') throw std::runtime_error("Could not parse floating point value \"" + str + "\"");
    return r;
}
inline int to_int(const std::string &str) {
    char *end = nullptr; int r = (int) strtol(str.c_str(), &end, 10);
    if (*end != '\0') throw std::runtime_error("Could not parse integer value \"" + str + "\"");
    return r;
}
inline unsigned to_uint(const std::string &str) {
    char *end = nullptr; unsigned r = (int) strtoul(str.c_str(), &end, 10);
    if (*end != '\0') throw std::runtime_error("Could not parse integer value \"" + str + "\"");
    return r;
}
inline V3 to_v3(const std::string &s) {
    auto t = tokenize(s); if (t.size() != 3) throw std::runtime_error("Expected 3 values");
    return mk3(to_float(t[0]), to_float(t[1]), to_float(t[2]));
}
inline float deg_to_rad(float v) { return v * (kPi / 180.0f); }

// ---- property list ---------------------------------------------------------------------------------------
struct Props {
    std::map<std::string, float> f; std::map<std::string, int> i; std::map<std::string, std::string> s;
    std::map<std::string, V3> v; std::map<std::string, Transform> t; std::map<std::string, bool> b; std::map<std::string, V2> v2;
    V2 getv2(const char *k, V2 d) const { auto it = v2.find(k); return it == v2.end() ? d : it->second; }
    float getf(const char *k, float d) const { auto it = f.find(k); return it == f.end() ? d : it->second; }
    int geti(const char *k, int d) const { auto it = i.find(k); return it == i.end() ? d : it->second; }
    V3 getv(const char *k, V3 d) const { auto it = v.find(k); return it == v.end() ? d : it->second; }
    bool hasv(const char *k) const { return v.count(k) != 0; }
    bool hass(const char *k) const { return s.count(k) != 0; }
    std::string gets(const char *k) const { auto it = s.find(k); if (it == s.end()) throw std::runtime_error(std::string("Property '") + k + "' is missing!"); return it->second; }
    std::string gets(const char *k, const std::string &d) const { auto it = s.find(k); return it == s.end() ? d : it->second; }
    Transform gett(const char *k) const { auto it = t.find(k); return it == t.end() ? Transform() : it->second; }
};

// ---- objects ---------------------------------------------------------------------------------------------
enum BsdfType { BSDF_DIFFUSE = 0, BSDF_MICROFACET = 1, BSDF_DIELECTRIC = 2, BSDF_MIRROR = 3, BSDF_SUBSURFACE = 4 };
struct Bsdf {
    int type = BSDF_DIFFUSE;
    V3 albedo = mk3(0.5f);                 // diffuse / subsurface (src/diffuse.cpp:65-72 default)
    int albedo_tex = -1;                   // index into Scene::images when the albedo is an image_color texture
    float alpha = 0.1f, intIOR = 1.5046f, extIOR = 1.000277f, ks = 0.5f;
    V3 kd = mk3(0.5f);
    V3 dmfp = mk3(0.1f);                   // subsurface (src/subsurface.cpp:44-46); s = 1 (:91)
};
// An image as the CUDA fork uploads it: RGBA bytes (colour, normal map) or the two 16-bit Sobel gradients (bump map)
// ... or one of the CPU renderer's procedural / scalar textures (src/checkerboard.cpp, src/consttexture.cpp, src/imagetexture.cpp:26-46)
enum ImageKind { IMG_COLOR = 0, IMG_BUMP = 1, IMG_NORMAL = 2, IMG_CHECKER_COLOR = 3, IMG_CHECKER_FLOAT = 4, IMG_IMAGE_FLOAT = 5 };
struct Image {
    int kind = IMG_COLOR; bool bilinear = true;
    float values[6] = {0, 0, 0, 0, 0, 0};  // IMG_CHECKER_COLOR: value1 rgb, value2 rgb; IMG_CHECKER_FLOAT: value1, value2
    int cols = 0, rows = 0;
    std::vector<uint8_t> rgba;             // IMG_COLOR / IMG_NORMAL: 4 per texel
    std::vector<int16_t> gx, gy;           // IMG_BUMP
    V2 scale = mk2(1.f, 1.f), delta = mk2(0.f, 0.f);
};
struct Emitter { V3 radiance; int shape = -1; };

enum ShapeType { SHAPE_MESH = 0, SHAPE_SPHERE = 1 };
struct Shape {
    int type = SHAPE_MESH;
    std::string name;
    std::vector<V3> V, N; std::vector<V2> UV; std::vector<uint32_t> F;   // F: 3 per triangle
    BBox bbox;
    V3 center = mk3(0.f); float radius = 1.f;                            // sphere
    Bsdf bsdf; bool has_bsdf = false;
    int normal_tex = -1;                                                 // index into Scene::images (bump_map / normal_map)
    int emitter = -1;                                                    // index into Scene::emitters
    std::vector<float> cdf; float cdf_sum = 0, cdf_norm = 0;             // DiscretePDF (include/nori/dpdf.h)
    uint32_t prim_count() const { return type == SHAPE_MESH ? (uint32_t) (F.size() / 3) : 1u; }

    // src/mesh.cpp:75-81
    float surface_area(uint32_t i) const {
        V3 p0 = V[F[3 * i]], p1 = V[F[3 * i + 1]], p2 = V[F[3 * i + 2]];
        return 0.5f * norm(cross(p1 - p0, p2 - p0));
    }
    // src/mesh.cpp:170-175 / src/sphere.cpp:38
    BBox prim_bbox(uint32_t i) const {
        if (type == SHAPE_SPHERE) return bbox;
        BBox r(V[F[3 * i]]); r.expand(V[F[3 * i + 1]]); r.expand(V[F[3 * i + 2]]); return r;
    }
    // src/mesh.cpp:177-182 / src/sphere.cpp:40
    V3 prim_centroid(uint32_t i) const {
        if (type == SHAPE_SPHERE) return center;
        return (1.0f / 3.0f) * ((V[F[3 * i]] + V[F[3 * i + 1]]) + V[F[3 * i + 2]]);
    }
    // src/mesh.cpp:30-38 + dpdf.h:55-57,97-109
    void activate() {
        if (type != SHAPE_MESH) return;
        cdf.clear(); cdf.push_back(0.0f);
        for (uint32_t i = 0; i < prim_count(); ++i) cdf.push_back(cdf.back() + surface_area(i));
        cdf_sum = cdf.back();
        if (cdf_sum > 0) {
            cdf_norm = 1.0f / cdf_sum;
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= cdf_norm;
            cdf.back() = 1.0f;
        } else cdf_norm = 0.0f;
    }
};

struct Camera {
    int width = 1280, height = 720;
    float fov = 30.0f, nearClip = 1e-4f, farClip = 1e4f;
    Transform cameraToWorld, sampleToCamera;
    float invW = 0, invH = 0;
    float rfilter_radius = 2.0f;           // default gaussian (src/perspective.cpp:83-87, src/rfilter.cpp:47)
    // the reconstruction filter itself (src/rfilter.cpp:43-135): 0 gaussian (stddev), 1 mitchell (B, C), 2 tent, 3 box
    int rfilter_kind = 0; float rfilter_p0 = 0.5f, rfilter_p1 = 0.f;
    float rfilter_eval(float x) const {
        const float r = rfilter_radius;
        switch (rfilter_kind) {
        case 0: { float alpha = -1.0f / (2.0f * rfilter_p0 * rfilter_p0); return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * r * r)); }
        case 1: {
            const float B = rfilter_p0, C = rfilter_p1;
            x = std::abs(2.0f * x / r); float x2 = x * x, x3 = x2 * x;
            if (x < 1) return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
            else if (x < 2) return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
            return 0.0f;
        }
        case 2: return std::max(0.0f, 1.0f - std::abs(x));
        default: return 1.0f;
        }
    }
    // src/perspective.cpp:53-80
    void activate() {
        invW = 1.0f / (float) width; invH = 1.0f / (float) height;
        float aspect = width / (float) height;
        float recip = 1.0f / (farClip - nearClip), cot = 1.0f / std::tan(deg_to_rad(fov / 2.0f));
        M4 persp; std::memset(&persp, 0, sizeof(persp));
        persp.m[0][0] = cot; persp.m[1][1] = cot;
        persp.m[2][2] = farClip * recip; persp.m[2][3] = -nearClip * farClip * recip;
        persp.m[3][2] = 1;
        // Diagonal(0.5, -0.5*aspect, 1) * Translation(1, -1/aspect, 0)
        M4 st = M4::identity();
        float sx = 0.5f, sy = -0.5f * aspect, sz = 1.0f;
        float tx = 1.0f, ty = -1.0f / aspect, tz = 0.0f;
        st.m[0][0] = sx; st.m[1][1] = sy; st.m[2][2] = sz;
        st.m[0][3] = sx * tx; st.m[1][3] = sy * ty; st.m[2][3] = sz * tz;
        sampleToCamera = Transform(mul(st, persp)).inverse_t();
    }
    // src/perspective.cpp:90-111
    void sample_ray(Ray &ray, V2 samplePosition) const {
        V3 nearP = sampleToCamera.apply_point(mk3(samplePosition.x * invW, samplePosition.y * invH, 0.0f));
        V3 d = normalized(nearP);
        float invZ = 1.0f / d.z;
        ray.o = cameraToWorld.apply_point(mk3(0, 0, 0));
        ray.d = cameraToWorld.apply_vector(d);
        ray.mint = nearClip * invZ;
        ray.maxt = farClip * invZ;
        ray.update();
    }
};

struct Scene {
    std::vector<Shape> shapes;
    std::vector<Emitter> emitters;
    std::vector<Image> images;
    Camera camera;
    int sample_count = 1;
    std::string integrator;
    // filled by build_accel() (obvh.hpp)
    std::vector<uint32_t> shape_offset;       // m_shapeOffset (src/bvh.cpp:308-312)
    BBox bbox;
    uint32_t prim_total() const { return shape_offset.empty() ? 0 : shape_offset.back(); }
};

struct TestSuite {                  // <test type="ttest"> (src/ttest.cpp)
    std::vector<Scene> scenes;
    std::vector<float> references;
    float significance = 0.01f;
    int sample_count = 100000;
};

// ---- transform ops, composed left-multiplied in XML order (src/parser.cpp:278-324) ----------------------
struct Affine {            // Eigen::Affine3f: linear 3x3 + translation
    float l[3][3]; float t[3];
    void set_identity() { for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) l[i][j] = (i == j); t[i] = 0; } }
    M4 matrix() const {
        M4 m = M4::identity();
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m.m[i][j] = l[i][j]; m.m[i][3] = t[i]; }
        return m;
    }
};
inline void premul_linear(Affine &a, const float L[3][3], const float T[3]) {
    // res.linear = L * a.linear ; res.translation = L * a.translation + T  (sequential sums)
    Affine r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.l[i][j] = (L[i][0] * a.l[0][j] + L[i][1] * a.l[1][j]) + L[i][2] * a.l[2][j];
        r.t[i] = ((L[i][0] * a.t[0] + L[i][1] * a.t[1]) + L[i][2] * a.t[2]) + T[i];
    }
    a = r;
}
inline void apply_transform_op(Affine &tr, const XmlNode &n) {
    const std::string &tag = n.name;
    if (tag == "translate") {
        V3 v = to_v3(n.attr("value")); tr.t[0] += v.x; tr.t[1] += v.y; tr.t[2] += v.z;
    } else if (tag == "scale") {
        V3 v = to_v3(n.attr("value"));
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) tr.l[i][j] = v[i] * tr.l[i][j]; tr.t[i] = v[i] * tr.t[i]; }
    } else if (tag == "rotate") {
        float angle = deg_to_rad(to_float(n.attr("angle")));
        V3 ax = to_v3(n.attr("axis"));
        // Eigen AngleAxis::toRotationMatrix
        V3 sin_axis = std::sin(angle) * ax; float c = std::cos(angle); V3 cos1_axis = (1.0f - c) * ax;
        float R[3][3]; float tmp;
        tmp = cos1_axis.x * ax.y; R[0][1] = tmp - sin_axis.z; R[1][0] = tmp + sin_axis.z;
        tmp = cos1_axis.x * ax.z; R[0][2] = tmp + sin_axis.y; R[2][0] = tmp - sin_axis.y;
        tmp = cos1_axis.y * ax.z; R[1][2] = tmp - sin_axis.x; R[2][1] = tmp + sin_axis.x;
        R[0][0] = cos1_axis.x * ax.x + c; R[1][1] = cos1_axis.y * ax.y + c; R[2][2] = cos1_axis.z * ax.z + c;
        // rotation * Affine: no added translation
        Affine r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r.l[i][j] = (R[i][0] * tr.l[0][j] + R[i][1] * tr.l[1][j]) + R[i][2] * tr.l[2][j];
            r.t[i] = (R[i][0] * tr.t[0] + R[i][1] * tr.t[1]) + R[i][2] * tr.t[2];
        }
        tr = r;
    } else if (tag == "lookat") {
        V3 origin = to_v3(n.attr("origin")), target = to_v3(n.attr("target")), up = to_v3(n.attr("up"));
        V3 dir = normalized(target - origin);
        V3 left = normalized(cross(normalized(up), dir));
        V3 newUp = normalized(cross(dir, left));
        float L[3][3] = {{left.x, newUp.x, dir.x}, {left.y, newUp.y, dir.y}, {left.z, newUp.z, dir.z}};
        float T[3] = {origin.x, origin.y, origin.z};
        premul_linear(tr, L, T);
    } else if (tag == "matrix") {
        auto tok = tokenize(n.attr("value")); if (tok.size() != 16) throw std::runtime_error("Expected 16 values");
        float L[3][3], T[3];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) L[i][j] = to_float(tok[i * 4 + j]); T[i] = to_float(tok[i * 4 + 3]); }
        premul_linear(tr, L, T);
    } else throw std::runtime_error("Unhandled transform element \"" + tag + "\"");
}

// ---- OBJ loader (src/obj.cpp:32-132) --------------------------------------------------------------------
struct ObjVertex { uint32_t p = (uint32_t) -1, n = (uint32_t) -1, uv = (uint32_t) -1;
    bool operator==(const ObjVertex &v) const { return v.p == p && v.n == n && v.uv == uv; } };
struct ObjVertexHash { size_t operator()(const ObjVertex &v) const {
    size_t h = std::hash<uint32_t>()(v.p); h = h * 37 + std::hash<uint32_t>()(v.uv); h = h * 37 + std::hash<uint32_t>()(v.n); return h; } };
inline ObjVertex parse_obj_vertex(const std::string &s) {
    auto tk = tokenize(s, "/", true);
    if (tk.size() < 1 || tk.size() > 3) throw std::runtime_error("Invalid vertex data: \"" + s + "\"");
    ObjVertex v; v.p = to_uint(tk[0]);
    if (tk.size() >= 2 && !tk[1].empty()) v.uv = to_uint(tk[1]);
    if (tk.size() >= 3 && !tk[2].empty()) v.n = to_uint(tk[2]);
    return v;
}
inline void load_obj(Shape &sh, const std::string &filename, const Transform &trafo) {
    std::ifstream is(filename);
    if (is.fail()) throw std::runtime_error("Unable to open OBJ file \"" + filename + "\"!");
    std::vector<V3> positions, normals; std::vector<V2> texcoords;
    std::vector<uint32_t> indices; std::vector<ObjVertex> vertices;
    std::unordered_map<ObjVertex, uint32_t, ObjVertexHash> vmap;
    std::string line_str;
    while (std::getline(is, line_str)) {
        std::istringstream line(line_str);
        std::string prefix; line >> prefix;
        if (prefix == "v") {
            V3 p = mk3(0.f); line >> p.x >> p.y >> p.z;
            p = trafo.apply_point(p); sh.bbox.expand(p); positions.push_back(p);
        } else if (prefix == "vt") {
            V2 tc = mk2(0.f, 0.f); line >> tc.x >> tc.y; texcoords.push_back(tc);
        } else if (prefix == "vn") {
            V3 n = mk3(0.f); line >> n.x >> n.y >> n.z; normals.push_back(normalized(trafo.apply_normal(n)));
        } else if (prefix == "f") {
            std::string v1, v2, v3, v4; line >> v1 >> v2 >> v3 >> v4;
            ObjVertex verts[6]; int nV = 3;
            verts[0] = parse_obj_vertex(v1); verts[1] = parse_obj_vertex(v2); verts[2] = parse_obj_vertex(v3);
            if (!v4.empty()) { verts[3] = parse_obj_vertex(v4); verts[4] = verts[0]; verts[5] = verts[2]; nV = 6; }
            for (int i = 0; i < nV; ++i) {
                auto it = vmap.find(verts[i]);
                if (it == vmap.end()) { vmap[verts[i]] = (uint32_t) vertices.size(); indices.push_back((uint32_t) vertices.size()); vertices.push_back(verts[i]); }
                else indices.push_back(it->second);
            }
        }
    }
    sh.F = indices;
    sh.V.resize(vertices.size());
    for (size_t i = 0; i < vertices.size(); ++i) sh.V[i] = positions.at(vertices[i].p - 1);
    if (!normals.empty()) { sh.N.resize(vertices.size()); for (size_t i = 0; i < vertices.size(); ++i) sh.N[i] = normals.at(vertices[i].n - 1); }
    if (!texcoords.empty()) { sh.UV.resize(vertices.size()); for (size_t i = 0; i < vertices.size(); ++i) sh.UV[i] = texcoords.at(vertices[i].uv - 1); }
    sh.name = filename;
}

// ---- XML → Scene ----------------------------------------------------------------------------------------
inline void parse_props(const XmlNode &node, Props &pl) {
    for (auto &chp : node.children) {
        const XmlNode &ch = *chp; const std::string &t = ch.name;
        if (t == "float") pl.f[ch.attr("name")] = to_float(ch.attr("value"));
        else if (t == "integer") pl.i[ch.attr("name")] = to_int(ch.attr("value"));
        else if (t == "string") pl.s[ch.attr("name")] = ch.attr("value");
        else if (t == "boolean") { std::string v = ch.attr("value"); for (auto &c : v) c = (char) tolower(c); pl.b[ch.attr("name")] = (v == "true"); }
        else if ((t == "point" || t == "vector") && tokenize(ch.attr("value")).size() == 2) {      // src/parser.cpp:242-266
            auto tk = tokenize(ch.attr("value")); pl.v2[ch.attr("name")] = mk2(to_float(tk[0]), to_float(tk[1]));
        }
        else if (t == "color" || t == "point" || t == "vector") pl.v[ch.attr("name")] = to_v3(ch.attr("value"));
        else if (t == "transform") {
            Affine a; a.set_identity();
            for (auto &op : ch.children) apply_transform_op(a, *op);
            pl.t[ch.attr("name")] = Transform(a.matrix());
        }
    }
}
// ---- images: binary PNM (P5/P6); for other formats the pre-decoded "<file>.ppm" / "<file>.pgm" next to it ----
inline bool read_pnm(const std::string &path, int &cols, int &rows, int &ch, std::vector<uint8_t> &data) {
    std::ifstream f(path, std::ios::binary); if (!f) return false;
    std::string magic; f >> magic; if (magic != "P5" && magic != "P6") return false;
    auto next_int = [&]() { for (;;) { f >> std::ws; if (f.peek() == '#') { std::string l; std::getline(f, l); } else break; } int v = 0; f >> v; return v; };
    cols = next_int(); rows = next_int(); int maxv = next_int(); f.get();
    if (cols <= 0 || rows <= 0 || maxv != 255) throw std::runtime_error("oracle: unsupported PNM " + path);
    ch = magic == "P6" ? 3 : 1; data.resize((size_t) cols * rows * ch);
    f.read((char *) data.data(), (std::streamsize) data.size());
    if ((size_t) f.gcount() != data.size()) throw std::runtime_error("oracle: truncated PNM " + path);
    return true;
}
inline void load_image_file(const std::string &path, int &cols, int &rows, int &ch, std::vector<uint8_t> &data) {
    if (read_pnm(path, cols, rows, ch, data) || read_pnm(path + ".ppm", cols, rows, ch, data) || read_pnm(path + ".pgm", cols, rows, ch, data)) return;
    throw std::runtime_error("Image " + path + " could not be found!");
}
inline Image load_image(const XmlNode &n, const std::string &dir, int kind) {
    Props pl; parse_props(n, pl);
    Image im; im.kind = kind; im.delta = pl.getv2("delta", mk2(0.f, 0.f)); im.scale = pl.getv2("scale", mk2(1.f, 1.f));
    im.bilinear = pl.gets("lookup", "bilinear") != "nearest";
    int ch; std::vector<uint8_t> px; load_image_file(dir + "/" + pl.gets("filename"), im.cols, im.rows, ch, px);
    const size_t N = (size_t) im.cols * im.rows;
    if (kind == IMG_IMAGE_FLOAT) {          // cv::imread(..., GRAYSCALE): fixed-point BT.601 luma of a colour file
        im.rgba.resize(N * 4);
        for (size_t i = 0; i < N; ++i) {
            const uint8_t g = ch == 1 ? px[i] : (uint8_t) ((px[3 * i] * 4899 + px[3 * i + 1] * 9617 + px[3 * i + 2] * 1868 + 8192) >> 14);
            im.rgba[4 * i] = im.rgba[4 * i + 1] = im.rgba[4 * i + 2] = g; im.rgba[4 * i + 3] = 255;
        }
        return im;
    }
    if (kind != IMG_BUMP) {
        im.rgba.resize(N * 4);
        for (size_t i = 0; i < N; ++i) for (int c = 0; c < 4; ++c) im.rgba[4 * i + c] = c == 3 ? 255 : (ch == 3 ? px[3 * i + c] : px[i]);
        return im;
    }
    // grey image, then cv::Sobel(CV_16S, ksize 3, scale = height, delta = grad_delta, BORDER_REFLECT_101) in x and in y
    float height = pl.getf("height", 5), grad_delta = pl.getf("grad_delta", 0);
    std::vector<int> g(N);
    for (size_t i = 0; i < N; ++i) g[i] = ch == 1 ? px[i] : ((px[3 * i] * 4899 + px[3 * i + 1] * 9617 + px[3 * i + 2] * 1868 + 8192) >> 14);
    auto refl = [](int i, int n) { if (n == 1) return 0; if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; };
    auto at = [&](int r, int c) { return g[(size_t) refl(r, im.rows) * im.cols + refl(c, im.cols)]; };
    auto sat16 = [](float v) { long q = lrintf(v); return (int16_t) std::min(32767l, std::max(-32768l, q)); };
    im.gx.resize(N); im.gy.resize(N);
    for (int r = 0; r < im.rows; ++r) for (int c = 0; c < im.cols; ++c) {
        int sx = 0, sy = 0; const int w3[3] = {1, 2, 1};
        for (int k = -1; k <= 1; ++k) { sx += w3[k + 1] * (at(r + k, c + 1) - at(r + k, c - 1)); sy += w3[k + 1] * (at(r + 1, c + k) - at(r - 1, c + k)); }
        im.gx[(size_t) r * im.cols + c] = sat16((float) sx * height + grad_delta); im.gy[(size_t) r * im.cols + c] = sat16((float) sy * height + grad_delta);
    }
    return im;
}

// checkerboard_color / checkerboard_float / constant_float (src/checkerboard.cpp:65-79, src/consttexture.cpp:38-41)
inline Image load_procedural(const XmlNode &n, int kind, bool constant) {
    Props pl; parse_props(n, pl);
    Image im; im.kind = kind; im.bilinear = false; im.delta = pl.getv2("delta", mk2(0.f, 0.f)); im.scale = pl.getv2("scale", mk2(1.f, 1.f));
    if (kind == IMG_CHECKER_COLOR) {
        V3 a = pl.getv("value1", mk3(0.f)), b = pl.getv("value2", mk3(1.f));
        const float v[6] = {a.x, a.y, a.z, b.x, b.y, b.z}; for (int k = 0; k < 6; ++k) im.values[k] = v[k];
        im.cols = 6; im.rows = 1;
    } else {
        im.values[0] = constant ? pl.getf("value", 0.f) : pl.getf("value1", 0.f); im.values[1] = constant ? im.values[0] : pl.getf("value2", 1.f);
        im.cols = 2; im.rows = 1;
    }
    return im;
}
inline Bsdf parse_bsdf(const XmlNode &n, const std::string &dir, std::vector<Image> &images) {
    Props pl; parse_props(n, pl);
    Bsdf b; std::string type = n.attr("type");
    auto albedo = [&]() {                  // colour property or <texture name="albedo"> child (src/diffuse.cpp:31-63)
        bool has = pl.hasv("albedo"); b.albedo = pl.getv("albedo", mk3(0.5f));
        for (auto &ch : n.children) if (ch->name == "texture") {
            if (std::string(ch->attr("name")) != "albedo") throw std::runtime_error("The name of this texture does not match any field!");
            if (has) throw std::runtime_error("There is already an albedo defined!");
            has = true; std::string tt = ch->attr("type");
            if (tt == "constant_color") { Props tp; parse_props(*ch, tp); b.albedo = tp.getv("value", mk3(0.f)); }
            else if (tt == "image_color") { b.albedo_tex = (int) images.size(); images.push_back(load_image(*ch, dir, IMG_COLOR)); }
            else if (tt == "checkerboard_color") { b.albedo_tex = (int) images.size(); images.push_back(load_procedural(*ch, IMG_CHECKER_COLOR, false)); }
            else throw std::runtime_error("oracle: unsupported texture type \"" + tt + "\"");
        }
    };
    if (type == "diffuse") { b.type = BSDF_DIFFUSE; albedo(); }
    else if (type == "subsurface") {       // src/subsurface.cpp:32-50
        b.type = BSDF_SUBSURFACE; albedo(); b.intIOR = pl.getf("intIOR", 1.3046f); b.extIOR = pl.getf("extIOR", 1.000277f); b.dmfp = pl.getv("dmfp", mk3(0.1f));
    }
    else if (type == "microfacet") {
        b.type = BSDF_MICROFACET; b.alpha = pl.getf("alpha", 0.1f); b.intIOR = pl.getf("intIOR", 1.5046f);
        b.extIOR = pl.getf("extIOR", 1.000277f); b.kd = pl.getv("kd", mk3(0.5f)); b.ks = 1 - maxcoeff(b.kd);
    } else if (type == "dielectric") { b.type = BSDF_DIELECTRIC; b.intIOR = pl.getf("intIOR", 1.5046f); b.extIOR = pl.getf("extIOR", 1.000277f); }
    else if (type == "mirror") b.type = BSDF_MIRROR;
    else throw std::runtime_error("oracle: unsupported bsdf type \"" + type + "\"");
    return b;
}
inline Scene parse_scene(const XmlNode &node, const std::string &dir) {
    Scene sc; bool have_camera = false;
    for (auto &chp : node.children) {
        const XmlNode &ch = *chp; const std::string &t = ch.name;
        if (t == "integrator") sc.integrator = ch.attr("type");
        else if (t == "sampler") { Props pl; parse_props(ch, pl); sc.sample_count = pl.geti("sampleCount", 1); }
        else if (t == "camera") {
            Props pl; parse_props(ch, pl); Camera &c = sc.camera;
            c.width = pl.geti("width", 1280); c.height = pl.geti("height", 720);
            c.cameraToWorld = pl.gett("toWorld"); c.fov = pl.getf("fov", 30.0f);
            c.nearClip = pl.getf("nearClip", 1e-4f); c.farClip = pl.getf("farClip", 1e4f);
            for (auto &rf : ch.children) if (rf->name == "rfilter") {
                Props rp; parse_props(*rf, rp); std::string ft = rf->attr("type");
                if (ft == "gaussian") { c.rfilter_kind = 0; c.rfilter_radius = rp.getf("radius", 2.0f); c.rfilter_p0 = rp.getf("stddev", 0.5f); }
                else if (ft == "mitchell") { c.rfilter_kind = 1; c.rfilter_radius = rp.getf("radius", 2.0f); c.rfilter_p0 = rp.getf("B", 1.0f / 3.0f); c.rfilter_p1 = rp.getf("C", 1.0f / 3.0f); }
                else if (ft == "tent") { c.rfilter_kind = 2; c.rfilter_radius = 1.0f; } else if (ft == "box") { c.rfilter_kind = 3; c.rfilter_radius = 0.5f; }
            }
            c.activate(); have_camera = true;
        } else if (t == "mesh") {
            Props pl; parse_props(ch, pl); Shape sh; std::string type = ch.attr("type");
            if (type == "obj") { sh.type = SHAPE_MESH; load_obj(sh, dir + "/" + pl.gets("filename"), pl.gett("toWorld")); }
            else if (type == "sphere") {
                sh.type = SHAPE_SPHERE; sh.center = pl.getv("center", mk3(0.f)); sh.radius = pl.getf("radius", 1.f);
                sh.bbox.expand(sh.center - mk3(sh.radius)); sh.bbox.expand(sh.center + mk3(sh.radius));
            } else throw std::runtime_error("oracle: unsupported mesh type \"" + type + "\"");
            bool is_emitter = false; Emitter em;
            for (auto &sub : ch.children) {
                if (sub->name == "bsdf") { if (sh.has_bsdf) throw std::runtime_error("Shape: tried to register multiple BSDF instances!"); sh.bsdf = parse_bsdf(*sub, dir, sc.images); sh.has_bsdf = true; }
                else if (sub->name == "normalModifier") {
                    if (sh.normal_tex >= 0) throw std::runtime_error("Shape: tried to register multiple Normal modifier instances!");
                    std::string mt = sub->attr("type");
                    if (mt != "bump_map" && mt != "normal_map") throw std::runtime_error("oracle: unsupported normalModifier type \"" + mt + "\"");
                    Props mp; parse_props(*sub, mp);
                    if (mt == "bump_map" && !mp.hass("filename")) {
                        // the CPU renderer's bump_map (src/bumpMap.cpp): a Texture<float> child named "bump_map"
                        const XmlNode *tex = nullptr;
                        for (auto &tc : sub->children) if (tc->name == "texture") {
                            if (std::string(tc->attr("name")) != "bump_map") throw std::runtime_error("The name of this texture does not match any field!");
                            if (tex) throw std::runtime_error("There is already an bump_map defined!");
                            tex = tc.get();
                        }
                        if (!tex) throw std::runtime_error("There is no texture with name bump_map given!");
                        const std::string tt = tex->attr("type");
                        sh.normal_tex = (int) sc.images.size();
                        if (tt == "checkerboard_float") sc.images.push_back(load_procedural(*tex, IMG_CHECKER_FLOAT, false));
                        else if (tt == "constant_float") sc.images.push_back(load_procedural(*tex, IMG_CHECKER_FLOAT, true));
                        else if (tt == "image_float") { Image im = load_image(*tex, dir, IMG_IMAGE_FLOAT); im.bilinear = false; sc.images.push_back(im); }
                        else throw std::runtime_error("oracle: unsupported float texture type \"" + tt + "\"");
                    } else {
                    sh.normal_tex = (int) sc.images.size(); sc.images.push_back(load_image(*sub, dir, mt == "bump_map" ? IMG_BUMP : IMG_NORMAL));
                    }
                }
                else if (sub->name == "emitter") {
                    if (is_emitter) throw std::runtime_error("Shape: tried to register multiple Emitter instances!");
                    Props ep; parse_props(*sub, ep);
                    if (std::string(sub->attr("type")) != "area") throw std::runtime_error("oracle: unsupported emitter type");
                    if (!ep.hasv("radiance")) throw std::runtime_error("Property 'radiance' is missing!");
                    em.radiance = ep.getv("radiance", mk3(0.f)); is_emitter = true;
                }
            }
            sh.activate();          // default diffuse(0.5) if no bsdf: Bsdf() default (src/shape.cpp:32-39)
            if (is_emitter) { em.shape = (int) sc.shapes.size(); sh.emitter = (int) sc.emitters.size(); sc.emitters.push_back(em); }
            sc.shapes.push_back(std::move(sh));
        }
    }
    if (sc.integrator.empty()) throw std::runtime_error("No integrator was specified!");
    if (!have_camera) throw std::runtime_error("No camera was specified!");
    return sc;
}

inline std::string dir_of(const std::string &path) {
    size_t p = path.find_last_of('/'); return p == std::string::npos ? std::string(".") : path.substr(0, p);
}
inline Scene load_scene_xml(const std::string &path) {
    auto root = xml_load_file(path);
    if (root->name != "scene") throw std::runtime_error("root element is not <scene>");
    return parse_scene(*root, dir_of(path));
}
inline TestSuite load_test_xml(const std::string &path) {
    auto root = xml_load_file(path);
    if (root->name != "test" || std::string(root->attr("type")) != "ttest") throw std::runtime_error("root element is not <test type=ttest>");
    TestSuite ts; Props pl; parse_props(*root, pl);
    ts.significance = pl.getf("significanceLevel", 0.01f);
    ts.sample_count = pl.geti("sampleCount", 100000);
    for (auto &r : tokenize(pl.gets("references", ""), ", \t\n\r")) ts.references.push_back(to_float(r));
    for (auto &ch : root->children) if (ch->name == "scene") ts.scenes.push_back(parse_scene(*ch, dir_of(path)));
    return ts;
}

} // namespace orc
