// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of the per-ray hot path: BVH traversal, ray/triangle and ray/sphere tests, hit-record
// reconstruction, BSDFs, area emitter, warps, PathMisIntegrator::Li, film accumulation, render driver, t-test.
//
// Reference: src/bvh.cpp:404-463, src/mesh.cpp:40-168, src/sphere.cpp:42-152, src/arealight.cpp:41-116,
//            src/diffuse.cpp:75-126, src/microfacet.cpp:55-161, src/dielectric.cpp:48-103, src/mirror.cpp:41-58,
//            src/warp.cpp:59-63,221-244,263-267, src/pathMisIntegrator.cpp:18-168, src/render.cpp:72-198,
//            src/block.cpp:38-195, src/independent.cpp:48-67, src/ttest.cpp:148-195, include/nori/dpdf.h:119-157,
//            src/subsurface.cpp:94-281 (+ the probe loop src/pathMisIntegrator.cpp:86-100),
//            cuda/nori/imagetexture.cpp.cu:12-26, bumpMap.cpp.cu:10-27, normalMap.cpp.cu:10-24, mesh.cpp.cu:181-185,
//            sphere.cpp.cu:145-149, src/imagetexture.cpp:49-77 (images and normal modifiers: parity unpinned).
#pragma once
#include "oscene.hpp"
#include "obvh.hpp"
#include <thread>
#include <atomic>
#include <mutex>

namespace orc {

// ---- unqualified <math.h> calls ---------------------------------------------------------------------------
// The reference calls cos/sin/log/acos/atan2/sqrt UNQUALIFIED on float arguments (src/warp.cpp:28-30,61-62,
// 239-242, src/sphere.cpp:59-60,137-138).  Which overload that picks depends on the C++ library: with GCC >= 6
// <math.h> (pulled in by ImathPlatform.h, include/nori/common.h:35) imports the std:: float overloads into the
// global namespace, so the calls stay in float; on older toolchains they promote to double.  mreal is the type
// those expressions are evaluated in; ORC_UNQUALIFIED_MATH_DOUBLE selects the old behaviour.  The choice made
// here is pinned against the compiled reference (oracle/_ref), see oracle/README.md.
#ifndef ORC_UNQUALIFIED_MATH_DOUBLE
#define ORC_UNQUALIFIED_MATH_DOUBLE 0
#endif
#if ORC_UNQUALIFIED_MATH_DOUBLE
typedef double mreal;
#else
typedef float mreal;
#endif
inline mreal u_cos(float x) { return std::cos((mreal) x); }
inline mreal u_sin(float x) { return std::sin((mreal) x); }
inline mreal u_log(float x) { return std::log((mreal) x); }
inline mreal u_sqrt(float x) { return std::sqrt((mreal) x); }
inline mreal u_atan2(float y, float x) { return std::atan2((mreal) y, (mreal) x); }

struct ScenePrims {
    const Scene &sc; const std::vector<uint32_t> &prim_shape;
    uint32_t count() const { return sc.prim_total(); }
    V3 centroid(uint32_t i) const { uint32_t s = prim_shape[i]; return sc.shapes[s].prim_centroid(i - sc.shape_offset[s]); }
    BBox bbox(uint32_t i) const { uint32_t s = prim_shape[i]; return sc.shapes[s].prim_bbox(i - sc.shape_offset[s]); }
};

struct Intersection {
    V3 p; float t; V2 uv; Frame sh, geo; int shape; uint32_t prim;    // prim = global primitive id (m_indices value)
};

struct TraceCounters { uint64_t nodes = 0, tris = 0; };

// How Sampler::next2D()'s two draws map to (x,y), and which of sample()'s two next2D() arguments is drawn first,
// is decided by the C++ compiler's argument evaluation order in the reference build (src/independent.cpp:62-67,
// src/pathMisIntegrator.cpp:103).  g++ (x86-64) evaluates right-to-left; see oracle/README.md and the _ref pin.
struct Sampler {
    Pcg32 rng;
    bool swap2d = true;          // true: Point2f(y drawn first)  [g++ order]
    float next1D() { return rng.next_float(); }
    V2 next2D() {
        float a = rng.next_float(), b = rng.next_float();
        return swap2d ? mk2(b, a) : mk2(a, b);
    }
};

// ---- images ---------------------------------------------------------------------------------------------
// One lookup of a 4-channel texel grid; cu runs along the columns, rv along the rows, both in [0,1).  Bilinear restates
// CUDA's tex2D for a normalised-coordinate, wrap-addressed, linearly filtered texture object (texel centres at
// i + 0.5, blend weights with 8 fractional bits); nearest is the CPU renderer's truncating lookup.
template <typename Texel>
inline void image_lookup(const Image &im, float cu, float rv, Texel texel, float out[4]) {
    const int W = im.cols, H = im.rows;
    if (!im.bilinear) {
        int col = std::min((int) (cu * (float) W), W - 1), row = std::min((int) (rv * (float) H), H - 1);
        texel(row, col, out); return;
    }
    float xb = cu * (float) W - 0.5f, yb = rv * (float) H - 0.5f;
    float fi = std::floor(xb), fj = std::floor(yb);
    float a = std::floor((xb - fi) * 256.0f + 0.5f) * (1.0f / 256.0f), b = std::floor((yb - fj) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i0 = (int) fi, j0 = (int) fj; if (i0 < 0) i0 += W; if (j0 < 0) j0 += H;
    int i1 = i0 + 1 == W ? 0 : i0 + 1, j1 = j0 + 1 == H ? 0 : j0 + 1;
    float t00[4], t10[4], t01[4], t11[4];
    texel(j0, i0, t00); texel(j0, i1, t10); texel(j1, i0, t01); texel(j1, i1, t11);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    for (int c = 0; c < 4; ++c) out[c] = ((w00 * t00[c] + w10 * t10[c]) + w01 * t01[c]) + w11 * t11[c];
}
inline V2 image_coords(const Image &im, V2 uv) {          // imagetexture.cpp.cu:15-20
    float x = (uv.x / im.scale.x) - im.delta.x, y = (uv.y / im.scale.y) - im.delta.y;
    return mk2(x - std::floor(x), y - std::floor(y));
}
inline void image_rgba(const Image &im, float cu, float rv, float out[4]) {
    image_lookup(im, cu, rv, [&](int r, int c, float *o) { const uint8_t *p = &im.rgba[((size_t) r * im.cols + c) * 4]; for (int k = 0; k < 4; ++k) o[k] = (float) p[k] / 255.0f; }, out);
}
inline float srgb_to_linear(float v) { return v <= 0.04045f ? v * (1.0f / 12.92f) : std::pow((v + 0.055f) * (1.0f / 1.055f), 2.4f); }   // src/common.cpp:208-222
// Checkerboard<T>::eval (src/checkerboard.cpp:32-55): true = value1
inline bool checker_first(const Image &im, V2 uv) {
    float x = (uv.x / im.scale.x) - im.delta.x, y = (uv.y / im.scale.y) - im.delta.y;
    float modx = std::fmod(x, 2.0f), mody = std::fmod(y, 2.0f);
    if (x < 0) modx = std::fmod(std::fabs(x) + 1, 2.0f);
    if (y < 0) mody = std::fmod(std::fabs(y) + 1, 2.0f);
    return (modx <= 1 && mody <= 1) || (modx >= 1 && mody >= 1);
}
inline V3 image_color(const Image &im, V2 uv) {           // ImageTexture<Color3f>::eval / Checkerboard<Color3f>::eval
    if (im.kind == IMG_CHECKER_COLOR) { const float *v = im.values + (checker_first(im, uv) ? 0 : 3); return mk3(v[0], v[1], v[2]); }
    V2 c = image_coords(im, uv); float t[4]; image_rgba(im, c.y, c.x, t);
    if (!im.bilinear) return mk3(t[0], t[1], t[2]);       // CPU variant: no colour-space conversion (src/imagetexture.cpp:49-77)
    return mk3(srgb_to_linear(t[0]), srgb_to_linear(t[1]), srgb_to_linear(t[2]));
}
inline V3 image_normal(const Image &im, V2 uv, const Frame &fr) {      // BumpMap::eval / NormalMap::eval
    // the CPU renderer's bump_map over a Texture<float> (src/bumpMap.cpp:44-50): normal = value, i.e. (v, v, v), as written
    if (im.kind == IMG_CHECKER_FLOAT) return mk3(im.values[checker_first(im, uv) ? 0 : 1]);
    if (im.kind == IMG_IMAGE_FLOAT) {                                  // ImageTexture<float>::eval, src/imagetexture.cpp:26-46
        V2 q = image_coords(im, uv);
        const int row = (int) (q.x * (float) im.rows), col = (int) (q.y * (float) im.cols);
        return mk3((float) im.rgba[((size_t) row * im.cols + col) * 4] / 255.0f);
    }
    V2 c = image_coords(im, uv); float t[4];
    if (im.kind == IMG_BUMP) {
        image_lookup(im, c.x, c.y, [&](int r, int cc, float *o) {
            float x = (float) im.gx[(size_t) r * im.cols + cc] / 32767.0f, y = (float) im.gy[(size_t) r * im.cols + cc] / 32767.0f;
            o[0] = x < -1.0f ? -1.0f : x; o[1] = y < -1.0f ? -1.0f : y; o[2] = o[3] = 0.f; }, t);
        return normalized((fr.n + t[0] * cross(fr.n, fr.s)) - t[1] * cross(fr.n, fr.t));
    }
    image_rgba(im, c.y, c.x, t);
    return fr.toWorld(normalized(mk3((2.0f * t[0]) - 1.0f, (2.0f * t[1]) - 1.0f, t[2])));
}

struct Accel {
    const Scene *scene = nullptr;
    std::vector<uint32_t> prim_shape;      // findShape() (include/nori/bvh.h:105-109) tabulated
    Bvh bvh;

    void build(Scene &sc) {
        scene = &sc;
        sc.shape_offset.assign(1, 0u); sc.bbox.reset();
        for (auto &sh : sc.shapes) { sc.shape_offset.push_back(sc.shape_offset.back() + sh.prim_count()); sc.bbox.expand(sh.bbox); }
        prim_shape.resize(sc.prim_total());
        for (size_t s = 0; s < sc.shapes.size(); ++s)
            for (uint32_t i = sc.shape_offset[s]; i < sc.shape_offset[s + 1]; ++i) prim_shape[i] = (uint32_t) s;
        ScenePrims sp{sc, prim_shape};
        BvhBuilder<ScenePrims> b(sp, bvh);
        b.build(sc.bbox);
    }

    // src/mesh.cpp:84-121
    static bool tri_intersect(const Shape &m, uint32_t index, const Ray &ray, float &u, float &v, float &t) {
        uint32_t i0 = m.F[3 * index], i1 = m.F[3 * index + 1], i2 = m.F[3 * index + 2];
        const V3 p0 = m.V[i0], p1 = m.V[i1], p2 = m.V[i2];
        V3 edge1 = p1 - p0, edge2 = p2 - p0;
        V3 pvec = cross(ray.d, edge2);
        float det = dot(edge1, pvec);
        if (det > -1e-8f && det < 1e-8f) return false;
        float inv_det = 1.0f / det;
        V3 tvec = ray.o - p0;
        u = dot(tvec, pvec) * inv_det;
        if (u < 0.0 || u > 1.0) return false;
        V3 qvec = cross(tvec, edge1);
        v = dot(ray.d, qvec) * inv_det;
        if (v < 0.0 || u + v > 1.0) return false;
        t = dot(edge2, qvec) * inv_det;
        return t >= ray.mint && t <= ray.maxt;
    }
    // src/sphere.cpp:42-113
    static bool sphere_intersect(const Shape &s, const Ray &ray, float &t) {
        V3 c = s.center; float r = s.radius;
        float B = 2 * dot(ray.o - c, ray.d);
        float C = ((-r * r + dot(c, c)) + dot(ray.o, ray.o)) - 2 * dot(c, ray.o);
        float discr = B * B - 4 * C;
        if (discr > 0) {
            float locT1 = (float) ((-B + u_sqrt(discr)) / (2.0));
            float locT2 = (float) ((-B - u_sqrt(discr)) / (2.0));
            float locT;
            bool t1Valid = locT1 >= ray.mint && locT1 <= ray.maxt;
            bool t2Valid = locT2 >= ray.mint && locT2 <= ray.maxt;
            if (t1Valid && t2Valid) locT = std::min(locT1, locT2);
            else if (t1Valid) locT = locT1;
            else if (t2Valid) locT = locT2;
            else return false;
            t = locT; return true;
        } else if (discr == 0) {
            float locT = (float) (-B / (2.0));
            if (locT >= ray.mint && locT < ray.maxt) { t = locT; return true; }
        }
        return false;
    }

    // src/mesh.cpp:123-168, src/sphere.cpp:115-141
    void set_hit_information(uint32_t f, const Ray &ray, Intersection &its) const {
        set_hit_information_base(f, ray, its);
        const Shape &m = scene->shapes[its.shape];
        if (m.normal_tex >= 0) {
            const Image &im = scene->images[m.normal_tex];
            // the float kinds are the CPU renderer's modifier: meshes with vertex normals only (src/mesh.cpp:151-164)
            const bool cpu_kind = im.kind == IMG_CHECKER_FLOAT || im.kind == IMG_IMAGE_FLOAT;
            if (!cpu_kind || (m.type == SHAPE_MESH && !m.N.empty())) its.sh = Frame(image_normal(im, its.uv, its.sh));   // cuda/nori/mesh.cpp.cu:181-185
        }
    }
    void set_hit_information_base(uint32_t f, const Ray &ray, Intersection &its) const {
        const Shape &m = scene->shapes[its.shape];
        if (m.type == SHAPE_SPHERE) {
            its.p = ray.o + its.t * ray.d;
            its.sh = Frame(normalized(its.p - m.center));
            its.geo = its.sh;
            V3 pl = its.p - m.center;
            float frac = pl.z / m.radius;
            if (frac < -1) frac = -1; else if (frac > 1) frac = 1;
            float theta = (float) (std::acos((mreal) frac) / (kPi));
            float phi = (float) ((u_atan2(pl.y, pl.x) + kPi) / (2 * kPi));
            its.uv = mk2(phi, theta);
            return;
        }
        uint32_t index = f - scene->shape_offset[its.shape];
        float b0 = 1 - (its.uv.x + its.uv.y), b1 = its.uv.x, b2 = its.uv.y;
        uint32_t i0 = m.F[3 * index], i1 = m.F[3 * index + 1], i2 = m.F[3 * index + 2];
        V3 p0 = m.V[i0], p1 = m.V[i1], p2 = m.V[i2];
        its.p = (b0 * p0 + b1 * p1) + b2 * p2;
        if (!m.UV.empty()) {
            V2 a = m.UV[i0], b = m.UV[i1], c = m.UV[i2];
            its.uv = mk2((b0 * a.x + b1 * b.x) + b2 * c.x, (b0 * a.y + b1 * b.y) + b2 * c.y);
        }
        its.geo = Frame(normalized(cross(p1 - p0, p2 - p0)));
        if (!m.N.empty()) {
            V3 n = normalized_seq((b0 * m.N[i0] + b1 * m.N[i1]) + b2 * m.N[i2]);
            its.sh = Frame(n);
        } else its.sh = its.geo;
    }

    // src/bvh.cpp:404-463.  mask_shape < 0 means no mask.
    bool ray_intersect(const Ray &_ray, Intersection &its, bool shadowRay, int mask_shape = -1, TraceCounters *ctr = nullptr) const {
        uint32_t node_idx = 0, stack_idx = 0, stack[64];
        its.t = kInf;
        Ray ray(_ray);
        if (ray.mint == kEpsilon)
            ray.mint = std::max(ray.mint, ray.mint * std::max(std::abs(ray.o.x), std::max(std::abs(ray.o.y), std::abs(ray.o.z))));
        if (bvh.nodes.empty() || ray.maxt < ray.mint) return false;
        bool found = false; uint32_t f = 0;
        while (true) {
            const BvhNode &node = bvh.nodes[node_idx];
            if (ctr) ctr->nodes++;
            if (!node.bbox().ray_intersect(ray)) {
                if (stack_idx == 0) break;
                node_idx = stack[--stack_idx];
                continue;
            }
            if (!node.is_leaf()) {
                stack[stack_idx++] = node.right();
                node_idx++;
            } else {
                for (uint32_t i = node.start(), end = node.start() + node.size(); i < end; ++i) {
                    uint32_t idx = bvh.indices[i];
                    int s = (int) prim_shape[idx];
                    if (mask_shape >= 0 && mask_shape != s) continue;
                    const Shape &shape = scene->shapes[s];
                    float u = 0, v = 0, t = 0; bool hit;
                    if (ctr) ctr->tris++;
                    if (shape.type == SHAPE_MESH) hit = tri_intersect(shape, idx - scene->shape_offset[s], ray, u, v, t);
                    else hit = sphere_intersect(shape, ray, t);
                    if (hit) {
                        if (shadowRay) return true;
                        found = true;
                        ray.maxt = its.t = t;
                        its.uv = mk2(u, v);
                        its.shape = s;
                        f = idx;
                    }
                }
                if (stack_idx == 0) break;
                node_idx = stack[--stack_idx];
                continue;
            }
        }
        if (found) { its.prim = f; set_hit_information(f, ray, its); }
        return found;
    }
};

// ---- warps (src/warp.cpp) -------------------------------------------------------------------------------
inline V2 square_to_uniform_disk(V2 s) {                     // src/warp.cpp:59-63
    float r = std::sqrt(s.x);
    float theta = 2 * kPi * s.y;
    return mk2((float) (r * u_cos(theta)), (float) (r * u_sin(theta)));
}
inline V3 square_to_cosine_hemisphere(V2 s) {                // src/warp.cpp:221-224
    V2 d = square_to_uniform_disk(s);
    return mk3(d.x, d.y, std::sqrt(1.0f - d.x * d.x - d.y * d.y));
}
inline V3 square_to_beckmann(V2 s, float alpha) {            // src/warp.cpp:237-244
    mreal inner = 1.0f / (1.0f - alpha * alpha * u_log(1.0f - s.x));
    float theta = (float) std::acos(std::sqrt(inner));
    float phi = s.y * 2.0f * kPi;
    return mk3((float) (u_sin(theta) * u_cos(phi)), (float) (u_sin(theta) * u_sin(phi)), (float) u_cos(theta));
}
inline V3 square_to_uniform_triangle(V2 s) {
    float su1 = sqrtf(s.x);
    float u = 1.f - su1, v = s.y * su1;
    return mk3(u, v, 1.f - u - v);
}

// ---- BSDFs ----------------------------------------------------------------------------------------------
struct BsdfRec { V3 wi, wo; V2 uv; bool isDelta = false; bool wo_valid = false; V3 albedo; };   // albedo: Texture::eval(uv), filled by the integrator

inline float ss_fresnel_transmission(const Bsdf &b, const BsdfRec &r) {     // src/subsurface.cpp:173-178
    float Ft_i = 1.0f - fresnel(r.wi.z, b.extIOR, b.intIOR);
    float Ft_o = 1.0f - fresnel(r.wo.z, b.extIOR, b.intIOR);
    return Ft_i * Ft_o;
}

inline float beckmann_D(const Bsdf &b, V3 m) {             // src/microfacet.cpp:55-61
    float temp = tan_theta(m) / b.alpha, ct = m.z, ct2 = ct * ct;
    return std::exp(-temp * temp) / (kPi * b.alpha * b.alpha * ct2 * ct2);
}
inline float smith_g1(const Bsdf &b, V3 v, V3 m) {         // src/microfacet.cpp:64-84
    float tanTheta = tan_theta(v);
    if (tanTheta == 0.0f) return 1.0f;
    if (dot(m, v) * v.z <= 0) return 0.0f;
    float a = 1.0f / (b.alpha * tanTheta);
    if (a >= 1.6f) return 1.0f;
    float a2 = a * a;
    return (3.535f * a + 2.181f * a2) / (1.0f + 2.276f * a + 2.577f * a2);
}
inline V3 bsdf_eval(const Bsdf &b, const BsdfRec &r) {
    switch (b.type) {
    case BSDF_DIFFUSE:                                       // src/diffuse.cpp:75-86
        if (r.wi.z <= 0 || r.wo.z <= 0) return mk3(0.f);
        return r.albedo * kInvPi;
    case BSDF_SUBSURFACE:                                    // src/subsurface.cpp:99-104
        return (ss_fresnel_transmission(b, r) * r.albedo) * kInvPi;
    case BSDF_MICROFACET: {                                  // src/microfacet.cpp:87-100
        V3 wh = normalized(r.wi + r.wo);
        float G = smith_g1(b, r.wi, wh) * smith_g1(b, r.wo, wh);
        float ci = r.wi.z, co = r.wo.z;
        if (co > 0 && ci > 0) {
            float spec = (b.ks * beckmann_D(b, wh) * fresnel(dot(wh, r.wi), b.extIOR, b.intIOR) * G) / (4 * ci * co);
            return b.kd / kPi + mk3(spec);
        }
        return mk3(0.f);
    }
    default: return mk3(0.f);
    }
}
inline float bsdf_pdf(const Bsdf &b, const BsdfRec &r) {
    switch (b.type) {
    case BSDF_DIFFUSE:                                       // src/diffuse.cpp:89-105
    case BSDF_SUBSURFACE:                                    // src/subsurface.cpp:181-197
        if (r.wi.z <= 0 || r.wo.z <= 0) return 0.0f;
        return kInvPi * r.wo.z;
    case BSDF_MICROFACET: {                                  // src/microfacet.cpp:103-117
        V3 wh = normalized(r.wi + r.wo);
        float ch = wh.z, co = r.wo.z, ci = r.wi.z;
        float Jh = 1.0f / (4 * dot(wh, r.wo));
        if (co > 0 && ci > 0) return b.ks * beckmann_D(b, wh) * ch * Jh + (1 - b.ks) * co / kPi;
        return 0;
    }
    default: return 0.0f;
    }
}
// sample(): returns the importance weight; sets r.wo / r.isDelta.  r.wo_valid=false marks the reference's
// early-return branches that leave bRec.wo uninitialised (src/diffuse.cpp:111-112, src/mirror.cpp:42-43).
inline V3 bsdf_sample(const Bsdf &b, BsdfRec &r, V2 sample, V2 sample2 = mk2(0.f, 0.f)) {
    r.wo_valid = true;
    switch (b.type) {
    case BSDF_DIFFUSE:                                       // src/diffuse.cpp:109-126
        if (r.wi.z <= 0) { r.wo_valid = false; return mk3(0.f); }
        r.wo = square_to_cosine_hemisphere(sample);
        return r.albedo;
    case BSDF_SUBSURFACE:                                    // src/subsurface.cpp:253-281 (uses the SECOND sample)
        r.wo = square_to_cosine_hemisphere(sample2);
        if (r.wi.z <= 0 || r.wo.z <= 0) return mk3(0.f);
        return ss_fresnel_transmission(b, r) * r.albedo;
    case BSDF_MICROFACET: {                                  // src/microfacet.cpp:121-161
        if (sample.x < b.ks) {
            sample.x /= b.ks;
            V3 wh = square_to_beckmann(sample, b.alpha);
            wh = wh / norm(wh);
            V3 wo = -r.wi + (2 * dot(wh, r.wi)) * wh;
            r.wo = wo;
            if (wo.z <= 0) return mk3(0.f);
        } else {
            sample.x = (sample.x - b.ks) / (1.0f - b.ks);
            r.wo = square_to_cosine_hemisphere(sample);
        }
        float co = r.wo.z;
        if (co > 0) {
            float pdf_val = bsdf_pdf(b, r);
            if (std::fabs((double) pdf_val) > 1e-6) return (bsdf_eval(b, r) * co) / pdf_val;
        }
        return mk3(0.f);
    }
    case BSDF_DIELECTRIC: {                                  // src/dielectric.cpp:48-103
        float ci = r.wi.z;
        float extIOR = b.extIOR, intIOR = b.intIOR;
        if (ci < 0) { extIOR = b.intIOR; intIOR = b.extIOR; }
        float F = fresnel(std::fabs(ci), extIOR, intIOR);
        r.isDelta = true;
        if (sample.x <= F) { r.wo = mk3(-r.wi.x, -r.wi.y, r.wi.z); return mk3(1.0f); }
        V3 n = mk3(0, 0, 1.0f); if (ci < 0) n = mk3(0, 0, -1.0f);
        float win = dot(r.wi, n);
        float IORrel = extIOR / intIOR;
        // -IORrel*(wi - win*n) - n*sqrt(1.0 - IORrel*IORrel*(1.0 - win*win))   [sqrt argument in double]
        float root = (float) std::sqrt(1.0 - (double) (IORrel * IORrel) * (1.0 - (double) (win * win)));
        V3 wo = (-IORrel) * (r.wi - win * n) - n * root;
        r.wo = wo / norm(wo);
        float inv = 1.0f / IORrel;
        return mk3(inv * inv);
    }
    case BSDF_MIRROR:                                        // src/mirror.cpp:41-58
        if (r.wi.z <= 0) { r.wo_valid = false; return mk3(0.f); }
        r.wo = mk3(-r.wi.x, -r.wi.y, r.wi.z); r.isDelta = true;
        return mk3(1.0f);
    }
    return mk3(0.f);
}

// ---- area emitter (src/arealight.cpp) -------------------------------------------------------------------
struct EmitterRec { V3 ref, p, n, wi; float pdf; Ray shadowRay; };

inline size_t dpdf_sample(const std::vector<float> &cdf, float v) {                  // dpdf.h:119-124
    auto entry = std::lower_bound(cdf.begin(), cdf.end(), v);
    size_t index = (size_t) std::max((ptrdiff_t) 0, entry - cdf.begin() - 1);
    return std::min(index, cdf.size() - 2);
}
inline void shape_sample_surface(const Shape &m, V2 sample, V3 &p, V3 &n, float &pdf) {
    if (m.type == SHAPE_SPHERE) {                                                    // src/sphere.cpp:143-149, src/warp.cpp:26-31,198-203
        float phi = 2 * kPi * sample.x; float zc = sample.y;
        float z = 2.0f * zc - 1.0f; float r = std::sqrt(1.0f - (z * z));
        V3 q = mk3((float) u_cos(phi) * r, (float) u_sin(phi) * r, z);
        p = m.center + m.radius * q; n = q;
        pdf = (float) (1.0 / (4.0 * kPi * m.radius * m.radius));
        return;
    }
    V2 s = sample;                                                                   // src/mesh.cpp:40-58
    size_t idT = dpdf_sample(m.cdf, s.x);
    s.x = (s.x - m.cdf[idT]) / (m.cdf[idT + 1] - m.cdf[idT]);                        // dpdf.h:152-157
    V3 bc = square_to_uniform_triangle(s);
    uint32_t i0 = m.F[3 * idT], i1 = m.F[3 * idT + 1], i2 = m.F[3 * idT + 2];
    p = (bc.x * m.V[i0] + bc.y * m.V[i1]) + bc.z * m.V[i2];
    if (!m.N.empty()) n = normalized_seq((bc.x * m.N[i0] + bc.y * m.N[i1]) + bc.z * m.N[i2]);
    else n = normalized(cross(m.V[i1] - m.V[i0], m.V[i2] - m.V[i0]));
    pdf = m.cdf_norm;
}
inline float shape_pdf_surface(const Shape &m) {
    if (m.type == SHAPE_SPHERE) { float a = 1.f / m.radius; return (a * a) * (1.0f / (4.0f * kPi)); }   // src/sphere.cpp:150-152
    return m.cdf_norm;
}
inline float emitter_pdf(const Shape &m, const EmitterRec &l) {                      // src/arealight.cpp:100-116
    float dist2 = sqnorm(l.ref - l.p);
    float costheta = -dot(normalized(l.wi), normalized(l.n));
    if (costheta <= 0) return 0.0f;
    return shape_pdf_surface(m) * dist2 / std::fabs(costheta);
}
inline V3 emitter_eval(const Emitter &e, const EmitterRec &l) {                      // src/arealight.cpp:41-54
    return dot(l.n, l.wi) < 0 ? e.radiance : mk3(0.f);
}
inline V3 emitter_sample(const Emitter &e, const Shape &m, EmitterRec &l, V2 sample) {   // src/arealight.cpp:56-98
    V3 p, n; float spdf;
    shape_sample_surface(m, sample, p, n, spdf);
    V3 dir = p - l.ref;
    float dist = norm(dir);
    l.wi = normalized(dir); l.n = n; l.p = p;
    l.shadowRay.o = p; l.shadowRay.d = -l.wi;
    float delta = 1e-3;
    l.shadowRay.mint = delta; l.shadowRay.maxt = dist - delta; l.shadowRay.update();
    float pdfv = emitter_pdf(m, l);
    l.pdf = pdfv;
    if (std::fabs((double) pdfv) > 1e-6) return e.radiance / pdfv;
    return mk3(0.f);
}

struct PathStats { uint64_t closest = 0, shadow = 0, bounces = 0; TraceCounters tc; };

// ---- PathMisIntegrator::Li (src/pathMisIntegrator.cpp:18-168) -------------------------------------------
// Differences from the reference, none of which changes the returned radiance:
//  * when sample() returns a zero weight the path is ended: the reference multiplies the throughput by zero and
//    keeps bouncing (with an uninitialised wo for the two early-return branches) until roulette/escape, adding
//    exact zeros.  Only the amount of random numbers consumed afterwards differs.
//  * the probe loop of a subsurface hit (:88-90) is unbounded in the reference; here a path is dropped after
//    kProbeTries unsuccessful probes (never reached on closed meshes).
// ---- Subsurface::sampleDx and its helpers (src/subsurface.cpp:94-150,199-249) ---------------------------
#ifdef ORC_SS_EXP_DOUBLE
#define SSEXP(x) std::exp((double) (x))
#else
#define SSEXP(x) std::exp(x)
#endif
inline float color_norm(V3 c) { return std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z); }
inline float ss_cdf(float r, float cs, float cd) { float rsd = -r * cs / cd; return 1.0f - 0.25f * SSEXP(rsd) - 0.75f * SSEXP(rsd / 3.0f); }
inline float ss_cdf_derivative(float r, float cs, float cd) { float sd = cs / cd; float rsd = -r * sd; return sd * 0.25f * (SSEXP(rsd) + SSEXP(rsd / 3.0f)); }
inline V2 ss_sample_disk(V2 sample2, float cs, float cd) {
    float sample = sample2.x;
    float sdInv = cd / cs;
    float r0 = -std::log((1.0f - sample) * 4.0 / 3.0) * 3.0f * sdInv;      // double until the assignment
    float r1;                                                               // (the reference's r1 / err0 / err1 never influence r0)
    for (int i = 0; i < 3; ++i) {
        float f = (ss_cdf(r0, cs, cd) - sample);
        float df = ss_cdf_derivative(r0, cs, cd);
        r1 = r0 - f / df;
        r0 = r1;
    }
    float theta = 2.0f * kPi * sample2.y;                                   // M_PI is a float literal in Nori (include/nori/common.h:56)
    return mk2((float) (r0 * u_cos(theta)), (float) (r0 * u_sin(theta)));
}
// Subsurface::sampleDx: one probe ray (direction picked from the shading frame, origin offset by the sampled disk point)
inline void ss_sample_dx(const Bsdf &bsdf, V3 p, const Frame &sh, Sampler &smp, Ray &probe, V2 &dx) {
    const float cs = color_norm(mk3(1.0f)), cd = color_norm(bsdf.dmfp);
    float thres = smp.next1D();
    float k = 0.5f / 4.0f;
    if (thres < 0.5f) probe.d = -sh.n;
    else if (thres < 0.5f + k) probe.d = sh.t;
    else if (thres < 0.5f + 2 * k) probe.d = -sh.t;
    else if (thres < 0.5f + 3 * k) probe.d = sh.s;
    else probe.d = -sh.s;
    dx = ss_sample_disk(smp.next2D(), cs, cd);
    Frame frame(-probe.d);
    probe.o = p + frame.toWorld(mk3(dx.x, dx.y, 0));
    float eps = 1e-3;
    probe.mint = -eps; probe.maxt = std::sqrt(dx.x * dx.x + dx.y * dx.y) + eps; probe.update();
}
static const int kProbeTries = 4096;

inline V3 path_mis_li(const Accel &acc, Sampler &smp, const Ray &_ray, PathStats *st = nullptr, bool bsdf_second_first = true) {
    const Scene &sc = *acc.scene;
    Ray ray = _ray;
    V3 li = mk3(0.f), t = mk3(1.0f);
    Intersection its;
    bool lastBsdfDelta = true; float last_bsdf_pdf = 1.0f;
    while (true) {
        if (st) st->closest++;
        if (!acc.ray_intersect(ray, its, false, -1, st ? &st->tc : nullptr)) break;
        const Shape &mesh = sc.shapes[its.shape];
        if (mesh.emitter >= 0) {
            const Emitter &light = sc.emitters[mesh.emitter];
            EmitterRec q; q.ref = ray.o; q.p = its.p; q.n = its.sh.n; q.wi = normalized(its.p - ray.o);
            V3 flight = emitter_eval(light, q);
            float light_pdf = emitter_pdf(mesh, q);
            float wmat = last_bsdf_pdf / (light_pdf + last_bsdf_pdf);
            if (lastBsdfDelta) wmat = 1.0f;
            if (wmat > 0.0f) li = li + (wmat * t) * flight;
        }
        float successProb = 0.95;
        if (smp.next1D() > successProb) break;
        t = t / successProb;
        if (st) st->bounces++;

        const Bsdf &bsdf = mesh.bsdf;
        BsdfRec br; br.wi = its.sh.toLocal(-ray.d); br.uv = its.uv;
        br.albedo = bsdf.albedo_tex >= 0 ? image_color(sc.images[bsdf.albedo_tex], its.uv) : bsdf.albedo;
        Intersection its_out = its;                                              // BSDF::sampleDx default (include/nori/bsdf.h:97)
        if (bsdf.type == BSDF_SUBSURFACE) {                                      // :86-95, src/subsurface.cpp:199-249
            bool found = false;
            for (int tries = 0; tries < kProbeTries && !found; ++tries) {
                Ray probe; V2 dx;
                ss_sample_dx(bsdf, its.p, its.sh, smp, probe, dx);
                if (st) st->closest++;
                found = acc.ray_intersect(probe, its_out, false, its.shape, st ? &st->tc : nullptr);
            }
            if (!found) break;
        }
        V2 s1, s2;
        if (bsdf_second_first) { s2 = smp.next2D(); s1 = smp.next2D(); } else { s1 = smp.next2D(); s2 = smp.next2D(); }
        V3 f = bsdf_sample(bsdf, br, s1, s2);
        last_bsdf_pdf = br.wo_valid ? bsdf_pdf(bsdf, br) : 0.0f;
        lastBsdfDelta = br.isDelta;
        V3 next_dir = br.wo_valid ? normalized(its_out.sh.toWorld(br.wo)) : mk3(0.f);

        float pick = smp.next1D();
        if (!br.isDelta) {
            size_t n = sc.emitters.size();                                          // include/nori/scene.h:67-73
            size_t index = std::min((size_t) std::floor(n * pick), n - 1);
            const Emitter &light = sc.emitters[index];
            const Shape &lshape = sc.shapes[light.shape];
            EmitterRec q; q.ref = its_out.p;
            V3 liLoc = emitter_sample(light, lshape, q, smp.next2D());
            V3 light_dir = q.wi;
            Intersection tmp;
            if (st) st->shadow++;
            if (!acc.ray_intersect(q.shadowRay, tmp, true, -1, st ? &st->tc : nullptr)) {
                br.wo = its_out.sh.toLocal(light_dir);
                float bpdf = bsdf_pdf(bsdf, br);
                float lpdf = emitter_pdf(lshape, q);
                float wem = lpdf / (lpdf + bpdf);
                float cosS = dot(normalized(light_dir), normalized(its_out.sh.n));
                if (wem > 0) li = li + ((((t * wem) * liLoc) * bsdf_eval(bsdf, br)) * std::fabs(cosS)) * (float) n;
            }
        }
        if (!br.wo_valid || is_zero(f)) break;
        ray.d = next_dir; ray.o = its_out.p; ray.mint = 1e-3; ray.maxt = kInf; ray.update();
        t = t * f;
    }
    return li;
}

// ---- PathMatsIntegrator::Li (src/pathMatsIntegrator.cpp:19-77): BSDF sampling only --------------------------------
// Same remark as above on zero BSDF weights.  Note what the reference leaves as it is: the next ray keeps the camera
// ray's maxt (only o, d and mint are reassigned, :66-69) and mint = 1e-4 = Epsilon switches the adaptive ray epsilon of
// BVH::rayIntersect on (src/bvh.cpp:411-412); the survival probability is min(0.99, max(t)) (:46).
inline V3 path_mats_li(const Accel &acc, Sampler &smp, const Ray &_ray, PathStats *st = nullptr) {
    const Scene &sc = *acc.scene;
    Ray ray = _ray;
    V3 li = mk3(0.f), t = mk3(1.0f);
    while (true) {
        Intersection its;
        if (st) st->closest++;
        if (!acc.ray_intersect(ray, its, false, -1, st ? &st->tc : nullptr)) break;
        const Shape &mesh = sc.shapes[its.shape];
        if (mesh.emitter >= 0) {
            EmitterRec q; q.ref = ray.o; q.p = its.p; q.n = its.sh.n; q.wi = normalized(its.p - ray.o);
            li = li + t * emitter_eval(sc.emitters[mesh.emitter], q);
        }
        float successProb = std::min(0.99f, maxcoeff(t));
        if (smp.next1D() > successProb) break;
        t = t / successProb;
        if (st) st->bounces++;
        const Bsdf &bsdf = mesh.bsdf;
        BsdfRec br; br.wi = its.sh.toLocal(-ray.d); br.uv = its.uv;
        br.albedo = bsdf.albedo_tex >= 0 ? image_color(sc.images[bsdf.albedo_tex], its.uv) : bsdf.albedo;
        V2 s2 = smp.next2D(), s1 = smp.next2D();              // right-to-left argument evaluation, as in path_mis
        V3 f = bsdf_sample(bsdf, br, s1, s2);
        if (!br.wo_valid || is_zero(f)) break;
        ray.d = normalized(its.sh.toWorld(br.wo)); ray.o = its.p; ray.mint = 1e-4; ray.update();
        t = t * f;
    }
    return li;
}
// ---- DirectMisIntegrator::Li (src/directMisIntegrator.cpp:19-141): one bounce, every emitter sampled once + one BSDF sample
// As written: no absolute value on the cosine of the emitter term (:80-85), the direct hit adds the full emission (:36-42),
// the BSDF-sampled ray keeps the unnormalised direction its.toWorld(wo) (:103-106), and for delta BSDFs pdf() = 0 makes the
// weight of the BSDF-sampled term zero (:127-133).
inline V3 direct_mis_li(const Accel &acc, Sampler &smp, const Ray &ray, PathStats *st = nullptr) {
    const Scene &sc = *acc.scene;
    Intersection its;
    if (st) st->closest++;
    if (!acc.ray_intersect(ray, its, false, -1, st ? &st->tc : nullptr)) return mk3(0.f);
    V3 li = mk3(0.f);
    const Frame shFrame = its.sh; const V2 uv = its.uv;
    const Shape &mesh = sc.shapes[its.shape];
    if (mesh.emitter >= 0) {
        EmitterRec q; q.ref = ray.o; q.p = its.p; q.n = its.sh.n; q.wi = normalized(its.p - ray.o);
        li = li + emitter_eval(sc.emitters[mesh.emitter], q);
    }
    const Bsdf &bsdf = mesh.bsdf;
    const V3 cam_dir = -ray.d;
    const V3 albedo = bsdf.albedo_tex >= 0 ? image_color(sc.images[bsdf.albedo_tex], uv) : bsdf.albedo;
    for (const Emitter &light : sc.emitters) {
        const Shape &lshape = sc.shapes[light.shape];
        EmitterRec q; q.ref = its.p;
        V3 liLoc = emitter_sample(light, lshape, q, smp.next2D());
        V3 light_dir = q.wi;
        Intersection tmp;
        if (st) st->shadow++;
        if (acc.ray_intersect(q.shadowRay, tmp, true, -1, st ? &st->tc : nullptr)) continue;
        BsdfRec br; br.wi = shFrame.toLocal(cam_dir); br.wo = shFrame.toLocal(light_dir); br.uv = uv; br.albedo = albedo;
        V3 f = bsdf_eval(bsdf, br);
        float bpdf = bsdf_pdf(bsdf, br);
        float lpdf = emitter_pdf(lshape, q);
        float wem = lpdf / (lpdf + bpdf);
        float cosS = dot(normalized(light_dir), normalized(shFrame.n));
        if (wem > 0) li = li + ((wem * liLoc) * f) * cosS;
    }
    {
        BsdfRec br; br.wi = shFrame.toLocal(cam_dir); br.uv = uv; br.albedo = albedo;
        V2 s2 = smp.next2D(), s1 = smp.next2D();
        V3 f = bsdf_sample(bsdf, br, s1, s2);
        if (!br.wo_valid) return li;                          // (the reference goes on with an uninitialised direction and a zero weight)
        Ray refl; refl.o = its.p; refl.d = its.sh.toWorld(br.wo); refl.mint = 1e-4; refl.maxt = kInf; refl.update();
        Intersection its2;
        if (st) st->closest++;
        if (acc.ray_intersect(refl, its2, false, -1, st ? &st->tc : nullptr)) {
            const Shape &m2 = sc.shapes[its2.shape];
            if (m2.emitter >= 0) {
                EmitterRec q; q.ref = refl.o; q.p = its2.p; q.n = its2.sh.n; q.wi = normalized(its2.p - refl.o);
                V3 flight = emitter_eval(sc.emitters[m2.emitter], q);
                float bpdf = bsdf_pdf(bsdf, br);
                float lpdf = emitter_pdf(m2, q);
                float wmat = bpdf / (lpdf + bpdf);
                if (wmat > 0) li = li + (wmat * f) * flight;
            }
        }
    }
    return li;
}
// the integrator a scene names
inline V3 scene_li(const Accel &acc, Sampler &smp, const Ray &ray, PathStats *st = nullptr) {
    if (acc.scene->integrator == "path_mats") return path_mats_li(acc, smp, ray, st);
    if (acc.scene->integrator == "direct_mis") return direct_mis_li(acc, smp, ray, st);
    return path_mis_li(acc, smp, ray, st);
}

// ---- film (src/block.cpp:38-140) ------------------------------------------------------------------------
struct Film {
    int w = 0, h = 0, border = 0;
    std::vector<float> px;         // (h+2b) x (w+2b) x 4, row-major
    // the tabulated reconstruction filter of ImageBlock::init (src/block.cpp:54-64); used only by the splat mode
    bool splat = false; float radius = 0.f, lookup = 0.f, table[33];
    void init(int W, int H, float filter_radius) {
        w = W; h = H; border = (int) std::ceil(filter_radius - 0.5f);
        px.assign((size_t) (h + 2 * border) * (w + 2 * border) * 4, 0.f);
    }
    void set_filter(const Camera &cam) {
        radius = cam.rfilter_radius;
        for (int i = 0; i < 32; ++i) { float pos = (radius * i) / 32; table[i] = cam.rfilter_eval(pos); }
        table[32] = 0.0f; lookup = 32 / radius;
    }
    // ImageBlock::put(sample) as UPSTREAM Nori has it: the code behind this fork's early return (src/block.cpp:113-128).
    // Block-local coordinates as in put(); with border = ceil(radius - 0.5) the rectangle never needs clipping for a
    // sample inside its block, so the block's borders carry the spill into the neighbours when the blocks are merged.
    bool put_splat(V2 pos, V3 value, int ox, int oy) {
        if (!color_valid(value)) return false;
        float fx = pos.x - 0.5f - (float) (ox - border), fy = pos.y - 0.5f - (float) (oy - border);
        int x0 = (int) std::ceil(fx - radius), y0 = (int) std::ceil(fy - radius), x1 = (int) std::floor(fx + radius), y1 = (int) std::floor(fy + radius);
        for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
            float wx = table[(int) (std::abs(x - fx) * lookup)], wy = table[(int) (std::abs(y - fy) * lookup)];
            int ix = x + ox, iy = y + oy;
            if (ix < 0 || iy < 0 || ix >= stride() || iy >= h + 2 * border) continue;
            float *p = &px[((size_t) iy * stride() + ix) * 4];
            p[0] += (value.x * wx) * wy; p[1] += (value.y * wx) * wy; p[2] += (value.z * wx) * wy; p[3] += (1.0f * wx) * wy;
        }
        return true;
    }
    int stride() const { return w + 2 * border; }
    // ImageBlock::put(sample) as written: early return after a single-pixel add (src/block.cpp:93-111), done in
    // the coordinates of the 32x32 block at offset (ox, oy) that generated the sample and then shifted into the
    // whole image as ImageBlock::put(ImageBlock&) does (:131-140).  Keeping the block-local arithmetic keeps the
    // float rounding of the reference.
    bool put(V2 pos, V3 value, int ox, int oy) {
        if (splat) return put_splat(pos, value, ox, oy);
        if (!color_valid(value)) return false;
        float fx = pos.x - 0.5f - (float) (ox - border), fy = pos.y - 0.5f - (float) (oy - border);
        int ix = (int) fx + ox, iy = (int) fy + oy;
        float *p = &px[((size_t) iy * stride() + ix) * 4];
        p[0] += value.x; p[1] += value.y; p[2] += value.z; p[3] += 1.0f;
        return true;
    }
    // toBitmap (src/block.cpp:76-82) → RGB, row-major h x w x 3
    std::vector<float> to_bitmap() const {
        std::vector<float> out((size_t) w * h * 3);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            const float *p = &px[((size_t) (y + border) * stride() + (x + border)) * 4];
            float *o = &out[((size_t) y * w + x) * 3];
            if (p[3] != 0) { o[0] = p[0] / p[3]; o[1] = p[1] / p[3]; o[2] = p[2] / p[3]; } else o[0] = o[1] = o[2] = 0.f;
        }
        return out;
    }
};

// spiral block order (src/block.cpp:147-195)
inline std::vector<std::pair<int, int>> spiral_blocks(int W, int H, int bs) {
    int nx = (int) std::ceil(W / (float) bs), ny = (int) std::ceil(H / (float) bs);
    int left = nx * ny; std::vector<std::pair<int, int>> out;
    int bx = nx / 2, by = ny / 2, dir = 0, stepsLeft = 1, numSteps = 1;      // ERight=0, EDown, ELeft, EUp
    while (left > 0) {
        out.emplace_back(bx, by);
        if (--left == 0) break;
        do {
            switch (dir) { case 0: ++bx; break; case 1: ++by; break; case 2: --bx; break; case 3: --by; break; }
            if (--stepsLeft == 0) { dir = (dir + 1) % 4; if (dir == 2 || dir == 0) ++numSteps; stepsLeft = numSteps; }
        } while (bx < 0 || by < 0 || bx >= nx || by >= ny);
    }
    return out;
}

enum SeedMode { SEED_REFERENCE_BLOCKS = 0, SEED_PER_SAMPLE = 1 };

// Render driver.  SEED_REFERENCE_BLOCKS restates src/render.cpp:134-198 (spp outer loop, 32x32 blocks, one pcg32
// stream per block seeded (offset.x, offset.y), persistent over passes).  SEED_PER_SAMPLE is the convention of
// the GPU path: sample s of pixel (x,y) uses pcg32.seed(initstate = y*W + x, initseq = spp_begin+s)
// and is otherwise consumed in the reference's order.
inline void render(const Accel &acc, Film &film, int spp_begin, int spp_end, int mode, int threads, PathStats *stats_out = nullptr) {
    const Scene &sc = *acc.scene; const Camera &cam = sc.camera;
    const int W = cam.width, H = cam.height, BS = 32;
    auto blocks = spiral_blocks(W, H, BS);
    int nx = (int) std::ceil(W / (float) BS);
    std::vector<Sampler> samplers(blocks.size());
    for (auto &b : blocks) samplers[b.second * nx + b.first].rng.seed((uint64_t) (b.first * BS), (uint64_t) (b.second * BS));
    std::mutex film_mutex; PathStats total;
    for (int k = spp_begin; k < spp_end; ++k) {
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            Film local; PathStats ps;
            for (;;) {
                size_t bi = next.fetch_add(1); if (bi >= blocks.size()) break;
                int ox = blocks[bi].first * BS, oy = blocks[bi].second * BS;
                int bw = std::min(BS, W - ox), bh = std::min(BS, H - oy);
                Sampler &smp = samplers[blocks[bi].second * nx + blocks[bi].first];
                std::vector<std::pair<V2, V3>> out; out.reserve((size_t) bw * bh);
                for (int y = 0; y < bh; ++y) for (int x = 0; x < bw; ++x) {
                    Sampler local_smp; Sampler *s = &smp;
                    if (mode == SEED_PER_SAMPLE) { local_smp.rng.seed((uint64_t) (y + oy) * (uint64_t) W + (uint64_t) (x + ox), (uint64_t) k); s = &local_smp; }
                    V2 j = s->next2D();
                    V2 pixelSample = mk2((float) (x + ox) + j.x, (float) (y + oy) + j.y);
                    (void) s->next2D();                                   // apertureSample (src/render.cpp:86)
                    Ray ray; cam.sample_ray(ray, pixelSample);
                    V3 value = scene_li(acc, *s, ray, &ps);
                    out.emplace_back(pixelSample, value);
                }
                std::lock_guard<std::mutex> lk(film_mutex);
                for (auto &o : out) film.put(o.first, o.second, ox, oy);
            }
            std::lock_guard<std::mutex> lk(film_mutex);
            total.closest += ps.closest; total.shadow += ps.shadow; total.bounces += ps.bounces;
            total.tc.nodes += ps.tc.nodes; total.tc.tris += ps.tc.tris;
        };
        if (threads <= 1) worker();
        else { std::vector<std::thread> th; for (int i = 0; i < threads; ++i) th.emplace_back(worker); for (auto &t : th) t.join(); }
    }
    if (stats_out) *stats_out = total;
}

// ---- t-test harness (src/ttest.cpp:152-185): mean / variance of luminance over N paths -------------------
inline void ttest_scene(const Accel &acc, Sampler &smp, int n, double &mean, double &variance) {
    const Camera &cam = acc.scene->camera;
    mean = 0; variance = 0;
    for (int k = 0; k < n; ++k) {
        V2 u = smp.next2D();
        V2 pixelSample = mk2(u.x * (float) cam.width, u.y * (float) cam.height);
        (void) smp.next2D();
        Ray ray; cam.sample_ray(ray, pixelSample);
        V3 value = scene_li(acc, smp, ray);
        double result = (double) luminance(value);
        double delta = result - mean;
        mean += delta / (double) (k + 1);
        variance += delta * (result - mean);
    }
    variance /= n - 1;
}

// ---- NL-means filter pass (cuda/filter.cc.cu:53-110, copy_weighted_data_to_surface :33-48) -------------
// image: sRGB-encoded RGB, indexed [x*height + y] in the CUDA fork; here plain row-major (y*w + x) — the
// arithmetic per pixel is identical, only the storage order differs.
inline void nl_means(const float *img, float *out, int w, int h, int r, int f, float sigma, float hh) {
    const float sigma2 = sigma * sigma, h2 = hh * hh;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        int minX = std::max(0, x - r), maxX = std::min(w - 1, x + r), minY = std::max(0, y - r), maxY = std::min(h - 1, y + r);
        float wsum = 0; float acc[3] = {0, 0, 0};
        for (int i = minX; i <= maxX; ++i) for (int j = minY; j <= maxY; ++j) {
            float d2 = 0;
            int pminx = std::max(std::max(-f, -x), -i), pmaxx = std::min(std::min(f, w - x - 1), w - i - 1);
            int pminy = std::max(std::max(-f, -y), -j), pmaxy = std::min(std::min(f, h - y - 1), h - j - 1);
            for (int k = pminx; k <= pmaxx; ++k) for (int l = pminy; l <= pmaxy; ++l) {
                const float *p = &img[((size_t) (y + l) * w + (x + k)) * 3], *q = &img[((size_t) (j + l) * w + (i + k)) * 3];
                float wk = std::exp(-(k * k + l * l) / (2.0f * sigma2));
                float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                float patchv = (dx * dx + dy * dy + dz * dz) / h2;
                d2 += wk * patchv;
            }
            d2 /= ((pmaxx - pminx + 1) * (pmaxy - pminy + 1));
            float wgt = std::exp(-std::max(d2, 0.0f));
            wsum += wgt;
            const float *c = &img[((size_t) j * w + i) * 3];
            acc[0] += wgt * c[0]; acc[1] += wgt * c[1]; acc[2] += wgt * c[2];
        }
        float *o = &out[((size_t) y * w + x) * 3];
        o[0] = acc[0] / wsum; o[1] = acc[1] / wsum; o[2] = acc[2] / wsum;
    }
}

} // namespace orc
