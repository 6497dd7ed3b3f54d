// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link or call this.
//
// Scalar float math restating the arithmetic ORDER of the reference's Eigen expressions, so the
// restatement reproduces the reference bit-for-bit where that is possible (x86-64, SSE2 scalar math,
// no FMA contraction: build with -ffp-contract=off).
//
// Reference: include/nori/vector.h, include/nori/frame.h:33-63, include/nori/bbox.h:42-397,
//            include/nori/ray.h:38-100, src/common.cpp:274-314 (coordinateSystem, fresnel),
//            vendored Eigen 3.2.90 (ext/nanogui/ext/eigen): Redux.h unrolled sum of 3 terms is
//            x0 + (x1 + x2); coefficient-based small products are sequential ((x0+x1)+x2).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <algorithm>

namespace orc {

static const float kEpsilon = 1e-4f;                       // include/nori/common.h:51
static const float kPi      = 3.14159265358979323846f;     // include/nori/common.h:56 (float literal!)
static const float kInvPi   = 0.31830988618379067154f;     // include/nori/common.h:57
static const float kInf     = std::numeric_limits<float>::infinity();

struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 mk3(float a, float b, float c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
inline V3 mk3(float a) { return mk3(a, a, a); }
inline V2 mk2(float a, float b) { V2 r; r.x = a; r.y = b; return r; }

inline V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }   // Color3f is an Eigen Array
inline V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }     // true division per component

// Eigen redux of 3 terms: x0 + (x1 + x2)
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float sqnorm(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(sqnorm(a)); }
inline V3 normalized(V3 a) { float n = norm(a); return a / n; }                  // Dot.h:114-120: n / n.norm()
// Reductions over DYNAMIC-size Eigen expressions (anything built directly on m_N.col(i) / m_V.col(i) blocks,
// e.g. src/mesh.cpp:69-73,154-157) are not unrolled: Redux.h falls back to the sequential loop (x0+x1)+x2.
inline V3 normalized_seq(V3 a) { float n = std::sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); return a / n; }
inline V3 cross(V3 a, V3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float maxcoeff(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }
inline bool is_zero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }

// Color3f::isValid (src/common.cpp:224-231) and getLuminance (:233-235)
inline bool color_valid(V3 c) {
    for (int i = 0; i < 3; ++i) { float v = c[i]; if (v < 0 || !std::isfinite(v)) return false; }
    return true;
}
inline float luminance(V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }

// src/common.cpp:274-283
inline void coordinate_system(V3 a, V3 &b, V3 &c) {
    if (std::abs(a.x) > std::abs(a.y)) {
        float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        c = mk3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        c = mk3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

// include/nori/frame.h:33-63
struct Frame {
    V3 s, t, n;
    Frame() {}
    explicit Frame(V3 nn) : n(nn) { coordinate_system(nn, s, t); }
    V3 toLocal(V3 v) const { return mk3(dot(v, s), dot(v, t), dot(v, n)); }
    V3 toWorld(V3 v) const { return (s * v.x + t * v.y) + n * v.z; }
};
inline float tan_theta(V3 v) {                              // include/nori/frame.h:82-87
    float temp = 1 - v.z * v.z;
    if (temp <= 0.0f) return 0.0f;
    return std::sqrt(temp) / v.z;
}

// src/common.cpp:285-314
inline float fresnel(float cosThetaI, float extIOR, float intIOR) {
    float etaI = extIOR, etaT = intIOR;
    if (extIOR == intIOR) return 0.0f;
    if (cosThetaI < 0.0f) { std::swap(etaI, etaT); cosThetaI = -cosThetaI; }
    float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1 - cosThetaI * cosThetaI);
    if (sinThetaTSqr > 1.0f) return 1.0f;
    float cosThetaT = std::sqrt(1.0f - sinThetaTSqr);
    float Rs = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    float Rp = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    return (Rs * Rs + Rp * Rp) / 2.0f;
}

// include/nori/ray.h:38-100
struct Ray {
    V3 o, d, dRcp;
    float mint, maxt;
    Ray() : mint(kEpsilon), maxt(kInf) {}
    void update() { dRcp = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
};

// include/nori/bbox.h
struct BBox {
    V3 mn, mx;
    BBox() { reset(); }
    explicit BBox(V3 p) : mn(p), mx(p) {}
    void reset() { mn = mk3(kInf); mx = mk3(-kInf); }
    void expand(V3 p) {
        mn = mk3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
        mx = mk3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
    }
    void expand(const BBox &b) { expand_min(b.mn); expand_max(b.mx); }
    void expand_min(V3 p) { mn = mk3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z)); }
    void expand_max(V3 p) { mx = mk3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z)); }
    static BBox merge(const BBox &a, const BBox &b) { BBox r = a; r.expand(b); return r; }
    // bbox.h:87-100
    float surface_area() const {
        V3 d = mx - mn;
        float result = 0.0f;
        for (int i = 0; i < 3; ++i) {
            float term = 1.0f;
            for (int j = 0; j < 3; ++j) { if (i == j) continue; term *= d[j]; }
            result += term;
        }
        return 2.0f * result;
    }
    // bbox.h:308-317
    int largest_axis() const {
        V3 e = mx - mn;
        if (e.x >= e.y && e.x >= e.z) return 0;
        else if (e.y >= e.x && e.y >= e.z) return 1;
        else return 2;
    }
    // bbox.h:336-363 — slab test; nearT is returned so callers can count/inspect
    bool ray_intersect(const Ray &ray) const {
        float nearT = -kInf, farT = kInf;
        for (int i = 0; i < 3; i++) {
            float origin = ray.o[i];
            float minVal = mn[i], maxVal = mx[i];
            if (ray.d[i] == 0) {
                if (origin < minVal || origin > maxVal) return false;
            } else {
                float t1 = (minVal - origin) * ray.dRcp[i];
                float t2 = (maxVal - origin) * ray.dRcp[i];
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return ray.mint <= farT && nearT <= ray.maxt;
    }
};

// 4x4 matrix, row-major storage m[r][c]
struct M4 {
    float m[4][4];
    static M4 identity() { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.f : 0.f; return r; }
};
// sequential coefficient-based product (ProductEvaluators.h:209-227)
inline M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = a.m[i][0] * b.m[0][j];
        for (int k = 1; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
        r.m[i][j] = s;
    }
    return r;
}
// General inverse in double (Gauss-Jordan, partial pivoting), rounded to float.  The reference uses Eigen's
// SSE cofactor inverse (LU/arch/Inverse_SSE.h) whose last-ulp behaviour is not restated: matrices agree to
// ~1e-7 relative, see DESIGN.md "float parity".
inline M4 inverse(const M4 &a) {
    double w[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { w[i][j] = a.m[i][j]; w[i][j + 4] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (p != c) for (int j = 0; j < 8; ++j) std::swap(w[p][j], w[c][j]);
        double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = w[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j]; }
    }
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (float) w[i][j + 4];
    return r;
}

// include/nori/transform.h:35-97
struct Transform {
    M4 mat, inv;
    Transform() : mat(M4::identity()), inv(M4::identity()) {}
    explicit Transform(const M4 &m) : mat(m), inv(inverse(m)) {}
    Transform(const M4 &m, const M4 &i) : mat(m), inv(i) {}
    Transform inverse_t() const { return Transform(inv, mat); }
    V3 apply_vector(V3 v) const {       // topLeftCorner<3,3>() * v, sequential
        V3 r;
        r.x = (mat.m[0][0] * v.x + mat.m[0][1] * v.y) + mat.m[0][2] * v.z;
        r.y = (mat.m[1][0] * v.x + mat.m[1][1] * v.y) + mat.m[1][2] * v.z;
        r.z = (mat.m[2][0] * v.x + mat.m[2][1] * v.y) + mat.m[2][2] * v.z;
        return r;
    }
    V3 apply_normal(V3 n) const {       // inverse.topLeftCorner<3,3>().transpose() * n
        V3 r;
        r.x = (inv.m[0][0] * n.x + inv.m[1][0] * n.y) + inv.m[2][0] * n.z;
        r.y = (inv.m[0][1] * n.x + inv.m[1][1] * n.y) + inv.m[2][1] * n.z;
        r.z = (inv.m[0][2] * n.x + inv.m[1][2] * n.y) + inv.m[2][2] * n.z;
        return r;
    }
    V3 apply_point(V3 p) const {        // (M * [p,1]).head<3>() / w  (packet path: ((c0*p0 + c1*p1) + c2*p2) + c3)
        float r[4];
        for (int i = 0; i < 4; ++i)
            r[i] = ((mat.m[i][0] * p.x + mat.m[i][1] * p.y) + mat.m[i][2] * p.z) + mat.m[i][3] * 1.0f;
        return mk3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
    }
};

// ext/pcg32/pcg32.h:38-110
struct Pcg32 {
    uint64_t state, inc;
    Pcg32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    void seed(uint64_t initstate, uint64_t initseq = 1) {
        state = 0U; inc = (initseq << 1u) | 1u;
        next_uint(); state += initstate; next_uint();
    }
    uint32_t next_uint() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
    }
    float next_float() {
        union { uint32_t u; float f; } x;
        x.u = (next_uint() >> 9) | 0x3f800000UL;
        return x.f - 1.0f;
    }
};

} // namespace orc
