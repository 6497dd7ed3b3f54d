// Host-side plugin surface of the MI355X path_mis renderer.
//
// This mirrors the reference's host API for the hot path — same class names, same XML vocabulary, same
// registration macro, same error behaviour (NoriException) — so scenes, plugins and tests written against Nori
// keep working, while everything per-ray runs in libnori_hip.so behind include/nori_hip.h.
//
// Reference: include/nori/object.h:31-197 (NoriObject, NoriObjectFactory, NORI_REGISTER_CLASS),
//            include/nori/proplist.h:32-128, include/nori/common.h:148-153 (NoriException),
//            include/nori/transform.h:35-97, include/nori/bbox.h, src/parser.cpp:28-340.
#pragma once
#include <cstdint>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/nori_hip.h"

namespace nori {

/// printf-style exception, as include/nori/common.h:148-153
class NoriException : public std::runtime_error {
public:
    explicit NoriException(const std::string &msg) : std::runtime_error(msg) {}
    static NoriException format(const char *fmt, ...) {
        char buf[2048]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
        return NoriException(buf);
    }
};

// ---- small fixed-size math.  Arithmetic order follows the reference's Eigen expressions ------
struct Vector3f { float x = 0, y = 0, z = 0; Vector3f() {} Vector3f(float a, float b, float c) : x(a), y(b), z(c) {} explicit Vector3f(float a) : x(a), y(a), z(a) {}
    float operator[](int i) const { return (&x)[i]; } float &operator[](int i) { return (&x)[i]; } };
typedef Vector3f Point3f; typedef Vector3f Normal3f; typedef Vector3f Color3f;
/// include/nori/ray.h:38-100 (dRcp is recomputed on the device)
struct Ray3f {
    Point3f o; Vector3f d; float mint = 1e-4f, maxt = __builtin_huge_valf();
    Ray3f() {}
    Ray3f(const Point3f &o_, const Vector3f &d_, float mint_ = 1e-4f, float maxt_ = __builtin_huge_valf()) : o(o_), d(d_), mint(mint_), maxt(maxt_) {}
};
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
inline Vector3f operator+(Vector3f a, Vector3f b) { return Vector3f(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vector3f operator-(Vector3f a, Vector3f b) { return Vector3f(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector3f operator*(float s, Vector3f a) { return Vector3f(s * a.x, s * a.y, s * a.z); }
inline Vector3f operator/(Vector3f a, float s) { return Vector3f(a.x / s, a.y / s, a.z / s); }
inline float dot(Vector3f a, Vector3f b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }           // Eigen 3-term redux
inline Vector3f cross(Vector3f a, Vector3f b) { return Vector3f(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float norm(Vector3f a) { return std::sqrt(dot(a, a)); }
inline Vector3f normalized(Vector3f a) { return a / norm(a); }

struct BoundingBox3f {
    Point3f min, max;
    BoundingBox3f() { reset(); }
    explicit BoundingBox3f(Point3f p) : min(p), max(p) {}
    void reset() { float inf = std::numeric_limits<float>::infinity(); min = Point3f(inf); max = Point3f(-inf); }
    void expandBy(Point3f p) { for (int i = 0; i < 3; ++i) { min[i] = std::min(min[i], p[i]); max[i] = std::max(max[i], p[i]); } }
    void expandBy(const BoundingBox3f &b) { for (int i = 0; i < 3; ++i) { min[i] = std::min(min[i], b.min[i]); max[i] = std::max(max[i], b.max[i]); } }
    float getSurfaceArea() const {                       // include/nori/bbox.h:87-100
        float d0 = max.x - min.x, d1 = max.y - min.y, d2 = max.z - min.z;
        float result = 0.0f;
        result += (1.0f * d1) * d2; result += (1.0f * d0) * d2; result += (1.0f * d0) * d1;
        return 2.0f * result;
    }
    int getLargestAxis() const {                         // include/nori/bbox.h:308-317
        float e0 = max.x - min.x, e1 = max.y - min.y, e2 = max.z - min.z;
        if (e0 >= e1 && e0 >= e2) return 0; else if (e1 >= e0 && e1 >= e2) return 1; else return 2;
    }
};

/// Homogeneous transform + inverse, row-major (include/nori/transform.h)
struct Transform {
    float m[4][4], inv[4][4];
    Transform();
    explicit Transform(const float mat[4][4]);
    Transform(const float mat[4][4], const float inverse[4][4]);
    Transform inverse() const { return Transform(inv, m); }
    Vector3f operator*(const Vector3f &v) const;   // direction
    Point3f point(const Point3f &p) const;         // Point3f overload of operator* in the reference
    Normal3f normal(const Normal3f &n) const;      // Normal3f overload
};

/// Property list handed to plugin constructors (include/nori/proplist.h)
class PropertyList {
public:
    void setBoolean(const std::string &n, bool v) { b[n] = v; }
    void setInteger(const std::string &n, int v) { i[n] = v; }
    void setFloat(const std::string &n, float v) { f[n] = v; }
    void setString(const std::string &n, const std::string &v) { s[n] = v; }
    void setColor(const std::string &n, const Color3f &v) { c[n] = v; }
    void setPoint3(const std::string &n, const Point3f &v) { p[n] = v; }
    void setVector3(const std::string &n, const Vector3f &v) { p[n] = v; }
    void setTransform(const std::string &n, const Transform &v) { t[n] = v; }
    void setPoint2(const std::string &n, const Point2f &v) { p2[n] = v; }       // 2-component <point>/<vector> (src/parser.cpp:242-266)
    void setVector2(const std::string &n, const Point2f &v) { p2[n] = v; }
    bool has(const std::string &n) const { return b.count(n) || i.count(n) || f.count(n) || s.count(n) || c.count(n) || p.count(n) || t.count(n) || p2.count(n); }
#define NORI_GETTER(Type, Name, member) \
    Type get##Name(const std::string &n) const { auto it = member.find(n); if (it == member.end()) { if (has(n)) throw NoriException::format("Property '%s' has the wrong type!", n.c_str()); throw NoriException::format("Property '%s' is missing!", n.c_str()); } return it->second; } \
    Type get##Name(const std::string &n, const Type &d) const { auto it = member.find(n); if (it == member.end()) { if (has(n)) throw NoriException::format("Property '%s' has the wrong type!", n.c_str()); return d; } return it->second; }
    NORI_GETTER(bool, Boolean, b)
    NORI_GETTER(int, Integer, i)
    NORI_GETTER(float, Float, f)
    NORI_GETTER(std::string, String, s)
    NORI_GETTER(Color3f, Color, c)
    NORI_GETTER(Point3f, Point3, p)
    NORI_GETTER(Vector3f, Vector3, p)
    NORI_GETTER(Transform, Transform, t)
    NORI_GETTER(Point2f, Point2, p2)
    NORI_GETTER(Point2f, Vector2, p2)
#undef NORI_GETTER
private:
    std::map<std::string, bool> b; std::map<std::string, int> i; std::map<std::string, float> f; std::map<std::string, std::string> s;
    std::map<std::string, Color3f> c; std::map<std::string, Point3f> p; std::map<std::string, Transform> t; std::map<std::string, Point2f> p2;
};

/// Base class of everything that can be named in a scene file (include/nori/object.h:31-123)
class NoriObject {
public:
    enum EClassType { EScene = 0, EMesh, ETexture, EBSDF, EPhaseFunction, EEmitter, EMedium, ECamera, EIntegrator, ESampler, ETest,
                      EReconstructionFilter, ENormalModifier, EEnvMap, EClassTypeCount };
    virtual ~NoriObject() {}
    virtual EClassType getClassType() const = 0;
    virtual void addChild(NoriObject *child);
    virtual void setParent(NoriObject *) {}
    virtual void activate() {}
    virtual std::string toString() const = 0;
    void setIdName(const std::string &name) { m_idname = name; }
    const std::string &getIdName() const { return m_idname; }
    static std::string classTypeName(EClassType type);
private:
    std::string m_idname;
};

/// String -> constructor registry (include/nori/object.h:130-174); a later registration under the same name replaces the earlier
class NoriObjectFactory {
public:
    typedef std::function<NoriObject *(const PropertyList &)> Constructor;
    static void registerClass(const std::string &name, const Constructor &constr);
    static NoriObject *createInstance(const std::string &name, const PropertyList &propList);
    static bool isRegistered(const std::string &name);
private:
    static std::map<std::string, Constructor> *m_constructors;
};

#define NORI_REGISTER_CLASS(cls, name) \
    cls *cls##_create(const PropertyList &list) { return new cls(list); } \
    static struct cls##_ { cls##_() { NoriObjectFactory::registerClass(name, cls##_create); } } cls##__NORI_;

/// Loads a scene (or test) description; throws NoriException with file context on errors (src/parser.cpp:28)
NoriObject *loadFromXML(const std::string &filename);

/// Directory stack used to resolve relative resource paths (ext/filesystem/resolver.h as used in src/render.cpp:110)
class FileResolver { public: void prepend(const std::string &dir) { m_paths.insert(m_paths.begin(), dir); } std::string resolve(const std::string &name) const; private: std::vector<std::string> m_paths; };
FileResolver *getFileResolver();

std::vector<std::string> tokenize(const std::string &s, const std::string &delim = ", ", bool includeEmpty = false);
float toFloat(const std::string &str);
int toInt(const std::string &str);
unsigned int toUInt(const std::string &str);
Vector3f toVector3f(const std::string &str);
inline float degToRad(float v) { return v * (3.14159265358979323846f / 180.0f); }
std::string indent(const std::string &string, int amount = 2);

} // namespace nori
