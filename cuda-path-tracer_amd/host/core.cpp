// Object registry, property parsing, transforms and the XML scene parser.
// Reference: src/object.cpp:25-29, src/common.cpp:39-143, src/parser.cpp:28-340, include/nori/transform.h.
// Derivative text: parseTag below follows src/parser.cpp:137-336 check by check (same flags, same order, same error texts),
// because which malformed files are rejected, and with which message, is part of the plugin surface the reference's scene files
// and tools rely on.  The XML reader underneath it, the transform composition and the 4x4 inverse are this repository's own
// (the reference uses pugixml and Eigen).
#include "scene.h"
#include <cstring>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <set>

namespace nori {

// ---- registry ---------------------------------------------------------------------------------------------
std::map<std::string, NoriObjectFactory::Constructor> *NoriObjectFactory::m_constructors = nullptr;
void NoriObjectFactory::registerClass(const std::string &name, const Constructor &constr) {
    if (!m_constructors) m_constructors = new std::map<std::string, Constructor>();
    (*m_constructors)[name] = constr;
}
NoriObject *NoriObjectFactory::createInstance(const std::string &name, const PropertyList &propList) {
    if (!m_constructors || m_constructors->find(name) == m_constructors->end())
        throw NoriException::format("A constructor for class \"%s\" could not be found!", name.c_str());
    return (*m_constructors)[name](propList);
}
bool NoriObjectFactory::isRegistered(const std::string &name) { return m_constructors && m_constructors->count(name); }

void NoriObject::addChild(NoriObject *child) {
    throw NoriException::format("NoriObject::addChild() is not implemented for objects of type '%s'!", classTypeName(getClassType()).c_str());
}
std::string NoriObject::classTypeName(EClassType type) {
    switch (type) {
        case EScene: return "scene"; case EMesh: return "mesh"; case ETexture: return "texture"; case EBSDF: return "bsdf";
        case EEmitter: return "emitter"; case ECamera: return "camera"; case EIntegrator: return "integrator"; case ESampler: return "sampler";
        case ETest: return "test"; case EReconstructionFilter: return "rfilter"; case ENormalModifier: return "normalModifier"; case EEnvMap: return "enviromentMap";
        case EPhaseFunction: return "phase"; case EMedium: return "medium"; default: return "<unknown>";
    }
}

// ---- strings ----------------------------------------------------------------------------------------------
std::vector<std::string> tokenize(const std::string &s, const std::string &delim, bool includeEmpty) {
    // a token is emitted whenever the delimiter is not directly at the scan position (or always, with includeEmpty);
    // like the reference, a string that ends in a delimiter yields one trailing empty token
    std::vector<std::string> tokens; size_t last = 0;
    for (;;) {
        size_t pos = s.find_first_of(delim, last);
        if (pos != last || includeEmpty) tokens.push_back(s.substr(last, pos == std::string::npos ? std::string::npos : pos - last));
        if (pos == std::string::npos) break;
        last = pos + 1;
    }
    return tokens;
}
float toFloat(const std::string &str) {
    char *end = nullptr; float r = strtof(str.c_str(), &end);
    if (*end != '\0') throw NoriException::format("Could not parse floating point value \"%s\"", str.c_str());
    return r;
}
int toInt(const std::string &str) {
    char *end = nullptr; int r = (int) strtol(str.c_str(), &end, 10);
    if (*end != '\0') throw NoriException::format("Could not parse integer value \"%s\"", str.c_str());
    return r;
}
unsigned int toUInt(const std::string &str) {
    char *end = nullptr; unsigned int r = (int) strtoul(str.c_str(), &end, 10);
    if (*end != '\0') throw NoriException::format("Could not parse integer value \"%s\"", str.c_str());
    return r;
}
Vector3f toVector3f(const std::string &str) {
    std::vector<std::string> t = tokenize(str);
    if (t.size() != 3) throw NoriException("Expected 3 values");
    return Vector3f(toFloat(t[0]), toFloat(t[1]), toFloat(t[2]));
}
static bool toBool(const std::string &str) {
    std::string v = str; for (auto &c : v) c = (char) tolower(c);
    if (v == "false") return false; if (v == "true") return true;
    throw NoriException::format("Could not parse boolean value \"%s\"", str.c_str());
}
std::string indent(const std::string &string, int amount) {
    std::istringstream iss(string); std::ostringstream oss; std::string spacer(amount, ' '); bool first = true;
    for (std::string line; std::getline(iss, line);) { if (!first) oss << spacer; oss << line; if (!iss.eof()) oss << "\n"; first = false; }
    return oss.str();
}

// ---- file resolver ----------------------------------------------------------------------------------------
std::string FileResolver::resolve(const std::string &name) const {
    if (!name.empty() && name[0] == '/') return name;
    for (auto &d : m_paths) { std::string c = d + "/" + name; std::ifstream f(c); if (f.good()) return c; }
    return name;
}
FileResolver *getFileResolver() { static FileResolver *r = new FileResolver(); return r; }

// ---- transforms -------------------------------------------------------------------------------------------
// General 4x4 inverse by Gauss-Jordan in double, rounded to float.  The reference uses Eigen's SSE cofactor
// kernel; results agree to the last ulp or two, and the matrices are inputs of the device path either way.
static void invert4(const float a[4][4], float out[4][4]) {
    double w[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { w[i][j] = a[i][j]; w[i][j + 4] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (p != c) for (int j = 0; j < 8; ++j) std::swap(w[p][j], w[c][j]);
        double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = w[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i][j] = (float) w[i][j + 4];
}
Transform::Transform() { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = inv[i][j] = (i == j) ? 1.f : 0.f; }
Transform::Transform(const float mat[4][4]) { std::memcpy(m, mat, sizeof(m)); invert4(m, inv); }
Transform::Transform(const float mat[4][4], const float inverse[4][4]) { std::memcpy(m, mat, sizeof(m)); std::memcpy(inv, inverse, sizeof(inv)); }
Vector3f Transform::operator*(const Vector3f &v) const {
    return Vector3f((m[0][0] * v.x + m[0][1] * v.y) + m[0][2] * v.z, (m[1][0] * v.x + m[1][1] * v.y) + m[1][2] * v.z, (m[2][0] * v.x + m[2][1] * v.y) + m[2][2] * v.z);
}
Normal3f Transform::normal(const Normal3f &n) const {
    return Normal3f((inv[0][0] * n.x + inv[1][0] * n.y) + inv[2][0] * n.z, (inv[0][1] * n.x + inv[1][1] * n.y) + inv[2][1] * n.z, (inv[0][2] * n.x + inv[1][2] * n.y) + inv[2][2] * n.z);
}
Point3f Transform::point(const Point3f &p) const {
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = ((m[i][0] * p.x + m[i][1] * p.y) + m[i][2] * p.z) + m[i][3] * 1.0f;
    return Point3f(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}

// ---- XML --------------------------------------------------------------------------------------------------
namespace {
struct XmlNode { std::string name; std::vector<std::pair<std::string, std::string>> attrs; std::vector<XmlNode> children; size_t offset = 0;
    const std::string *find(const std::string &k) const { for (auto &a : attrs) if (a.first == k) return &a.second; return nullptr; }
    std::string attr(const std::string &k) const { const std::string *v = find(k); return v ? *v : std::string(); } };

struct XmlReader {
    const std::string &s; size_t i = 0; std::string file;
    XmlReader(const std::string &text, const std::string &f) : s(text), file(f) {}
    std::string where(size_t pos) const {
        int line = 1, col = 0;
        for (size_t k = 0; k < pos && k < s.size(); ++k) { if (s[k] == '\n') { ++line; col = 0; } else ++col; }
        char buf[64]; snprintf(buf, sizeof(buf), "row %d, col %d", line, col); return buf;
    }
    [[noreturn]] void fail(const std::string &what) const { throw NoriException::format("Error while parsing \"%s\": %s (at %s)", file.c_str(), what.c_str(), where(i).c_str()); }
    void ws() { while (i < s.size() && isspace((unsigned char) s[i])) ++i; }
    bool at(const char *lit) const { return s.compare(i, strlen(lit), lit) == 0; }
    void misc() {
        for (;;) { ws();
            if (at("<?")) { size_t e = s.find("?>", i); if (e == std::string::npos) fail("unterminated declaration"); i = e + 2; }
            else if (at("<!--")) { size_t e = s.find("-->", i); if (e == std::string::npos) fail("unterminated comment"); i = e + 3; }
            else break; }
    }
    std::string name() { size_t b = i; while (i < s.size() && (isalnum((unsigned char) s[i]) || strchr("_-:.", s[i]))) ++i; if (b == i) fail("expected a name"); return s.substr(b, i - b); }
    XmlNode element() {
        XmlNode n; n.offset = i;
        if (i >= s.size() || s[i] != '<') fail("expected '<'");
        ++i; n.name = name();
        for (;;) {
            ws(); if (i >= s.size()) fail("unexpected end of file");
            if (s[i] == '/') { if (!at("/>")) fail("expected '/>'"); i += 2; return n; }
            if (s[i] == '>') { ++i; break; }
            std::string k = name(); ws(); if (i >= s.size() || s[i] != '=') fail("expected '='"); ++i; ws();
            char q = s[i]; if (q != '"' && q != '\'') fail("expected a quoted attribute value");
            size_t e = s.find(q, i + 1); if (e == std::string::npos) fail("unterminated attribute value");
            std::string v = s.substr(i + 1, e - i - 1);
            for (auto &c : v) if (c == '\n' || c == '\t' || c == '\r') c = ' ';       // attribute whitespace conversion, as pugixml's default
            n.attrs.emplace_back(k, v); i = e + 1;
        }
        for (;;) {
            misc();
            if (i >= s.size()) fail("unexpected end of file inside <" + n.name + ">");
            if (s[i] != '<') fail("unexpected content");
            if (at("</")) { i += 2; std::string c = name(); if (c != n.name) fail("mismatched closing tag </" + c + ">"); ws(); if (i >= s.size() || s[i] != '>') fail("expected '>'"); ++i; return n; }
            n.children.push_back(element());
        }
    }
};

enum ETag { EBoolean = NoriObject::EClassTypeCount, EInteger, EFloat, EString, EPoint, EVector, EColor, ETransform, ETranslate, EMatrix, ERotate, EScale, ELookAt, EInvalid };

// Eigen::Affine3f as the parser composes it: linear part + translation; every operation is multiplied from the left
struct Affine { float l[3][3], t[3];
    void identity() { for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) l[i][j] = (i == j); t[i] = 0; } }
    void leftMultiply(const float L[3][3], const float T[3]) {
        Affine r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r.l[i][j] = (L[i][0] * l[0][j] + L[i][1] * l[1][j]) + L[i][2] * l[2][j];
            r.t[i] = (L[i][0] * t[0] + L[i][1] * t[1]) + L[i][2] * t[2];
            if (T) r.t[i] = r.t[i] + T[i];
        }
        *this = r;
    }
    Transform toTransform() const { float m[4][4] = {{0}}; for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[i][j] = l[i][j]; m[i][3] = t[i]; } m[3][3] = 1.f; return Transform(m); }
};
} // namespace

NoriObject *loadFromXML(const std::string &filename) {
    std::ifstream is(filename);
    if (!is.good()) throw NoriException::format("Error while parsing \"%s\": File was not found", filename.c_str());
    std::stringstream ss; ss << is.rdbuf(); std::string text = ss.str();
    XmlReader rd(text, filename);
    rd.misc();
    XmlNode rootNode = rd.element();

    std::map<std::string, int> tags = {
        {"scene", NoriObject::EScene}, {"mesh", NoriObject::EMesh}, {"texture", NoriObject::ETexture}, {"bsdf", NoriObject::EBSDF},
        {"emitter", NoriObject::EEmitter}, {"camera", NoriObject::ECamera}, {"medium", NoriObject::EMedium}, {"phase", NoriObject::EPhaseFunction},
        {"integrator", NoriObject::EIntegrator}, {"sampler", NoriObject::ESampler}, {"rfilter", NoriObject::EReconstructionFilter},
        {"test", NoriObject::ETest}, {"normalModifier", NoriObject::ENormalModifier}, {"enviromentMap", NoriObject::EEnvMap} /* the fork's spelling, cuda/nori/parser.cpp:118 */,
        {"boolean", EBoolean}, {"integer", EInteger}, {"float", EFloat}, {"string", EString}, {"point", EPoint}, {"vector", EVector},
        {"color", EColor}, {"transform", ETransform}, {"translate", ETranslate}, {"matrix", EMatrix}, {"rotate", ERotate}, {"scale", EScale}, {"lookat", ELookAt}};

    auto err = [&](const XmlNode &n, const std::string &what) { return NoriException::format("Error while parsing \"%s\": %s (at %s)", filename.c_str(), what.c_str(), rd.where(n.offset).c_str()); };
    auto checkAttributes = [&](const XmlNode &n, std::set<std::string> attrs) {
        for (auto &a : n.attrs) { auto it = attrs.find(a.first); if (it == attrs.end()) throw err(n, "unexpected attribute \"" + a.first + "\" in \"" + n.name + "\""); attrs.erase(it); }
        if (!attrs.empty()) throw err(n, "missing attribute \"" + *attrs.begin() + "\" in \"" + n.name + "\"");
    };

    Affine transform; transform.identity();
    std::function<NoriObject *(const XmlNode &, PropertyList &, int)> parseTag = [&](const XmlNode &node, PropertyList &list, int parentTag) -> NoriObject * {
        auto it = tags.find(node.name);
        if (it == tags.end()) throw err(node, "unexpected tag \"" + node.name + "\"");
        int tag = it->second;
        bool hasParent = parentTag != EInvalid, parentIsObject = hasParent && parentTag < NoriObject::EClassTypeCount;
        bool currentIsObject = tag < NoriObject::EClassTypeCount, parentIsTransform = parentTag == ETransform;
        bool currentIsTransformOp = tag == ETranslate || tag == ERotate || tag == EScale || tag == ELookAt || tag == EMatrix;
        if (!hasParent && !currentIsObject) throw err(node, "root element \"" + node.name + "\" must be a Nori object");
        if (parentIsTransform != currentIsTransformOp) throw err(node, "transform nodes can only contain transform operations");
        if (hasParent && !parentIsObject && !(parentIsTransform && currentIsTransformOp)) throw err(node, "node \"" + node.name + "\" requires a Nori object as parent");
        if (tag == ETransform) transform.identity();

        PropertyList propList; std::vector<NoriObject *> children;
        for (const XmlNode &ch : node.children) { NoriObject *c = parseTag(ch, propList, tag); if (c) children.push_back(c); }

        NoriObject *result = nullptr;
        try {
            if (currentIsObject) {
                std::string type = tag == NoriObject::EScene ? std::string("scene") : node.attr("type");
                result = NoriObjectFactory::createInstance(type, propList);
                if (result->getClassType() != (NoriObject::EClassType) tag)
                    throw NoriException::format("Unexpectedly constructed an object of type <%s> (expected type <%s>): %s",
                        NoriObject::classTypeName(result->getClassType()).c_str(), NoriObject::classTypeName((NoriObject::EClassType) tag).c_str(), result->toString().c_str());
                result->setIdName(node.attr("name"));
                for (auto ch : children) { result->addChild(ch); ch->setParent(result); }
                result->activate();
            } else {
                switch (tag) {
                    case EString: checkAttributes(node, {"name", "value"}); list.setString(node.attr("name"), node.attr("value")); break;
                    case EFloat: checkAttributes(node, {"name", "value"}); list.setFloat(node.attr("name"), toFloat(node.attr("value"))); break;
                    case EInteger: checkAttributes(node, {"name", "value"}); list.setInteger(node.attr("name"), toInt(node.attr("value"))); break;
                    case EBoolean: checkAttributes(node, {"name", "value"}); list.setBoolean(node.attr("name"), toBool(node.attr("value"))); break;
                    case EPoint: case EVector: {                           // 2 or 3 components (src/parser.cpp:242-266)
                        checkAttributes(node, {"name", "value"});
                        std::vector<std::string> tk = tokenize(node.attr("value"));
                        if (tk.size() == 2) { Point2f v(toFloat(tk[0]), toFloat(tk[1])); if (tag == EPoint) list.setPoint2(node.attr("name"), v); else list.setVector2(node.attr("name"), v); }
                        else if (tk.size() == 3) { if (tag == EPoint) list.setPoint3(node.attr("name"), toVector3f(node.attr("value"))); else list.setVector3(node.attr("name"), toVector3f(node.attr("value"))); }
                        else throw NoriException::format("%s %s (value: %s) is not of size 2 or 3", tag == EPoint ? "Point" : "Vector", node.attr("name").c_str(), node.attr("value").c_str());
                    } break;
                    case EColor: checkAttributes(node, {"name", "value"}); list.setColor(node.attr("name"), toVector3f(node.attr("value"))); break;
                    case ETransform: checkAttributes(node, {"name"}); list.setTransform(node.attr("name"), transform.toTransform()); break;
                    case ETranslate: { checkAttributes(node, {"value"}); Vector3f v = toVector3f(node.attr("value")); for (int i = 0; i < 3; ++i) transform.t[i] += v[i]; } break;
                    case EScale: { checkAttributes(node, {"value"}); Vector3f v = toVector3f(node.attr("value"));
                        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) transform.l[i][j] = v[i] * transform.l[i][j]; transform.t[i] = v[i] * transform.t[i]; } } break;
                    case EMatrix: { checkAttributes(node, {"value"}); std::vector<std::string> tk = tokenize(node.attr("value"));
                        if (tk.size() != 16) throw NoriException("Expected 16 values");
                        float L[3][3], T[3]; for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) L[i][j] = toFloat(tk[i * 4 + j]); T[i] = toFloat(tk[i * 4 + 3]); }
                        transform.leftMultiply(L, T); } break;
                    case ERotate: { checkAttributes(node, {"angle", "axis"}); float angle = degToRad(toFloat(node.attr("angle"))); Vector3f ax = toVector3f(node.attr("axis"));
                        // Eigen::AngleAxis::toRotationMatrix
                        float sn = std::sin(angle), c = std::cos(angle); Vector3f sa = sn * ax, ca = (1.0f - c) * ax; float R[3][3], tmp;
                        tmp = ca.x * ax.y; R[0][1] = tmp - sa.z; R[1][0] = tmp + sa.z;
                        tmp = ca.x * ax.z; R[0][2] = tmp + sa.y; R[2][0] = tmp - sa.y;
                        tmp = ca.y * ax.z; R[1][2] = tmp - sa.x; R[2][1] = tmp + sa.x;
                        R[0][0] = ca.x * ax.x + c; R[1][1] = ca.y * ax.y + c; R[2][2] = ca.z * ax.z + c;
                        transform.leftMultiply(R, nullptr); } break;
                    case ELookAt: { checkAttributes(node, {"origin", "target", "up"});
                        Vector3f origin = toVector3f(node.attr("origin")), target = toVector3f(node.attr("target")), up = toVector3f(node.attr("up"));
                        Vector3f dir = normalized(target - origin), left = normalized(cross(normalized(up), dir)), newUp = normalized(cross(dir, left));
                        float L[3][3] = {{left.x, newUp.x, dir.x}, {left.y, newUp.y, dir.y}, {left.z, newUp.z, dir.z}}; float T[3] = {origin.x, origin.y, origin.z};
                        transform.leftMultiply(L, T); } break;
                    default: throw NoriException::format("Unhandled element \"%s\"", node.name.c_str());
                }
            }
        } catch (const NoriException &e) {
            for (auto ch : children) if (!result) delete ch;
            throw err(node, e.what());
        }
        return result;
    };
    PropertyList list;
    return parseTag(rootNode, list, EInvalid);
}

} // namespace nori
