// ORACLE — TEST INFRASTRUCTURE ONLY.
// Minimal XML reader sufficient for Nori scene files (the reference uses pugixml, src/parser.cpp:30-31):
// elements, attributes in single or double quotes, comments, <?xml ...?> declarations, self-closing tags.
#pragma once
#include <string>
#include <vector>
#include <stdexcept>
#include <fstream>
#include <sstream>
#include <memory>

namespace orc {

struct XmlNode {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;
    const char *attr(const char *k) const {
        for (auto &a : attrs) if (a.first == k) return a.second.c_str();
        return "";
    }
    bool has(const char *k) const { for (auto &a : attrs) if (a.first == k) return true; return false; }
};

class XmlParser {
public:
    explicit XmlParser(const std::string &text) : s(text), i(0) {}
    std::unique_ptr<XmlNode> parse_document() {
        skip_misc();
        auto root = parse_element();
        return root;
    }
private:
    const std::string &s; size_t i;
    [[noreturn]] void fail(const std::string &m) { throw std::runtime_error("XML parse error at byte " + std::to_string(i) + ": " + m); }
    void skip_ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }
    bool starts(const char *p) { return s.compare(i, strlen_(p), p) == 0; }
    static size_t strlen_(const char *p) { size_t n = 0; while (p[n]) ++n; return n; }
    void skip_misc() {
        for (;;) {
            skip_ws();
            if (starts("<?")) { size_t e = s.find("?>", i); if (e == std::string::npos) fail("unterminated <?"); i = e + 2; }
            else if (starts("<!--")) { size_t e = s.find("-->", i); if (e == std::string::npos) fail("unterminated comment"); i = e + 3; }
            else break;
        }
    }
    std::string parse_name() {
        size_t b = i;
        while (i < s.size() && (isalnum((unsigned char) s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
        if (b == i) fail("expected name");
        return s.substr(b, i - b);
    }
    std::unique_ptr<XmlNode> parse_element() {
        if (i >= s.size() || s[i] != '<') fail("expected '<'");
        ++i;
        std::unique_ptr<XmlNode> n(new XmlNode());
        n->name = parse_name();
        for (;;) {
            skip_ws();
            if (i >= s.size()) fail("unexpected end");
            if (s[i] == '/') { if (s.compare(i, 2, "/>") != 0) fail("expected '/>'"); i += 2; return n; }
            if (s[i] == '>') { ++i; break; }
            std::string k = parse_name();
            skip_ws();
            if (i >= s.size() || s[i] != '=') fail("expected '='");
            ++i; skip_ws();
            char q = s[i];
            if (q != '"' && q != '\'') fail("expected quote");
            size_t e = s.find(q, i + 1);
            if (e == std::string::npos) fail("unterminated attribute");
            n->attrs.emplace_back(k, s.substr(i + 1, e - i - 1));
            i = e + 1;
        }
        for (;;) {
            skip_misc();
            // text content is ignored (Nori has none)
            while (i < s.size() && s[i] != '<') ++i;
            if (i >= s.size()) fail("unexpected end in <" + n->name + ">");
            if (starts("<!--") || starts("<?")) continue;
            if (starts("</")) {
                i += 2; std::string c = parse_name();
                if (c != n->name) fail("mismatched </" + c + ">");
                skip_ws(); if (s[i] != '>') fail("expected '>'"); ++i;
                return n;
            }
            n->children.push_back(parse_element());
        }
    }
};

inline std::unique_ptr<XmlNode> xml_load_file(const std::string &path, std::string *text_keep = nullptr) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open \"" + path + "\"");
    std::stringstream ss; ss << f.rdbuf();
    std::string text = ss.str();
    XmlParser p(text);
    auto r = p.parse_document();
    if (text_keep) *text_keep = text;
    return r;
}

} // namespace orc
