// yaml_lite.hpp -- the YAML subset needed to accept the reference's
// `params/icp-settings-*.yaml` (lines 7-46) and `params/kitti-default.yaml`
// verbatim: block maps, block sequences of maps, plain/quoted scalars, `#`
// comments, and the mola-yaml preprocessor tokens `$include{...}` and
// `$(mola-dir PKG)` (params/kitti-default.yaml:43,46,50).
//
// Stands in for mrpt::containers::yaml + mola-yaml, which the reference uses
// at src/LidarOdometry.cpp:57-88,102-128 and which are not available here.
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mola_icp_amd {

struct YamlNode {
    enum Kind { Null, Scalar, Map, Seq } kind = Null;
    std::string scalar;
    std::vector<std::pair<std::string, YamlNode>> map;
    std::vector<YamlNode> seq;

    bool is_null() const { return kind == Null; }
    bool is_map() const { return kind == Map; }
    bool is_seq() const { return kind == Seq; }
    bool is_scalar() const { return kind == Scalar; }
    bool has(const std::string& k) const { return find(k) != nullptr; }
    const YamlNode* find(const std::string& k) const
    {
        if (kind != Map) return nullptr;
        for (auto& kv : map)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    const YamlNode& at(const std::string& k) const
    {
        const YamlNode* n = find(k);
        if (!n) throw std::runtime_error("missing YAML entry `" + k + "`");
        return *n;
    }
    std::string as_string() const
    {
        if (kind != Scalar) throw std::runtime_error("YAML node is not a scalar");
        return scalar;
    }
    double as_double() const
    {
        const std::string s = as_string();
        char* end = nullptr;
        const double v = std::strtod(s.c_str(), &end);
        if (end == s.c_str() || *end != '\0') throw std::runtime_error("YAML scalar `" + s + "` is not a number");
        return v;
    }
    long as_int() const
    {
        const double v = as_double();
        if (v != (double)(long)v) throw std::runtime_error("YAML scalar `" + scalar + "` is not an integer");
        return (long)v;
    }
    bool as_bool() const
    {
        std::string s = as_string();
        for (auto& c : s) c = (char)std::tolower((unsigned char)c);
        if (s == "true" || s == "yes" || s == "on" || s == "1") return true;
        if (s == "false" || s == "no" || s == "off" || s == "0") return false;
        throw std::runtime_error("YAML scalar `" + scalar + "` is not a boolean");
    }
};

namespace yaml_detail {

struct Line { int indent; std::string text; int lineno; };

inline std::string rtrim(std::string s)
{
    while (!s.empty() && std::isspace((unsigned char)s.back())) s.pop_back();
    return s;
}
inline std::string trim(std::string s)
{
    s = rtrim(s);
    size_t i = 0;
    while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    return s.substr(i);
}

// strips a trailing comment that is outside quotes and preceded by whitespace (or at column 0)
inline std::string strip_comment(const std::string& s)
{
    bool sq = false, dq = false;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || std::isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
    }
    return s;
}

inline std::string unquote(const std::string& s)
{
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        return s.substr(1, s.size() - 2);
    return s;
}

inline std::vector<Line> split_lines(const std::string& text)
{
    std::vector<Line> out;
    std::istringstream is(text);
    std::string raw;
    int n = 0;
    while (std::getline(is, raw)) {
        ++n;
        if (!raw.empty() && raw.back() == '\r') raw.pop_back();
        std::string s = rtrim(strip_comment(raw));
        if (trim(s).empty()) continue;
        if (trim(s) == "---") continue;
        int ind = 0;
        while ((size_t)ind < s.size() && s[ind] == ' ') ++ind;
        if ((size_t)ind < s.size() && s[ind] == '\t')
            throw std::runtime_error("YAML line " + std::to_string(n) + ": tab indentation is not supported");
        out.push_back({ind, s.substr(ind), n});
    }
    return out;
}

// position of the key/value separator ": " (or trailing ':') outside quotes/braces, npos if none
inline size_t find_colon(const std::string& s)
{
    bool sq = false, dq = false;
    int brace = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if ((c == '{' || c == '(') && !sq && !dq) ++brace;
        else if ((c == '}' || c == ')') && !sq && !dq) --brace;
        else if (c == ':' && !sq && !dq && brace == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) {
            // "::" inside class names (mp2p_icp::ICP) never matches: next char is ':' or prev is ':'
            if (i > 0 && s[i - 1] == ':') continue;
            return i;
        }
    }
    return std::string::npos;
}

struct Parser {
    std::vector<Line> L;
    size_t pos = 0;

    YamlNode parse_block(int indent)
    {
        if (pos >= L.size() || L[pos].indent < indent) return YamlNode{};
        const int ind = L[pos].indent;
        if (L[pos].text.rfind("- ", 0) == 0 || L[pos].text == "-") return parse_seq(ind);
        return parse_map(ind);
    }

    YamlNode parse_seq(int ind)
    {
        YamlNode n;
        n.kind = YamlNode::Seq;
        while (pos < L.size() && L[pos].indent == ind && (L[pos].text.rfind("- ", 0) == 0 || L[pos].text == "-")) {
            std::string rest = L[pos].text.size() > 2 ? L[pos].text.substr(2) : std::string();
            size_t lead = 0;
            while (lead < rest.size() && rest[lead] == ' ') ++lead;
            rest = rest.substr(lead);
            if (rest.empty()) {
                ++pos;
                n.seq.push_back(parse_block(ind + 1));
            } else if (find_colon(rest) != std::string::npos) {
                // "- key: value": a map whose first key sits at column ind+2+lead
                L[pos].indent = ind + 2 + (int)lead;
                L[pos].text = rest;
                n.seq.push_back(parse_map(L[pos].indent));
            } else {
                YamlNode s;
                s.kind = YamlNode::Scalar;
                s.scalar = unquote(trim(rest));
                n.seq.push_back(s);
                ++pos;
            }
        }
        if (pos < L.size() && L[pos].indent > ind)
            throw std::runtime_error("YAML line " + std::to_string(L[pos].lineno) + ": bad indentation in sequence");
        return n;
    }

    YamlNode parse_map(int ind)
    {
        YamlNode n;
        n.kind = YamlNode::Map;
        while (pos < L.size() && L[pos].indent == ind) {
            const std::string& t = L[pos].text;
            if (t.rfind("- ", 0) == 0) break;
            const size_t c = find_colon(t);
            if (c == std::string::npos)
                throw std::runtime_error("YAML line " + std::to_string(L[pos].lineno) + ": expected `key: value`");
            const std::string key = unquote(trim(t.substr(0, c)));
            const std::string val = trim(t.substr(c + 1));
            const int lineno = L[pos].lineno;
            ++pos;
            YamlNode child;
            if (val.empty()) {
                if (pos < L.size() && (L[pos].indent > ind ||
                                       (L[pos].indent == ind && L[pos].text.rfind("- ", 0) == 0)))
                    child = parse_block(L[pos].indent);
            } else {
                child.kind = YamlNode::Scalar;
                child.scalar = unquote(val);
            }
            for (auto& kv : n.map)
                if (kv.first == key)
                    throw std::runtime_error("YAML line " + std::to_string(lineno) + ": duplicate key `" + key + "`");
            n.map.emplace_back(key, std::move(child));
        }
        if (pos < L.size() && L[pos].indent > ind)
            throw std::runtime_error("YAML line " + std::to_string(L[pos].lineno) + ": bad indentation");
        return n;
    }
};

}  // namespace yaml_detail

inline YamlNode yaml_parse(const std::string& text)
{
    yaml_detail::Parser p;
    p.L = yaml_detail::split_lines(text);
    if (p.L.empty()) return YamlNode{};
    YamlNode root = p.parse_block(p.L[0].indent);
    if (p.pos != p.L.size())
        throw std::runtime_error("YAML line " + std::to_string(p.L[p.pos].lineno) + ": unexpected content");
    return root;
}

inline std::string read_text_file(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open `" + path + "`");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// mola-yaml preprocessing: `$(mola-dir PKG)` -> mola_dir, then `$include{PATH}`
// scalars replaced by the parsed content of PATH (relative paths: next to `base_dir`).
inline void yaml_resolve_includes(YamlNode& n, const std::string& base_dir, const std::string& mola_dir, int depth = 0)
{
    if (depth > 8) throw std::runtime_error("$include{} nesting too deep");
    if (n.kind == YamlNode::Scalar) {
        std::string s = n.scalar;
        for (;;) {
            const size_t a = s.find("$(mola-dir");
            if (a == std::string::npos) break;
            const size_t b = s.find(')', a);
            if (b == std::string::npos) throw std::runtime_error("unterminated $(mola-dir ...)");
            if (mola_dir.empty()) throw std::runtime_error("`$(mola-dir ...)` used but no mola_dir was given");
            s = s.substr(0, a) + mola_dir + s.substr(b + 1);
        }
        if (s.rfind("$include{", 0) == 0 && s.back() == '}') {
            std::string path = yaml_detail::trim(s.substr(9, s.size() - 10));
            if (!path.empty() && path[0] != '/' && !base_dir.empty()) path = base_dir + "/" + path;
            YamlNode inc = yaml_parse(read_text_file(path));
            const size_t slash = path.find_last_of('/');
            yaml_resolve_includes(inc, slash == std::string::npos ? std::string(".") : path.substr(0, slash), mola_dir,
                                  depth + 1);
            n = std::move(inc);
        } else {
            n.scalar = s;
        }
        return;
    }
    for (auto& kv : n.map) yaml_resolve_includes(kv.second, base_dir, mola_dir, depth);
    for (auto& e : n.seq) yaml_resolve_includes(e, base_dir, mola_dir, depth);
}

}  // namespace mola_icp_amd
