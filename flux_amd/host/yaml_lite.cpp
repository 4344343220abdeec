// yaml_lite.cpp -- see yaml_lite.hpp.
#include "yaml_lite.hpp"

#include <fstream>
#include <sstream>

namespace yaml_lite {
namespace {

struct Line {
    int indent;
    std::string text;  // without indentation, comments and trailing blanks
    int number;
};

std::string rstrip(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}
std::string lstrip(const std::string &s) {
    size_t i = 0;
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) i++;
    return s.substr(i);
}
std::string strip(const std::string &s) { return rstrip(lstrip(s)); }

// remove a trailing comment (a '#' at line start or preceded by whitespace, outside quotes)
std::string strip_comment(const std::string &s) {
    bool sq = false, dq = false;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        if (c == '"' && !sq) dq = !dq;
        if (c == '#' && !sq && !dq && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;
    std::map<std::string, Node> anchors;

    [[noreturn]] void fail(const std::string &msg, int line) const {
        std::ostringstream o;
        o << "yaml: " << msg << " (line " << line << ")";
        throw Error(o.str());
    }

    static std::string unquote(const std::string &s) {
        if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
            return s.substr(1, s.size() - 2);
        return s;
    }

    Node scalar_or_flow(const std::string &raw, int line) {
        std::string s = strip(raw);
        Node n;
        if (s.empty() || s == "~" || s == "null") return n;
        if (s[0] == '*') {
            auto it = anchors.find(s.substr(1));
            if (it == anchors.end()) fail("unknown alias " + s, line);
            return it->second;
        }
        if (s[0] == '[') {
            if (s.back() != ']') fail("unterminated flow sequence", line);
            n.kind = Node::Seq;
            std::string inner = s.substr(1, s.size() - 2);
            int depth = 0;
            std::string cur;
            auto flush = [&]() {
                if (!strip(cur).empty()) n.seq.push_back(scalar_or_flow(cur, line));
                cur.clear();
            };
            for (char c : inner) {
                if (c == '[') depth++;
                if (c == ']') depth--;
                if (c == ',' && depth == 0)
                    flush();
                else
                    cur.push_back(c);
            }
            flush();
            return n;
        }
        if (s[0] == '{') fail("flow mappings are not supported", line);
        n.kind = Node::Scalar;
        n.scalar = unquote(s);
        return n;
    }

    // splits "key: value" at the first ": " (or trailing ':') outside brackets/quotes
    static bool split_key(const std::string &s, std::string &key, std::string &rest) {
        int depth = 0;
        bool sq = false, dq = false;
        for (size_t i = 0; i < s.size(); i++) {
            char c = s[i];
            if (c == '\'' && !dq) sq = !sq;
            if (c == '"' && !sq) dq = !dq;
            if (sq || dq) continue;
            if (c == '[' || c == '{') depth++;
            if (c == ']' || c == '}') depth--;
            if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) {
                key = unquote(strip(s.substr(0, i)));
                rest = strip(s.substr(i + 1));
                return true;
            }
        }
        return false;
    }

    Node parse_block(int indent) {
        if (pos >= lines.size() || lines[pos].indent < indent) return Node();
        const int my = lines[pos].indent;
        if (lines[pos].text.rfind("- ", 0) == 0 || lines[pos].text == "-") return parse_seq(my);
        std::string k, r;
        if (split_key(lines[pos].text, k, r)) return parse_map(my);
        Node n = scalar_or_flow(lines[pos].text, lines[pos].number);
        pos++;
        return n;
    }

    Node parse_seq(int indent) {
        Node n;
        n.kind = Node::Seq;
        while (pos < lines.size() && lines[pos].indent == indent &&
               (lines[pos].text.rfind("- ", 0) == 0 || lines[pos].text == "-")) {
            Line &L = lines[pos];
            std::string rest = L.text.size() > 1 ? L.text.substr(2) : std::string();
            const std::string body = lstrip(rest);
            if (body.empty()) {
                pos++;
                n.seq.push_back(parse_block(indent + 1));
                continue;
            }
            std::string k, r;
            if (body[0] != '[' && split_key(body, k, r)) {
                // "- key: ..." starts an inline mapping whose column is that of `key`
                const int col = indent + 2 + (int)(rest.size() - body.size());
                L.indent = col;
                L.text = body;
                n.seq.push_back(parse_map(col));
            } else {
                n.seq.push_back(scalar_or_flow(body, L.number));
                pos++;
            }
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail("bad indentation in sequence", lines[pos].number);
        return n;
    }

    Node parse_map(int indent) {
        Node n;
        n.kind = Node::Map;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line L = lines[pos];
            if (L.text.rfind("- ", 0) == 0) break;
            std::string key, rest;
            if (!split_key(L.text, key, rest)) fail("expected `key: value`", L.number);
            pos++;
            std::string anchor;
            if (!rest.empty() && rest[0] == '&') {
                size_t sp = rest.find(' ');
                anchor = rest.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
                rest = sp == std::string::npos ? std::string() : strip(rest.substr(sp + 1));
            }
            Node v;
            if (rest.empty()) {
                // nested block; a sequence may sit at the same indentation as its key
                if (pos < lines.size() &&
                    (lines[pos].indent > indent ||
                     (lines[pos].indent == indent && lines[pos].text.rfind("- ", 0) == 0)))
                    v = parse_block(lines[pos].indent);
            } else {
                v = scalar_or_flow(rest, L.number);
            }
            if (!anchor.empty()) anchors[anchor] = v;
            n.map.emplace_back(key, std::move(v));
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail("bad indentation in mapping", lines[pos].number);
        return n;
    }
};

}  // namespace

Node parse(const std::string &text) {
    Parser p;
    std::istringstream in(text);
    std::string raw;
    int number = 0;
    while (std::getline(in, raw)) {
        number++;
        std::string s = rstrip(strip_comment(raw));
        if (strip(s).empty()) continue;
        if (s == "---") continue;
        int indent = 0;
        while (indent < (int)s.size() && s[indent] == ' ') indent++;
        if (indent < (int)s.size() && s[indent] == '\t') throw Error("yaml: tab indentation is not allowed");
        p.lines.push_back({indent, s.substr(indent), number});
    }
    if (p.lines.empty()) return Node();
    Node root = p.parse_block(0);
    if (p.pos != p.lines.size()) p.fail("unexpected content", p.lines[p.pos].number);
    return root;
}

Node parse_file(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw Error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse(ss.str());
}

}  // namespace yaml_lite
