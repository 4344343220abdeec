// yaml_lite.hpp -- the YAML subset the reference's scene files use (serde_yaml input of
// flux/src/main.rs:27-29): block maps and sequences, flow sequences `[a, b, c]`, plain scalars,
// comments, anchors on map values (`mat1: &mat1`) and aliases (`material: *mat1`).
// No yaml-cpp/libyaml headers exist in this image, hence this small parser.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace yaml_lite {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Node {
    enum Kind { Null, Scalar, Seq, Map } kind = Null;
    std::string scalar;
    std::vector<Node> seq;
    std::vector<std::pair<std::string, Node>> map;  // insertion order kept

    const Node *find(const std::string &key) const {
        for (const auto &kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

Node parse(const std::string &text);
Node parse_file(const std::string &path);

}  // namespace yaml_lite
