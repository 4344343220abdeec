// cbor.hpp -- the subset of CBOR (RFC 8949) the reference's node protocol uses, written to match what
// serde_cbor 0.9.0 (Cargo.lock:580-587) puts on the wire and tolerant in what it accepts.
//
// serde_cbor is a third-party dependency absent from /root/reference; its encoding is restated here from
// its published behaviour (wire format UNVERIFIED against the real crate in this environment):
//   * integers: shortest form (major 0/1); text strings major 3; arrays/maps with definite lengths;
//   * structs -> maps keyed by field name; tuple structs -> arrays; Option::None -> null (0xf6);
//   * floats: f64 is written as the SHORTEST of f16/f32/f64 that holds the value exactly (the crate depends
//     on `half` for this, Cargo.lock:585); +-inf and NaN as f16;
//   * enums in the pre-0.10 ("legacy") layout: unit variant -> "Name"; newtype variant -> ["Name", value];
//     tuple variant -> ["Name", v0, v1, ..]; struct variant -> ["Name", {fields}].
// The decoder additionally accepts the 0.10+ enum layout ({"Name": value}), indefinite-length strings /
// arrays / maps, any float width, integers where floats are expected, and skips tags and unknown map keys.
// The codec itself is pinned by the RFC 8949 Appendix A examples (flux_host_test.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace flux_host {
namespace cbor {

// ---- encoding ---------------------------------------------------------------------------------
class Encoder {
public:
    std::string out;
    void uint(uint64_t v) { head(0, v); }
    void nint(uint64_t minus_one_minus) { head(1, minus_one_minus); }  // encodes -1 - n
    void integer(int64_t v) { v >= 0 ? head(0, (uint64_t)v) : head(1, (uint64_t)(-1 - v)); }
    void text(const std::string &s) { head(3, s.size()); out += s; }
    void bytes(const std::string &s) { head(2, s.size()); out += s; }
    void array(uint64_t n) { head(4, n); }
    void map(uint64_t n) { head(5, n); }
    void boolean(bool b) { out.push_back((char)(b ? 0xf5 : 0xf4)); }
    void null() { out.push_back((char)0xf6); }
    void real(double v);  // shortest exact float (serde_cbor 0.9 serialize_f64)
    void key(const char *k) { text(k); }
private:
    void head(int major, uint64_t v);
};

// ---- decoding: a pull parser over a blocking byte source ------------------------------------------
struct Reader {
    virtual ~Reader() = default;
    virtual bool read(void *dst, size_t n) = 0;  // all n bytes or false (EOF / error)
};

struct StringReader : Reader {
    const std::string &s;
    size_t pos = 0;
    explicit StringReader(const std::string &str) : s(str) {}
    bool read(void *dst, size_t n) override;
    bool at_end() const { return pos >= s.size(); }
};

enum class Type { UInt, NInt, Bytes, Text, Array, Map, Bool, Null, Float, Break, End, Error };

class Decoder {
public:
    explicit Decoder(Reader &r) : r_(r) {}
    // Type of the next item without consuming it (tags are skipped).  End = clean EOF before any byte.
    Type peek();
    bool read_uint(uint64_t &v);          // UInt only
    bool read_int(int64_t &v);            // UInt or NInt
    bool read_number(double &v);          // Float of any width, or an integer
    bool read_bool(bool &v);
    bool read_null();
    bool read_text(std::string &s);       // definite or indefinite (chunked) text
    bool read_bytes(std::string &s);
    // Container headers: n = element (array) / pair (map) count, or kIndefinite; then read the elements and,
    // for kIndefinite, call at_break() before each element (it consumes the 0xff when it returns true).
    static constexpr uint64_t kIndefinite = ~0ull;
    static constexpr uint64_t kMaxString = 1ull << 24;  // bytes of one (possibly chunked) string accepted from a peer
    static constexpr int kMaxDepth = 128;               // container nesting accepted by skip() (serde_cbor's limit)
    bool read_array(uint64_t &n);
    bool read_map(uint64_t &n);
    bool at_break();
    bool skip();                          // skips one whole item of any type
    const std::string &error() const { return err_; }
    bool failed() const { return !err_.empty(); }
private:
    bool fill();
    bool fail(const char *what);
    bool read_string(int major, std::string &s);
    Reader &r_;
    bool have_ = false, eof_ = false;
    int major_ = 0, info_ = 0;
    uint64_t val_ = 0;  // argument of the buffered head
    int depth_ = 0;     // current nesting inside skip()
    std::string err_;
};

// half-precision helpers (exposed for the tests)
uint16_t f32_to_f16_bits(float f, bool &exact);
float f16_bits_to_f32(uint16_t h);

}  // namespace cbor
}  // namespace flux_host
