// cbor.cpp -- see cbor.hpp.
#include "cbor.hpp"

#include <cmath>
#include <cstring>

namespace flux_host {
namespace cbor {

// ---- half precision -----------------------------------------------------------------------------
float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu;
    const uint32_t man = h & 0x3ffu;
    float out;
    if (exp == 0) {  // zero / subnormal: man * 2^-24
        out = std::ldexp((float)man, -24);
    } else if (exp == 31) {
        out = man ? std::nanf("") : INFINITY;
    } else {
        out = std::ldexp((float)(man | 0x400u), (int)exp - 25);
    }
    uint32_t bits;
    std::memcpy(&bits, &out, 4);
    bits |= sign;
    std::memcpy(&out, &bits, 4);
    return out;
}

// round-to-nearest-even conversion; exact = the f16 value equals f
uint16_t f32_to_f16_bits(float f, bool &exact) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const int32_t exp = (int32_t)((x >> 23) & 0xffu) - 127;
    uint32_t man = x & 0x7fffffu;
    uint16_t h;
    if (((x >> 23) & 0xffu) == 0xffu) {  // inf / nan
        h = (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0u));
    } else if (exp > 15) {
        h = (uint16_t)(sign | 0x7c00u);  // overflow -> inf
    } else if (exp >= -14) {  // normal half
        uint32_t m = man >> 13;
        const uint32_t rest = man & 0x1fffu;
        uint32_t e = (uint32_t)(exp + 15);
        if (rest > 0x1000u || (rest == 0x1000u && (m & 1u))) {
            if (++m == 0x400u) {
                m = 0;
                ++e;
            }
        }
        h = e >= 31 ? (uint16_t)(sign | 0x7c00u) : (uint16_t)(sign | (e << 10) | m);
    } else if (exp >= -25) {  // subnormal half
        man |= 0x800000u;
        const int shift = -exp - 14 + 13;  // 14..24
        uint32_t m = man >> shift;
        const uint32_t rest = man & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1);
        if (rest > halfway || (rest == halfway && (m & 1u))) ++m;
        h = (uint16_t)(sign | m);
    } else {
        h = sign;  // underflow -> +-0
    }
    const float back = f16_bits_to_f32(h);
    exact = (back == f) && !(std::isnan(f));
    return h;
}

// ---- encoder --------------------------------------------------------------------------------------
void Encoder::head(int major, uint64_t v) {
    const unsigned char m = (unsigned char)(major << 5);
    if (v < 24) {
        out.push_back((char)(m | v));
    } else if (v <= 0xffu) {
        out.push_back((char)(m | 24));
        out.push_back((char)v);
    } else if (v <= 0xffffu) {
        out.push_back((char)(m | 25));
        out.push_back((char)(v >> 8));
        out.push_back((char)v);
    } else if (v <= 0xffffffffull) {
        out.push_back((char)(m | 26));
        for (int s = 24; s >= 0; s -= 8) out.push_back((char)(v >> s));
    } else {
        out.push_back((char)(m | 27));
        for (int s = 56; s >= 0; s -= 8) out.push_back((char)(v >> s));
    }
}

void Encoder::real(double v) {
    if (std::isnan(v)) {
        out += std::string("\xf9\x7e\x00", 3);
        return;
    }
    if (std::isinf(v)) {
        out += v > 0 ? std::string("\xf9\x7c\x00", 3) : std::string("\xf9\xfc\x00", 3);
        return;
    }
    const float f = (float)v;
    if ((double)f == v) {
        bool exact = false;
        const uint16_t h = f32_to_f16_bits(f, exact);
        if (exact) {
            out.push_back((char)0xf9);
            out.push_back((char)(h >> 8));
            out.push_back((char)h);
            return;
        }
        uint32_t bits;
        std::memcpy(&bits, &f, 4);
        out.push_back((char)0xfa);
        for (int s = 24; s >= 0; s -= 8) out.push_back((char)(bits >> s));
        return;
    }
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    out.push_back((char)0xfb);
    for (int s = 56; s >= 0; s -= 8) out.push_back((char)(bits >> s));
}

// ---- decoder --------------------------------------------------------------------------------------
bool StringReader::read(void *dst, size_t n) {
    if (pos + n > s.size()) return false;
    std::memcpy(dst, s.data() + pos, n);
    pos += n;
    return true;
}

bool Decoder::fail(const char *what) {
    if (err_.empty()) err_ = what;
    return false;
}

// Buffers the next head (skipping tags).  Returns false on EOF (eof_ set if it was clean) or error.
bool Decoder::fill() {
    if (have_) return true;
    if (failed()) return false;
    for (;;) {
        unsigned char b;
        if (!r_.read(&b, 1)) {
            eof_ = true;
            return false;
        }
        major_ = b >> 5;
        info_ = b & 31;
        val_ = 0;
        if (info_ < 24) {
            val_ = (uint64_t)info_;
        } else if (info_ <= 27) {
            const int nb = 1 << (info_ - 24);
            unsigned char buf[8];
            if (!r_.read(buf, (size_t)nb)) return fail("truncated item head");
            for (int k = 0; k < nb; k++) val_ = (val_ << 8) | buf[k];
        } else if (info_ == 31) {
            if (major_ == 0 || major_ == 1 || major_ == 6) return fail("invalid indefinite-length head");
        } else {
            return fail("reserved additional-information value");
        }
        if (major_ == 6) continue;  // tag: ignore, decode the tagged item
        have_ = true;
        return true;
    }
}

Type Decoder::peek() {
    if (!fill()) return failed() ? Type::Error : Type::End;
    switch (major_) {
        case 0: return Type::UInt;
        case 1: return Type::NInt;
        case 2: return Type::Bytes;
        case 3: return Type::Text;
        case 4: return Type::Array;
        case 5: return Type::Map;
        default: break;
    }
    if (info_ == 20 || info_ == 21) return Type::Bool;
    if (info_ == 22 || info_ == 23) return Type::Null;  // null / undefined
    if (info_ >= 25 && info_ <= 27) return Type::Float;
    if (info_ == 31) return Type::Break;
    fail("unsupported simple value");
    return Type::Error;
}

bool Decoder::read_uint(uint64_t &v) {
    if (peek() != Type::UInt) return fail("expected an unsigned integer");
    v = val_;
    have_ = false;
    return true;
}

bool Decoder::read_int(int64_t &v) {
    const Type t = peek();
    if (t == Type::UInt) {
        v = (int64_t)val_;
    } else if (t == Type::NInt) {
        v = -1 - (int64_t)val_;
    } else {
        return fail("expected an integer");
    }
    have_ = false;
    return true;
}

bool Decoder::read_number(double &v) {
    const Type t = peek();
    if (t == Type::UInt) {
        v = (double)val_;
    } else if (t == Type::NInt) {
        v = -1.0 - (double)val_;
    } else if (t == Type::Float) {
        if (info_ == 25) {
            v = (double)f16_bits_to_f32((uint16_t)val_);
        } else if (info_ == 26) {
            const uint32_t bits = (uint32_t)val_;
            float f;
            std::memcpy(&f, &bits, 4);
            v = (double)f;
        } else {
            std::memcpy(&v, &val_, 8);
        }
    } else {
        return fail("expected a number");
    }
    have_ = false;
    return true;
}

bool Decoder::read_bool(bool &v) {
    if (peek() != Type::Bool) return fail("expected a bool");
    v = info_ == 21;
    have_ = false;
    return true;
}

bool Decoder::read_null() {
    if (peek() != Type::Null) return fail("expected null");
    have_ = false;
    return true;
}

bool Decoder::read_string(int major, std::string &s) {
    s.clear();
    if (!fill() || major_ != major) return fail(major == 3 ? "expected a text string" : "expected a byte string");
    if (info_ != 31) {
        const uint64_t n = val_;
        have_ = false;
        // the head is peer-controlled: nothing is allocated beyond kMaxString (the protocol's strings are scene names and
        // enum tags; a 9-byte head must not be able to demand gigabytes)
        if (n > kMaxString) return fail("string too long");
        s.resize((size_t)n);
        if (n && !r_.read(&s[0], (size_t)n)) return fail("truncated string");
        return true;
    }
    have_ = false;  // indefinite: definite chunks of the same major type until break
    for (;;) {
        if (!fill()) return fail("truncated chunked string");
        if (major_ == 7 && info_ == 31) {
            have_ = false;
            return true;
        }
        if (major_ != major || info_ == 31) return fail("bad chunk in indefinite string");
        const uint64_t n = val_;
        have_ = false;
        if (n > kMaxString || s.size() + n > kMaxString) return fail("string too long");  // total over all chunks
        const size_t at = s.size();
        s.resize(at + (size_t)n);
        if (n && !r_.read(&s[at], (size_t)n)) return fail("truncated string");
    }
}

bool Decoder::read_text(std::string &s) { return read_string(3, s); }
bool Decoder::read_bytes(std::string &s) { return read_string(2, s); }

bool Decoder::read_array(uint64_t &n) {
    if (peek() != Type::Array) return fail("expected an array");
    n = info_ == 31 ? kIndefinite : val_;
    have_ = false;
    return true;
}

bool Decoder::read_map(uint64_t &n) {
    if (peek() != Type::Map) return fail("expected a map");
    n = info_ == 31 ? kIndefinite : val_;
    have_ = false;
    return true;
}

bool Decoder::at_break() {
    if (peek() == Type::Break) {
        have_ = false;
        return true;
    }
    return false;
}

bool Decoder::skip() {
    // nesting is peer-controlled: bounded like serde_cbor's recursion limit (128) instead of the C++ stack
    struct Depth {
        int &d;
        explicit Depth(int &x) : d(x) { ++d; }
        ~Depth() { --d; }
    } guard(depth_);
    if (depth_ > kMaxDepth) return fail("nesting too deep");
    const Type t = peek();
    uint64_t n;
    std::string s;
    switch (t) {
        case Type::UInt:
        case Type::NInt:
        case Type::Bool:
        case Type::Null:
        case Type::Float:
            have_ = false;
            return true;
        case Type::Bytes: return read_bytes(s);
        case Type::Text: return read_text(s);
        case Type::Array:
            if (!read_array(n)) return false;
            for (uint64_t k = 0; n == kIndefinite ? !at_break() : k < n; k++)
                if (failed() || !skip()) return false;
            return !failed();
        case Type::Map:
            if (!read_map(n)) return false;
            for (uint64_t k = 0; n == kIndefinite ? !at_break() : k < n; k++)
                if (failed() || !skip() || !skip()) return false;
            return !failed();
        case Type::Break: return fail("unexpected break");
        case Type::End: return fail("unexpected end of stream");
        default: return false;
    }
}

}  // namespace cbor
}  // namespace flux_host
