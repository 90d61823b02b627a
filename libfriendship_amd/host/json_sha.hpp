// json_sha.hpp -- the two small format pieces the effect-file loader needs: a JSON reader/writer sufficient for
// serde_json's output of `EffectDesc` (reference src/routing/effect.rs:44-74, adjlist.rs:11-15, routegraph.rs:20-44)
// and SHA-256 (the reference identifies effect files by the sha256 of their bytes, src/resman.rs:44-60).
// Self-contained: no third-party JSON or crypto library is available in this environment.
#pragma once

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace friendship {
namespace json {

struct Value;
using Array = std::vector<Value>;
using Object = std::vector<std::pair<std::string, Value>>;   // insertion order kept (serde emits declaration order)

struct Value {
    enum Kind { Null, Bool, Int, Float, String, Arr, Obj } kind = Null;
    bool b = false;
    int64_t i = 0;
    double f = 0;
    std::string s;
    std::shared_ptr<Array> a;
    std::shared_ptr<Object> o;

    static Value null() { return Value{}; }
    static Value integer(int64_t v) { Value x; x.kind = Int; x.i = v; return x; }
    static Value string(std::string v) { Value x; x.kind = String; x.s = std::move(v); return x; }
    static Value array(Array v = {}) { Value x; x.kind = Arr; x.a = std::make_shared<Array>(std::move(v)); return x; }
    static Value object(Object v = {}) { Value x; x.kind = Obj; x.o = std::make_shared<Object>(std::move(v)); return x; }

    bool is_null() const { return kind == Null; }
    const Value &at(const std::string &key) const {
        if (kind != Obj) throw std::runtime_error("json: not an object");
        for (auto &kv : *o) if (kv.first == key) return kv.second;
        throw std::runtime_error("json: missing field `" + key + "`");
    }
    const Array &arr() const {
        if (kind != Arr) throw std::runtime_error("json: not an array");
        return *a;
    }
    const std::string &str() const {
        if (kind != String) throw std::runtime_error("json: not a string");
        return s;
    }
    uint64_t u64() const {
        if (kind != Int || i < 0) throw std::runtime_error("json: not an unsigned integer");
        return (uint64_t)i;
    }
};

class Parser {
    const std::string &t_;
    size_t p_ = 0;
    void ws() { while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) ++p_; }
    [[noreturn]] void fail(const char *m) const { throw std::runtime_error(std::string("json: ") + m + " at byte " + std::to_string(p_)); }
    char peek() { ws(); if (p_ >= t_.size()) fail("unexpected end"); return t_[p_]; }
    void expect(char c) { if (peek() != c) fail("unexpected character"); ++p_; }
    std::string parse_string() {
        expect('"');
        std::string out;
        while (true) {
            if (p_ >= t_.size()) fail("unterminated string");
            char c = t_[p_++];
            if (c == '"') break;
            if (c != '\\') { out.push_back(c); continue; }
            if (p_ >= t_.size()) fail("bad escape");
            char e = t_[p_++];
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                if (p_ + 4 > t_.size()) fail("bad \\u escape");
                unsigned cp = (unsigned)std::stoul(t_.substr(p_, 4), nullptr, 16);
                p_ += 4;
                if (cp < 0x80) out.push_back((char)cp);
                else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                break;
            }
            default: fail("bad escape");
            }
        }
        return out;
    }

public:
    explicit Parser(const std::string &text) : t_(text) {}
    Value parse_value() {
        char c = peek();
        if (c == '{') {
            ++p_;
            Value v = Value::object();
            if (peek() == '}') { ++p_; return v; }
            while (true) {
                ws();
                std::string k = parse_string();
                expect(':');
                v.o->emplace_back(std::move(k), parse_value());
                char d = peek();
                ++p_;
                if (d == '}') break;
                if (d != ',') fail("expected , or }");
            }
            return v;
        }
        if (c == '[') {
            ++p_;
            Value v = Value::array();
            if (peek() == ']') { ++p_; return v; }
            while (true) {
                v.a->push_back(parse_value());
                char d = peek();
                ++p_;
                if (d == ']') break;
                if (d != ',') fail("expected , or ]");
            }
            return v;
        }
        if (c == '"') return Value::string(parse_string());
        if (t_.compare(p_, 4, "null") == 0) { p_ += 4; return Value::null(); }
        if (t_.compare(p_, 4, "true") == 0) { p_ += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (t_.compare(p_, 5, "false") == 0) { p_ += 5; Value v; v.kind = Value::Bool; return v; }
        size_t s = p_;
        bool is_float = false;
        while (p_ < t_.size() && (std::strchr("+-0123456789.eE", t_[p_]) != nullptr)) {
            if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') is_float = true;
            ++p_;
        }
        if (s == p_) fail("unexpected token");
        Value v;
        if (is_float) { v.kind = Value::Float; v.f = std::stod(t_.substr(s, p_ - s)); }
        else { v.kind = Value::Int; v.i = std::stoll(t_.substr(s, p_ - s)); }
        return v;
    }
    Value parse_document() {
        Value v = parse_value();
        ws();
        if (p_ != t_.size()) fail("trailing characters");
        return v;
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse_document(); }

// Compact form, as serde_json::to_writer / to_vec emit it (no whitespace, declaration order).
inline void write(const Value &v, std::string &out) {
    switch (v.kind) {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v.b ? "true" : "false"; break;
    case Value::Int: out += std::to_string(v.i); break;
    case Value::Float: { char buf[40]; std::snprintf(buf, sizeof buf, "%.17g", v.f); out += buf; break; }
    case Value::String:
        out.push_back('"');
        for (unsigned char c : v.s) {
            switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; }
                else out.push_back((char)c);
            }
        }
        out.push_back('"');
        break;
    case Value::Arr:
        out.push_back('[');
        for (size_t i = 0; i < v.a->size(); ++i) { if (i) out.push_back(','); write((*v.a)[i], out); }
        out.push_back(']');
        break;
    case Value::Obj:
        out.push_back('{');
        for (size_t i = 0; i < v.o->size(); ++i) {
            if (i) out.push_back(',');
            write(Value::string((*v.o)[i].first), out);
            out.push_back(':');
            write((*v.o)[i].second, out);
        }
        out.push_back('}');
        break;
    }
}
inline std::string to_string(const Value &v) { std::string s; write(v, s); return s; }

}  // namespace json

// FIPS 180-4 SHA-256.
inline std::array<uint8_t, 32> sha256(const uint8_t *data, size_t len) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    std::vector<uint8_t> msg(data, data + len);
    msg.push_back(0x80);
    while (msg.size() % 64 != 56) msg.push_back(0);
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 7; i >= 0; --i) msg.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < msg.size(); off += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i)
            w[i] = ((uint32_t)msg[off + 4 * i] << 24) | ((uint32_t)msg[off + 4 * i + 1] << 16) | ((uint32_t)msg[off + 4 * i + 2] << 8) | msg[off + 4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    std::array<uint8_t, 32> out;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
    return out;
}
inline std::array<uint8_t, 32> sha256(const std::string &s) { return sha256((const uint8_t *)s.data(), s.size()); }

}  // namespace friendship
