// Minimal JSON + UBJSON DOM for the calibration CLI (host side, no dependencies).
//
// Plays the role nlohmann::json (vendored by the reference as
// include/OpenCameraCalibrator/utils/json.h) plays in src/io/*.cc: parse the input
// files, write the result file.  Objects are std::map (sorted keys), like
// nlohmann's default object type, so iteration order and the key order of the
// output file match the reference's.
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace oicc_json {

class Value {
 public:
  enum Type { Null, Bool, Number, String, Array, Object };
  Type type = Null;
  bool b = false;
  double num = 0.0;
  bool is_int = false;
  int64_t inum = 0;
  std::string str;
  std::vector<Value> arr;
  std::map<std::string, Value> obj;

  Value() {}
  Value(double v) : type(Number), num(v) {}                 // NOLINT
  Value(int64_t v) : type(Number), num(double(v)), is_int(true), inum(v) {}  // NOLINT
  Value(const std::string& s) : type(String), str(s) {}     // NOLINT
  Value(const char* s) : type(String), str(s) {}            // NOLINT
  Value(bool v) : type(Bool), b(v) {}                       // NOLINT

  bool is_null() const { return type == Null; }
  bool is_object() const { return type == Object; }
  bool is_array() const { return type == Array; }
  bool contains(const std::string& k) const { return type == Object && obj.count(k) > 0; }
  size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }

  Value& operator[](const std::string& k) {
    if (type == Null) type = Object;
    if (type != Object) throw std::runtime_error("json: not an object (key " + k + ")");
    return obj[k];
  }
  const Value& at(const std::string& k) const {
    if (type != Object) throw std::runtime_error("json: not an object (key " + k + ")");
    auto it = obj.find(k);
    if (it == obj.end()) throw std::runtime_error("json: missing key " + k);
    return it->second;
  }
  const Value& at(size_t i) const {
    if (type != Array || i >= arr.size()) throw std::runtime_error("json: bad array index");
    return arr[i];
  }
  double as_double() const {
    if (type == Number) return num;
    if (type == String) return std::stod(str);
    throw std::runtime_error("json: not a number");
  }
  int64_t as_int() const { return type == Number ? (is_int ? inum : int64_t(num)) : throw std::runtime_error("json: not a number"); }
  const std::string& as_string() const { if (type != String) throw std::runtime_error("json: not a string"); return str; }
  void push_back(const Value& v) { if (type == Null) type = Array; arr.push_back(v); }
};

// ------------------------------------------------------------------ JSON text
class Parser {
  const char* p; const char* e;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json parse error: ") + m); }
  std::string parse_string() {
    if (*p != '"') fail("expected string");
    ++p;
    std::string s;
    while (p < e && *p != '"') {
      if (*p == '\\') {
        ++p; if (p >= e) fail("bad escape");
        switch (*p) {
          case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
          case 'b': s += '\b'; break; case 'f': s += '\f'; break;
          case 'u': {
            auto hex4 = [&](const char* q) -> unsigned { if (e - q < 4) fail("bad \\u"); return unsigned(std::stoul(std::string(q, q + 4), nullptr, 16)); };
            unsigned cp = hex4(p + 1); p += 4;
            if (cp >= 0xD800 && cp < 0xDC00 && e - p > 6 && p[1] == '\\' && p[2] == 'u') {   // surrogate pair
              const unsigned lo = hex4(p + 3);
              if (lo >= 0xDC00 && lo < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); p += 6; }
            }
            if (cp < 0x80) s += char(cp);
            else if (cp < 0x800) { s += char(0xC0 | (cp >> 6)); s += char(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { s += char(0xE0 | (cp >> 12)); s += char(0x80 | ((cp >> 6) & 0x3F)); s += char(0x80 | (cp & 0x3F)); }
            else { s += char(0xF0 | (cp >> 18)); s += char(0x80 | ((cp >> 12) & 0x3F)); s += char(0x80 | ((cp >> 6) & 0x3F)); s += char(0x80 | (cp & 0x3F)); }
            break; }
          default: s += *p;
        }
        ++p;
      } else s += *p++;
    }
    if (p >= e) fail("unterminated string");
    ++p;
    return s;
  }
  int depth = 0;
  struct Nest { int& d; explicit Nest(int& x) : d(x) { ++d; } ~Nest() { --d; } };
  Value parse_value() {
    ws(); if (p >= e) fail("unexpected end");
    const Nest nest(depth); if (depth > 256) fail("nesting too deep");
    Value v;
    if (*p == '{') {
      ++p; v.type = Value::Object; ws();
      if (*p == '}') { ++p; return v; }
      while (true) { ws(); std::string k = parse_string(); ws(); if (*p != ':') fail("expected ':'"); ++p; v.obj[k] = parse_value(); ws();
        if (*p == ',') { ++p; continue; } if (*p == '}') { ++p; break; } fail("expected ',' or '}'"); }
    } else if (*p == '[') {
      ++p; v.type = Value::Array; ws();
      if (*p == ']') { ++p; return v; }
      while (true) { v.arr.push_back(parse_value()); ws(); if (*p == ',') { ++p; continue; } if (*p == ']') { ++p; break; } fail("expected ',' or ']'"); }
    } else if (*p == '"') { v.type = Value::String; v.str = parse_string(); }
    else if (!strncmp(p, "true", 4)) { v = Value(true); p += 4; }
    else if (!strncmp(p, "false", 5)) { v = Value(false); p += 5; }
    else if (!strncmp(p, "null", 4)) { p += 4; }
    else {
      const char* s = p; bool isint = true;
      if (*p == '-') ++p;
      while (p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) { if (*p == '.' || *p == 'e' || *p == 'E') isint = false; ++p; }
      if (p == s) fail("unexpected character");
      std::string t(s, p);
      v.type = Value::Number; v.num = std::strtod(t.c_str(), nullptr);   // (std::stod throws on subnormal results)
      if (isint) {   // every int64 exactly (ns timestamps exceed 2^53); beyond int64 the double stands in
        errno = 0; char* endp = nullptr; const long long ll = std::strtoll(t.c_str(), &endp, 10);
        if (errno == 0 && endp && *endp == '\0') { v.is_int = true; v.inum = ll; }
      }
    }
    return v;
  }
 public:
  static Value parse(const std::string& text) { Parser q; q.p = text.data(); q.e = q.p + text.size(); Value v = q.parse_value(); q.ws(); if (q.p != q.e) q.fail("trailing data"); return v; }
};

inline bool read_file(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return false;
  std::stringstream ss; ss << f.rdbuf(); *out = ss.str(); return true;
}
inline bool parse_file(const std::string& path, Value* out) {
  std::string text; if (!read_file(path, &text)) return false;
  *out = Parser::parse(text); return true;
}

// shortest decimal that round-trips (what nlohmann's dump prints)
inline std::string number_to_string(double v) {
  if (std::isnan(v) || std::isinf(v)) return "null";
  if (v == std::floor(v) && std::fabs(v) < 1e15) { char b[40]; snprintf(b, sizeof(b), "%.1f", v); return b; }
  char b[40];
  for (int prec = 1; prec <= 17; ++prec) { snprintf(b, sizeof(b), "%.*g", prec, v); if (std::strtod(b, nullptr) == v) break; }
  return b;
}
// JSON string escapes as nlohmann's dump writes them (ensure_ascii = false): quote, backslash, control characters
inline void dump_string(const std::string& str, std::ostream& os) {
  os << '"';
  for (unsigned char c : str) {
    switch (c) {
      case '"': os << "\\\""; break; case '\\': os << "\\\\"; break; case '\n': os << "\\n"; break; case '\t': os << "\\t"; break;
      case '\r': os << "\\r"; break; case '\b': os << "\\b"; break; case '\f': os << "\\f"; break;
      default: if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", unsigned(c)); os << b; } else os << char(c);
    }
  }
  os << '"';
}
inline void dump(const Value& v, std::ostream& os, int indent, int level = 0) {
  const std::string pad(size_t(indent * (level + 1)), ' '), padc(size_t(indent * level), ' ');
  switch (v.type) {
    case Value::Null: os << "null"; break;
    case Value::Bool: os << (v.b ? "true" : "false"); break;
    case Value::Number: if (v.is_int) os << v.inum; else os << number_to_string(v.num); break;
    case Value::String: dump_string(v.str, os); break;
    case Value::Array: {
      if (v.arr.empty()) { os << "[]"; break; }
      os << "[\n"; for (size_t i = 0; i < v.arr.size(); ++i) { os << pad; dump(v.arr[i], os, indent, level + 1); os << (i + 1 < v.arr.size() ? ",\n" : "\n"); } os << padc << "]"; break; }
    case Value::Object: {
      if (v.obj.empty()) { os << "{}"; break; }
      os << "{\n"; size_t i = 0; for (const auto& kv : v.obj) { os << pad; dump_string(kv.first, os); os << ": "; dump(kv.second, os, indent, level + 1); os << (++i < v.obj.size() ? ",\n" : "\n"); } os << padc << "}"; break; }
  }
}

// ------------------------------------------------------------------ UBJSON
// Reader for what nlohmann::json::to_ubjson produces (src/core/board_extractor.cc
// writes the corner file with it; src/io/read_scene.cc:25-41 reads it back),
// including the optional '$' type / '#' count container optimisations.
class UbjsonReader {
  const uint8_t* p; const uint8_t* e;
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("ubjson parse error: ") + m); }
  uint8_t get() { if (p >= e) fail("unexpected end"); return *p++; }
  template <class T> T be() { if (size_t(e - p) < sizeof(T)) fail("unexpected end"); typename std::make_unsigned<T>::type u = 0; for (size_t i = 0; i < sizeof(T); ++i) u = (u << 8) | *p++; T v; std::memcpy(&v, &u, sizeof(T)); return v; }
  int64_t read_int(uint8_t t) {
    switch (t) { case 'i': return be<int8_t>(); case 'U': return be<uint8_t>(); case 'I': return be<int16_t>(); case 'l': return be<int32_t>(); case 'L': return be<int64_t>(); default: fail("expected integer type"); }
  }
  std::string read_str_body() { const int64_t n = read_int(get()); if (n < 0 || e - p < n) fail("bad string length"); std::string s(reinterpret_cast<const char*>(p), size_t(n)); p += n; return s; }
  int depth = 0;
  struct Nest { int& d; explicit Nest(int& x) : d(x) { ++d; } ~Nest() { --d; } };
  // a counted container cannot hold more elements than bytes are left, unless its elements carry no payload (Z, T, F): those are capped
  void check_count(int64_t n, uint8_t ct) { if (n < 0) fail("negative count"); const bool empty_type = ct == 'Z' || ct == 'T' || ct == 'F'; if (empty_type ? n > (int64_t(1) << 20) : n > int64_t(e - p)) fail("container count exceeds the file"); }
  Value read_value(uint8_t t) {
    const Nest nest(depth); if (depth > 256) fail("nesting too deep");
    Value v;
    switch (t) {
      case 'Z': return v; case 'T': return Value(true); case 'F': return Value(false);
      case 'i': case 'U': case 'I': case 'l': case 'L': return Value(int64_t(read_int(t)));
      case 'd': { uint32_t u = be<uint32_t>(); float f; std::memcpy(&f, &u, 4); return Value(double(f)); }
      case 'D': { uint64_t u = be<uint64_t>(); double d; std::memcpy(&d, &u, 8); return Value(d); }
      case 'C': return Value(std::string(1, char(get())));
      case 'S': return Value(read_str_body());
      case 'H': return Value(std::stod(read_str_body()));
      case '[': {
        v.type = Value::Array; uint8_t ct = 0; int64_t n = -1;
        if (p < e && *p == '$') { ++p; ct = get(); }
        if (p < e && *p == '#') { ++p; n = read_int(get()); check_count(n, ct); }
        if (n >= 0) { for (int64_t i = 0; i < n; ++i) v.arr.push_back(read_value(ct ? ct : get())); }
        else { while (true) { uint8_t c = get(); if (c == ']') break; v.arr.push_back(read_value(c)); } }
        return v; }
      case '{': {
        v.type = Value::Object; uint8_t ct = 0; int64_t n = -1;
        if (p < e && *p == '$') { ++p; ct = get(); }
        if (p < e && *p == '#') { ++p; n = read_int(get()); check_count(n, ct); }
        if (n >= 0) { for (int64_t i = 0; i < n; ++i) { std::string k = read_str_body(); v.obj[k] = read_value(ct ? ct : get()); } }
        else { while (true) { if (p < e && *p == '}') { ++p; break; } std::string k = read_str_body(); v.obj[k] = read_value(get()); } }
        return v; }
      default: fail("unknown type marker");
    }
  }
 public:
  static Value parse(const std::string& bytes) {
    UbjsonReader r; r.p = reinterpret_cast<const uint8_t*>(bytes.data()); r.e = r.p + bytes.size();
    return r.read_value(r.get());
  }
};

}  // namespace oicc_json
