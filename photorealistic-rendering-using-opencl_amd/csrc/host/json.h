// json.h -- a small recursive-descent JSON reader (objects, arrays, numbers, strings, bools,
// null).  The reference parses its scene files with rapidjson (include/Scene/scene.h:10-11);
// only the accessor subset host_scene::load needs is provided here.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace prt {
namespace json {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool is_integer = false;     // literal had no '.', 'e' or 'E' (rapidjson IsInt analogue)
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool IsObject() const { return kind == Object; }
    bool IsArray() const { return kind == Array; }
    bool IsNumber() const { return kind == Number; }
    bool IsInt() const { return kind == Number && is_integer; }
    bool IsString() const { return kind == String; }
    bool HasMember(const char* k) const { return find(k) != nullptr; }
    const Value* find(const char* k) const {
        if (kind != Object) return nullptr;
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    const Value& operator[](const char* k) const {
        const Value* v = find(k);
        if (!v) throw std::runtime_error(std::string("json: missing member '") + k + "'");
        return *v;
    }
    const Value& operator[](size_t i) const {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    size_t Size() const { return arr.size(); }
    // rapidjson GetFloat(): double -> float conversion of the parsed number
    float GetFloat() const { return (float)num; }
    int GetInt() const { return (int)num; }
    const std::string& GetString() const { return str; }
};

class Parser {
public:
    explicit Parser(const std::string& text) : s(text), p(0) {}
    Value parse() {
        Value v = value();
        ws();
        if (p != s.size()) fail("trailing characters");
        return v;
    }
private:
    const std::string& s;
    size_t p;
    int depth = 0;                                 // nesting of objects / arrays: bounded, the parser recurses
    struct Nest { int& d; explicit Nest(int& dd) : d(dd) { ++d; } ~Nest() { --d; } };
    [[noreturn]] void fail(const char* msg) {
        throw std::runtime_error(std::string("json: ") + msg + " at offset " + std::to_string(p));
    }
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
    Value value() {
        ws();
        if (p >= s.size()) fail("unexpected end");
        char c = s[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { Value v; v.kind = Value::String; v.str = string(); return v; }
        if (c == 't' && s.compare(p, 4, "true") == 0) { p += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (c == 'f' && s.compare(p, 5, "false") == 0) { p += 5; Value v; v.kind = Value::Bool; return v; }
        if (c == 'n' && s.compare(p, 4, "null") == 0) { p += 4; return Value(); }
        return number();
    }
    Value object() {
        Nest nest(depth);
        if (depth > 64) fail("nesting deeper than 64 levels");
        Value v; v.kind = Value::Object;
        ++p; ws();
        if (p < s.size() && s[p] == '}') { ++p; return v; }
        for (;;) {
            ws();
            if (p >= s.size() || s[p] != '"') fail("expected member name");
            std::string k = string();
            ws();
            if (p >= s.size() || s[p] != ':') fail("expected ':'");
            ++p;
            v.obj.emplace_back(k, value());
            ws();
            if (p < s.size() && s[p] == ',') { ++p; continue; }
            if (p < s.size() && s[p] == '}') { ++p; break; }
            fail("expected ',' or '}'");
        }
        return v;
    }
    Value array() {
        Nest nest(depth);
        if (depth > 64) fail("nesting deeper than 64 levels");
        Value v; v.kind = Value::Array;
        ++p; ws();
        if (p < s.size() && s[p] == ']') { ++p; return v; }
        for (;;) {
            v.arr.push_back(value());
            ws();
            if (p < s.size() && s[p] == ',') { ++p; continue; }
            if (p < s.size() && s[p] == ']') { ++p; break; }
            fail("expected ',' or ']'");
        }
        return v;
    }
    std::string string() {
        std::string out;
        ++p;
        while (p < s.size() && s[p] != '"') {
            char c = s[p++];
            if (c == '\\') {
                if (p >= s.size()) fail("bad escape");
                char e = s[p++];
                switch (e) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': p += 4; out += '?'; break;   // non-ASCII never occurs in scene files
                    default: out += e;
                }
            } else out += c;
        }
        if (p >= s.size()) fail("unterminated string");
        ++p;
        return out;
    }
    Value number() {
        size_t b = p;
        bool integer = true;
        if (p < s.size() && (s[p] == '-' || s[p] == '+')) ++p;
        while (p < s.size() && ((s[p] >= '0' && s[p] <= '9') || s[p] == '.' || s[p] == 'e' || s[p] == 'E' || s[p] == '-' || s[p] == '+')) {
            if (s[p] == '.' || s[p] == 'e' || s[p] == 'E') integer = false;
            ++p;
        }
        if (p == b) fail("unexpected character");
        Value v; v.kind = Value::Number;
        v.num = strtod(s.substr(b, p - b).c_str(), nullptr);   // correctly rounded, like rapidjson's full-precision path for these short literals
        v.is_integer = integer;
        return v;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace prt
