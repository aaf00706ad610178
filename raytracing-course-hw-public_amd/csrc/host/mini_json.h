// mini_json.h — a small recursive-descent JSON reader for the glTF loader (host only, one-off I/O).
// The reference vendors nlohmann/json 3.12.0 (src/json.hpp) for this; only the subset glTF needs is restated:
// objects, arrays, strings (with escapes), numbers (kept as double + integer flag), booleans, null.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mjson {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    bool is_int = false;
    int64_t inum = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj; // insertion order kept

    bool is_null() const { return kind == Null; }
    bool is_array() const { return kind == Array; }
    bool is_object() const { return kind == Object; }
    bool contains(const std::string &k) const {
        if (kind != Object)
            return false;
        for (auto &kv : obj)
            if (kv.first == k)
                return true;
        return false;
    }
    const Value &operator[](const std::string &k) const {
        static const Value null_value;
        if (kind != Object)
            return null_value;
        for (auto &kv : obj)
            if (kv.first == k)
                return kv.second;
        return null_value;
    }
    const Value &operator[](size_t i) const {
        if (kind != Array || i >= arr.size())
            throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
    // nlohmann-style conversions: integers convert exactly, floats via double
    float as_float() const {
        if (kind != Number)
            throw std::runtime_error("json: number expected");
        return is_int ? static_cast<float>(inum) : static_cast<float>(num);
    }
    int64_t as_int() const {
        if (kind != Number)
            throw std::runtime_error("json: number expected");
        return is_int ? inum : static_cast<int64_t>(num);
    }
    const std::string &as_string() const {
        if (kind != String)
            throw std::runtime_error("json: string expected");
        return str;
    }
};

class Parser {
  public:
    explicit Parser(const std::string &text) : s(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (p != s.size())
            fail("trailing characters");
        return v;
    }

  private:
    const std::string &s;
    size_t p = 0;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p)); }
    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r'))
            ++p;
    }
    Value value() {
        ws();
        if (p >= s.size())
            fail("unexpected end");
        char c = s[p];
        if (c == '{')
            return object();
        if (c == '[')
            return array();
        if (c == '"') {
            Value v;
            v.kind = Value::String;
            v.str = string();
            return v;
        }
        if (c == 't' || c == 'f' || c == 'n')
            return literal();
        return number();
    }
    Value literal() {
        Value v;
        if (s.compare(p, 4, "true") == 0) {
            v.kind = Value::Bool;
            v.b = true;
            p += 4;
        } else if (s.compare(p, 5, "false") == 0) {
            v.kind = Value::Bool;
            p += 5;
        } else if (s.compare(p, 4, "null") == 0) {
            p += 4;
        } else
            fail("bad literal");
        return v;
    }
    Value number() {
        size_t b = p;
        bool integral = true;
        if (p < s.size() && s[p] == '-')
            ++p;
        while (p < s.size() && ((s[p] >= '0' && s[p] <= '9') || s[p] == '.' || s[p] == 'e' || s[p] == 'E' || s[p] == '+' || s[p] == '-')) {
            if (s[p] == '.' || s[p] == 'e' || s[p] == 'E')
                integral = false;
            ++p;
        }
        if (b == p)
            fail("bad number");
        std::string tok = s.substr(b, p - b);
        Value v;
        v.kind = Value::Number;
        v.num = std::strtod(tok.c_str(), nullptr);
        if (integral && tok.size() < 19) {
            v.is_int = true;
            v.inum = std::strtoll(tok.c_str(), nullptr, 10);
        }
        return v;
    }
    static void append_utf8(std::string &out, uint32_t cp) {
        if (cp < 0x80)
            out += char(cp);
        else if (cp < 0x800) {
            out += char(0xC0 | (cp >> 6));
            out += char(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += char(0xE0 | (cp >> 12));
            out += char(0x80 | ((cp >> 6) & 0x3F));
            out += char(0x80 | (cp & 0x3F));
        } else {
            out += char(0xF0 | (cp >> 18));
            out += char(0x80 | ((cp >> 12) & 0x3F));
            out += char(0x80 | ((cp >> 6) & 0x3F));
            out += char(0x80 | (cp & 0x3F));
        }
    }
    std::string string() {
        ++p; // opening quote
        std::string out;
        while (true) {
            if (p >= s.size())
                fail("unterminated string");
            char c = s[p++];
            if (c == '"')
                break;
            if (c != '\\') {
                out += c;
                continue;
            }
            if (p >= s.size())
                fail("bad escape");
            char e = s[p++];
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                if (p + 4 > s.size())
                    fail("bad \\u escape");
                uint32_t cp = (uint32_t)std::strtoul(s.substr(p, 4).c_str(), nullptr, 16);
                p += 4;
                if (cp >= 0xD800 && cp <= 0xDBFF && p + 6 <= s.size() && s[p] == '\\' && s[p + 1] == 'u') {
                    uint32_t lo = (uint32_t)std::strtoul(s.substr(p + 2, 4).c_str(), nullptr, 16);
                    p += 6;
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                append_utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
        return out;
    }
    Value array() {
        ++p;
        Value v;
        v.kind = Value::Array;
        ws();
        if (p < s.size() && s[p] == ']') {
            ++p;
            return v;
        }
        while (true) {
            v.arr.push_back(value());
            ws();
            if (p >= s.size())
                fail("unterminated array");
            if (s[p] == ',') {
                ++p;
                continue;
            }
            if (s[p] == ']') {
                ++p;
                break;
            }
            fail("expected , or ]");
        }
        return v;
    }
    Value object() {
        ++p;
        Value v;
        v.kind = Value::Object;
        ws();
        if (p < s.size() && s[p] == '}') {
            ++p;
            return v;
        }
        while (true) {
            ws();
            if (p >= s.size() || s[p] != '"')
                fail("expected key");
            std::string k = string();
            ws();
            if (p >= s.size() || s[p] != ':')
                fail("expected :");
            ++p;
            Value val = value();
            bool replaced = false;
            for (auto &kv : v.obj)
                if (kv.first == k) {
                    kv.second = std::move(val);
                    replaced = true;
                    break;
                }
            if (!replaced)
                v.obj.emplace_back(std::move(k), std::move(val));
            ws();
            if (p >= s.size())
                fail("unterminated object");
            if (s[p] == ',') {
                ++p;
                continue;
            }
            if (s[p] == '}') {
                ++p;
                break;
            }
            fail("expected , or }");
        }
        return v;
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

} // namespace mjson
