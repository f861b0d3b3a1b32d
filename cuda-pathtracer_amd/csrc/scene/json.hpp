// Minimal JSON reader for scene files (objects, arrays, strings, numbers,
// true/false/null, // and /* */ comments).  It offers the handful of
// accessors the scene loader needs, named after the rapidjson calls the
// reference parser makes (CUDA-PathTracer/tracer/parsescene.cpp:61-67).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <cstdlib>
#include <cctype>

namespace cjson {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<ValuePtr> arr;
  std::vector<std::pair<std::string, ValuePtr>> obj;  // keeps file order

  bool IsArray() const { return kind == Array; }
  bool IsString() const { return kind == String; }
  bool IsObject() const { return kind == Object; }
  bool HasMember(const std::string& k) const {
    for (auto& kv : obj) if (kv.first == k) return true;
    return false;
  }
  const Value& operator[](const std::string& k) const {
    for (auto& kv : obj) if (kv.first == k) return *kv.second;
    throw std::runtime_error("missing key \"" + k + "\"");
  }
  double GetDouble() const {
    if (kind != Number) throw std::runtime_error("expected a number");
    return num;
  }
  int GetInt() const { return (int)GetDouble(); }
  bool GetBool() const {
    if (kind != Bool) throw std::runtime_error("expected a boolean");
    return b;
  }
  const std::string& GetString() const {
    if (kind != String) throw std::runtime_error("expected a string");
    return str;
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}
  ValuePtr parse() {
    ValuePtr v = value();
    ws();
    if (p_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t p_ = 0;

  [[noreturn]] void fail(const std::string& m) {
    throw std::runtime_error("JSON parse error at byte " + std::to_string(p_) + ": " + m);
  }
  void ws() {
    for (;;) {
      while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) ++p_;
      if (p_ + 1 < s_.size() && s_[p_] == '/' && s_[p_ + 1] == '/') {
        while (p_ < s_.size() && s_[p_] != '\n') ++p_;
      } else if (p_ + 1 < s_.size() && s_[p_] == '/' && s_[p_ + 1] == '*') {
        p_ += 2;
        while (p_ + 1 < s_.size() && !(s_[p_] == '*' && s_[p_ + 1] == '/')) ++p_;
        p_ += 2;
      } else {
        return;
      }
    }
  }
  ValuePtr value() {
    ws();
    if (p_ >= s_.size()) fail("unexpected end");
    char c = s_[p_];
    auto v = std::make_shared<Value>();
    if (c == '{') {
      v->kind = Value::Object;
      ++p_;
      ws();
      if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
      for (;;) {
        ws();
        std::string k = string();
        ws();
        if (p_ >= s_.size() || s_[p_] != ':') fail("expected ':'");
        ++p_;
        v->obj.emplace_back(k, value());
        ws();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == '}') { ++p_; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v->kind = Value::Array;
      ++p_;
      ws();
      if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
      for (;;) {
        v->arr.push_back(value());
        ws();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == ']') { ++p_; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v->kind = Value::String;
      v->str = string();
    } else if (s_.compare(p_, 4, "true") == 0) {
      v->kind = Value::Bool; v->b = true; p_ += 4;
    } else if (s_.compare(p_, 5, "false") == 0) {
      v->kind = Value::Bool; v->b = false; p_ += 5;
    } else if (s_.compare(p_, 4, "null") == 0) {
      p_ += 4;
    } else {
      const char* st = s_.c_str() + p_;
      char* en = nullptr;
      double d = std::strtod(st, &en);
      if (en == st) fail("unexpected character");
      v->kind = Value::Number;
      v->num = d;
      p_ += (size_t)(en - st);
    }
    return v;
  }
  std::string string() {
    if (p_ >= s_.size() || s_[p_] != '"') fail("expected string");
    ++p_;
    std::string out;
    while (p_ < s_.size() && s_[p_] != '"') {
      char c = s_[p_++];
      if (c == '\\' && p_ < s_.size()) {
        char e = s_[p_++];
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // keep only the low byte: scene files are ASCII
            if (p_ + 4 > s_.size()) fail("bad \\u escape");
            out += (char)std::strtol(s_.substr(p_, 4).c_str(), nullptr, 16);
            p_ += 4;
            break;
          }
          default: out += e;
        }
      } else {
        out += c;
      }
    }
    if (p_ >= s_.size()) fail("unterminated string");
    ++p_;
    return out;
  }
};

inline ValuePtr parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace cjson
