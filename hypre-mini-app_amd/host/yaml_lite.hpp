// Minimal YAML reader for the mini-app's input contract:
// nested block maps, scalars, inline lists [a, b], block lists, comments.
// yaml-cpp headers are not available in this image, so the driver carries its
// own reader with the small part of the yaml-cpp surface it needs:
//   node["key"], if (node["key"]), node.as<T>(), YAML::LoadFile(path).
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace YAML {

class Node {
 public:
  enum Kind { Null, Scalar, Map, Seq };
  Node() : d_(std::make_shared<Data>()) {}

  explicit operator bool() const { return d_->kind != Null; }
  bool IsMap() const { return d_->kind == Map; }
  bool IsSequence() const { return d_->kind == Seq; }
  bool IsScalar() const { return d_->kind == Scalar; }
  size_t size() const { return d_->kind == Seq ? d_->seq.size() : d_->kind == Map ? d_->keys.size() : 0; }

  // map lookup; a missing key yields a Null node (never inserted)
  Node operator[](const std::string &key) const {
    if (d_->kind != Map) return Node();
    auto it = d_->map.find(key);
    return it == d_->map.end() ? Node() : it->second;
  }
  Node operator[](const char *key) const { return (*this)[std::string(key)]; }
  Node operator[](size_t i) const {
    if (d_->kind != Seq || i >= d_->seq.size()) return Node();
    return d_->seq[i];
  }
  const std::vector<std::string> &keys() const { return d_->keys; }

  template <class T>
  T as() const {
    return convert<T>(static_cast<T *>(nullptr));
  }

  // construction helpers (used by the parser)
  static Node scalar(const std::string &s) {
    Node n;
    n.d_->kind = Scalar;
    n.d_->text = s;
    return n;
  }
  void set(const std::string &key, const Node &v) {
    d_->kind = Map;
    if (!d_->map.count(key)) d_->keys.push_back(key);
    d_->map[key] = v;
  }
  void push(const Node &v) {
    d_->kind = Seq;
    d_->seq.push_back(v);
  }
  const std::string &text() const { return d_->text; }

 private:
  struct Data {
    Kind kind = Null;
    std::string text;
    std::map<std::string, Node> map;
    std::vector<std::string> keys;
    std::vector<Node> seq;
  };
  std::shared_ptr<Data> d_;

  void need_scalar() const {
    if (d_->kind != Scalar) throw std::runtime_error("yaml: scalar expected");
  }
  template <class T>
  T convert(std::string *) const {
    need_scalar();
    return d_->text;
  }
  template <class T>
  T convert(int *) const {
    need_scalar();
    char *end = nullptr;
    long v = std::strtol(d_->text.c_str(), &end, 10);
    if (end == d_->text.c_str()) throw std::runtime_error("yaml: integer expected, got '" + d_->text + "'");
    if (*end == '.' || *end == 'e' || *end == 'E') v = (long)std::strtod(d_->text.c_str(), nullptr);
    return (int)v;
  }
  template <class T>
  T convert(long long *) const {
    need_scalar();
    return std::strtoll(d_->text.c_str(), nullptr, 10);
  }
  template <class T>
  T convert(double *) const {
    need_scalar();
    char *end = nullptr;
    double v = std::strtod(d_->text.c_str(), &end);
    if (end == d_->text.c_str()) throw std::runtime_error("yaml: number expected, got '" + d_->text + "'");
    return v;
  }
  template <class T>
  T convert(bool *) const {
    need_scalar();
    const std::string &s = d_->text;
    if (s == "true" || s == "True" || s == "yes" || s == "on" || s == "1") return true;
    if (s == "false" || s == "False" || s == "no" || s == "off" || s == "0") return false;
    throw std::runtime_error("yaml: boolean expected, got '" + s + "'");
  }
  template <class T, class E>
  T convert(std::vector<E> *) const {
    if (d_->kind != Seq) throw std::runtime_error("yaml: sequence expected");
    std::vector<E> out;
    for (const Node &n : d_->seq) out.push_back(n.as<E>());
    return out;
  }
};

namespace detail {
inline std::string trim(const std::string &s) {
  size_t b = s.find_first_not_of(" \t\r\n");
  if (b == std::string::npos) return "";
  size_t e = s.find_last_not_of(" \t\r\n");
  return s.substr(b, e - b + 1);
}
inline std::string strip_comment(const std::string &s) {
  bool sq = false, dq = false;
  for (size_t i = 0; i < s.size(); i++) {
    if (s[i] == '\'' && !dq) sq = !sq;
    if (s[i] == '"' && !sq) dq = !dq;
    if (s[i] == '#' && !sq && !dq && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
  }
  return s;
}
inline std::string unquote(const std::string &s) {
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
    return s.substr(1, s.size() - 2);
  return s;
}
inline Node parse_value(const std::string &v) {
  std::string t = trim(v);
  if (!t.empty() && t.front() == '[' && t.back() == ']') {
    Node seq;
    seq.push(Node());  // force Seq kind, then drop the placeholder
    Node out;
    std::string inner = t.substr(1, t.size() - 2), item;
    std::stringstream ss(inner);
    bool any = false;
    while (std::getline(ss, item, ',')) {
      std::string it = trim(item);
      if (it.empty()) continue;
      out.push(Node::scalar(unquote(it)));
      any = true;
    }
    if (!any) {
      Node empty;
      empty.push(Node());
      return Node();  // empty list reads as Null
    }
    return out;
  }
  return Node::scalar(unquote(t));
}
struct Line {
  int indent;
  std::string text;
};
inline Node parse_block(const std::vector<Line> &lines, size_t &pos, int indent) {
  Node node;
  while (pos < lines.size() && lines[pos].indent >= indent) {
    const Line &ln = lines[pos];
    if (ln.indent > indent) throw std::runtime_error("yaml: unexpected indentation near '" + ln.text + "'");
    if (ln.text.rfind("- ", 0) == 0 || ln.text == "-") {
      std::string rest = trim(ln.text.substr(1));
      pos++;
      if (rest.empty()) {
        node.push(pos < lines.size() && lines[pos].indent > indent ? parse_block(lines, pos, lines[pos].indent) : Node());
      } else
        node.push(parse_value(rest));
      continue;
    }
    size_t colon = ln.text.find(':');
    if (colon == std::string::npos) throw std::runtime_error("yaml: 'key: value' expected near '" + ln.text + "'");
    std::string key = unquote(trim(ln.text.substr(0, colon)));
    std::string val = trim(ln.text.substr(colon + 1));
    pos++;
    if (val.empty()) {
      if (pos < lines.size() && lines[pos].indent > indent)
        node.set(key, parse_block(lines, pos, lines[pos].indent));
      else
        node.set(key, Node());
    } else
      node.set(key, parse_value(val));
  }
  return node;
}
}  // namespace detail

inline Node Load(const std::string &text) {
  std::vector<detail::Line> lines;
  std::stringstream ss(text);
  std::string raw;
  while (std::getline(ss, raw)) {
    std::string s = detail::strip_comment(raw);
    if (detail::trim(s).empty() || detail::trim(s) == "---") continue;
    int indent = 0;
    while (indent < (int)s.size() && s[(size_t)indent] == ' ') indent++;
    lines.push_back({indent, detail::trim(s)});
  }
  size_t pos = 0;
  if (lines.empty()) return Node();
  return detail::parse_block(lines, pos, lines[0].indent);
}

inline Node LoadFile(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("yaml: cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return Load(ss.str());
}

}  // namespace YAML
