// parse_options.h -- the command-line conventions of the reference's tools, for the native tools of this repository.
//
// Behaviour follows /root/reference/src/util/parse-options.{h,cc} (written from its description, not its text):
//   * options are `--name=value`; `--name` alone is allowed for booleans only (= true)            (parse-options.cc:521-539,561-584)
//   * names are case-insensitive and `_` == `-`                                                     (:542-556 NormalizeArgName)
//   * named options must PRECEDE the positional arguments: the first argument that does not start with `--` ends them, and so
//     does a lone `--` (which is itself dropped)                                                   (:358-389)
//   * standard options of every tool: --config=<file> (one `--name=value` per line, `#` starts a comment, blank lines ignored;
//     may be repeated; read in a FIRST pass, so the command line overrides it), --print-args (default true: the command line is
//     echoed on stderr), --help (usage on stderr, exit 0), --verbose=<int>                          (parse-options.h:38-51, cc:338-398,470-506)
//   * integers accept decimal, 0x.. and 0.. (strtol base 0); a value that does not parse, a boolean that is not
//     true|t|1|false|f|0 (any case), `--bool=`, a string option without `=`, or an unknown name print the usage with the
//     command line and raise "Invalid option ..." -- the tools catch it, print it and return -1 like the reference's main()
//                                                                                                   (:373-376,561-656, train-ctc-parallel.cc:259-263)
//   * usage layout: blank line, usage text, "Options:" with `  --name<pad to 25> : doc (type, default = v)`, "Standard options:".
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace eesen_tools {

class ParseOptions {
 public:
  explicit ParseOptions(const std::string& usage) : usage_(usage) {
    reg("config", &config_, "Configuration file to read (this option may be repeated)", true);
    reg("print-args", &print_args_, "Print the command line arguments (to stderr)", true);
    reg("help", &help_, "Print out usage message", true);
    reg("verbose", &verbose_, "Verbose level (higher->more logging)", true);
  }
  void Register(const std::string& name, bool* p, const std::string& doc) { reg(name, p, doc, false); }
  void Register(const std::string& name, int32_t* p, const std::string& doc) { reg(name, p, doc, false); }
  void Register(const std::string& name, float* p, const std::string& doc) { reg(name, p, doc, false); }
  void Register(const std::string& name, double* p, const std::string& doc) { reg(name, p, doc, false); }
  void Register(const std::string& name, std::string* p, const std::string& doc) { reg(name, p, doc, false); }

  // Returns normally with the positional arguments collected; --help prints the usage and exits 0; errors throw.
  void Read(int argc, const char* const* argv) {
    argc_ = argc; argv_ = argv;
    std::string key, value;
    bool eq;
    for (int i = 1; i < argc; ++i) {   // first pass: --config and --help, wherever they stand among the named options
      if (std::strncmp(argv[i], "--", 2) != 0) continue;
      if (std::strcmp(argv[i], "--") == 0) break;
      split(argv[i], &key, &value, &eq);
      if (key == "config") read_config(value);
      if (key == "help") { PrintUsage(); std::exit(0); }
    }
    int i = 1;
    bool dd = false;
    for (; i < argc; ++i) {            // second pass: the named options, up to the first positional argument or a lone "--"
      if (std::strncmp(argv[i], "--", 2) != 0) break;
      if (std::strcmp(argv[i], "--") == 0) { ++i; dd = true; break; }
      split(argv[i], &key, &value, &eq);
      if (!set(key, value, eq)) { PrintUsage(true); throw std::runtime_error(std::string("Invalid option ") + argv[i]); }
    }
    for (; i < argc; ++i) {
      if (std::strcmp(argv[i], "--") == 0 && !dd) dd = true;
      else args_.push_back(argv[i]);
    }
    if (print_args_) {
      std::ostringstream s;
      for (int j = 0; j < argc; ++j) s << escape(argv[j]) << " ";
      s << '\n';
      std::cerr << s.str() << std::flush;
    }
  }
  int NumArgs() const { return (int)args_.size(); }
  const std::string& GetArg(int i) const {   // 1-based, as in the reference
    if (i < 1 || i > (int)args_.size()) throw std::runtime_error("ParseOptions::GetArg, invalid index " + std::to_string(i));
    return args_[i - 1];
  }
  int Verbose() const { return verbose_; }

  void PrintUsage(bool print_command_line = false) const {
    std::cerr << '\n' << usage_ << '\n';
    bool header = false;
    for (const auto& kv : doc_)
      if (!kv.second.standard) {
        if (!header) { std::cerr << "Options:" << '\n'; header = true; }
        std::cerr << "  --" << std::setw(25) << std::left << kv.second.name << " : " << kv.second.doc << '\n';
      }
    if (header) std::cerr << '\n';
    std::cerr << "Standard options:" << '\n';
    for (const auto& kv : doc_)
      if (kv.second.standard) std::cerr << "  --" << std::setw(25) << std::left << kv.second.name << " : " << kv.second.doc << '\n';
    std::cerr << '\n';
    if (print_command_line) {
      std::ostringstream s;
      s << "Command line was: ";
      for (int j = 0; j < argc_; ++j) s << escape(argv_[j]) << " ";
      s << '\n';
      std::cerr << s.str() << std::flush;
    }
  }

 private:
  struct Doc { std::string name, doc; bool standard; };
  enum Kind { kBool, kInt, kFloat, kDouble, kString };
  struct Slot { Kind kind; void* p; };

  static std::string norm(const std::string& s) {
    std::string o;
    for (char c : s) o += c == '_' ? '-' : (char)std::tolower((unsigned char)c);
    return o;
  }
  static std::string trim(const std::string& s) {
    const char* ws = " \t\n\r\f\v";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
  }
  template <class T> static std::string show(const T& v) { std::ostringstream s; s << v; return s.str(); }
  void reg(const std::string& name, bool* p, const std::string& doc, bool st) { add(name, {kBool, p}, doc + " (bool, default = " + (*p ? "true)" : "false)"), st); }
  void reg(const std::string& name, int32_t* p, const std::string& doc, bool st) { add(name, {kInt, p}, doc + " (int, default = " + show(*p) + ")", st); }
  void reg(const std::string& name, float* p, const std::string& doc, bool st) { add(name, {kFloat, p}, doc + " (float, default = " + show(*p) + ")", st); }
  void reg(const std::string& name, double* p, const std::string& doc, bool st) { add(name, {kDouble, p}, doc + " (double, default = " + show(*p) + ")", st); }
  void reg(const std::string& name, std::string* p, const std::string& doc, bool st) { add(name, {kString, p}, doc + " (string, default = \"" + *p + "\")", st); }
  void add(const std::string& name, Slot s, const std::string& doc, bool st) {
    const std::string k = norm(name);
    if (slot_.count(k)) { std::cerr << "WARNING (ParseOptions) Registering option twice, ignoring second time: " << name << '\n'; return; }
    slot_[k] = s;
    doc_[k] = Doc{name, doc, st};
  }
  void split(const std::string& in, std::string* key, std::string* value, bool* eq) const {
    const size_t pos = in.find('=');
    if (pos == std::string::npos) { *key = in.substr(2); *value = ""; *eq = false; }
    else if (pos == 2) { PrintUsage(true); throw std::runtime_error("Invalid option (no key): " + in); }
    else { *key = in.substr(2, pos - 2); *value = in.substr(pos + 1); *eq = true; }
    *key = norm(*key);
    *value = trim(*value);
  }
  [[noreturn]] void bad(const std::string& what) const { PrintUsage(true); throw std::runtime_error(what); }
  bool set(const std::string& key, const std::string& value, bool eq) {
    auto it = slot_.find(key);
    if (it == slot_.end()) return false;
    const Slot s = it->second;
    char* end = nullptr;
    switch (s.kind) {
      case kBool: {
        if (eq && value.empty()) throw std::runtime_error("Invalid option --" + key + "=");
        std::string v = value;
        std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        if (v == "true" || v == "t" || v == "1" || v.empty()) *static_cast<bool*>(s.p) = true;
        else if (v == "false" || v == "f" || v == "0") *static_cast<bool*>(s.p) = false;
        else bad("Invalid format for boolean argument [expected true or false]: " + value);
        break;
      }
      case kInt: {
        const long v = std::strtol(value.c_str(), &end, 0);
        if (end == value.c_str()) bad("Invalid integer option \"" + value + "\"");
        *static_cast<int32_t*>(s.p) = (int32_t)v;
        break;
      }
      case kFloat: {
        const double v = std::strtod(value.c_str(), &end);
        if (end == value.c_str()) bad("Invalid floating-point option \"" + value + "\"");
        *static_cast<float*>(s.p) = (float)v;
        break;
      }
      case kDouble: {
        const double v = std::strtod(value.c_str(), &end);
        if (end == value.c_str()) bad("Invalid floating-point option  \"" + value + "\"");
        *static_cast<double*>(s.p) = v;
        break;
      }
      case kString:
        if (!eq) throw std::runtime_error("Invalid option --" + key);
        *static_cast<std::string*>(s.p) = value;
        break;
    }
    return true;
  }
  void read_config(const std::string& filename) {
    std::ifstream is(filename.c_str());
    if (!is.good()) throw std::runtime_error("Cannot open config file: " + filename);
    std::string line, key, value;
    bool eq;
    int n = 0;
    while (std::getline(is, line)) {
      ++n;
      const size_t h = line.find('#');
      if (h != std::string::npos) line.erase(h);
      line = trim(line);
      if (line.empty()) continue;
      if (line.compare(0, 2, "--") != 0)
        throw std::runtime_error("Reading config file " + filename + ": line " + std::to_string(n) + " does not look like a line from a "
                                 "command-line program's config file: should be of the form --x=y.  Note: config files intended to be "
                                 "sourced by shell scripts lack the '--'.");
      split(line, &key, &value, &eq);
      if (!set(key, value, eq)) { PrintUsage(true); throw std::runtime_error("Invalid option " + line + " in config file " + filename); }
    }
  }
  // shell-style quoting of an argument for the echo of the command line (parse-options.cc:262-306): left alone when made of
  // characters the shell does not interpret, otherwise single-quoted (double-quoted when it contains a single quote itself)
  static std::string escape(const std::string& s) {
    const char* ok = "[]~#^_-+=:.,/";
    bool plain = !s.empty();
    for (char c : s)
      if (!std::isalnum((unsigned char)c) && !std::strchr(ok, c)) { plain = false; break; }
    if (plain) return s;
    const char q = s.find('\'') == std::string::npos ? '\'' : '"';
    std::string o(1, q);
    for (char c : s) {
      if (q == '"' && (c == '"' || c == '\\' || c == '$' || c == '`')) o += '\\';
      o += c;
    }
    o += q;
    return o;
  }

  std::string usage_;
  std::string config_;
  bool print_args_ = true, help_ = false;
  int32_t verbose_ = 0;
  int argc_ = 0;
  const char* const* argv_ = nullptr;
  std::map<std::string, Slot> slot_;
  std::map<std::string, Doc> doc_;
  std::vector<std::string> args_;
};

}  // namespace eesen_tools
