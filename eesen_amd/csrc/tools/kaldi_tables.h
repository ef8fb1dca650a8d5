// kaldi_tables.h -- the Kaldi table formats the two tools read and write, in plain host C++ (shared by
// train_ctc_parallel.cc and net_output_extract.cc).  Formats follow the reference: archive entry = key + space + object
// (src/util/kaldi-table-inl.h), binary objects start with \0B (src/base/io-funcs-inl.h:183-187), float matrix FM
// (src/cpucompute/matrix.cc:968-994), compressed CM / CM2 (src/cpucompute/compressed-matrix.cc:437-520), int32 vectors
// (src/util/kaldi-holder-inl.h:190-260), script files `key path[:offset]` (src/util/kaldi-table.h).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ext/stdio_filebuf.h>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

namespace ktab {

// The "extended filenames" of src/util/kaldi-io.cc that the recipes use: a plain path, `-` (stdin / stdout), `cmd |` (read
// the command's output) and `| cmd` (write into the command) -- steps/train_ctc_parallel.sh:95-115 always hands the
// trainer `ark,s,cs:apply-cmvn ... |` and `ark:gunzip -c labels.tr.gz|`, the decoding scripts pipe `net-output-extract ... ark:-`.
class InStream {
 public:
  explicit InStream(const std::string& xname) {
    std::string n = xname;
    while (!n.empty() && n.back() == ' ') n.pop_back();
    if (n == "-") { is_ = &std::cin; return; }
    if (!n.empty() && n.back() == '|') {
      n.pop_back();
      cmd_ = n;
      pipe_ = popen(n.c_str(), "r");
      if (!pipe_) throw std::runtime_error("cannot run '" + n + "'");
      buf_.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_, std::ios::in | std::ios::binary));
      own_.reset(new std::istream(buf_.get()));
      is_ = own_.get();
      return;
    }
    auto f = new std::ifstream(n, std::ios::binary);
    own_.reset(f);
    if (!*f) throw std::runtime_error("cannot open " + n);
    is_ = f;
  }
  // A filter that failed leaves a TRUNCATED table behind an ordinary end-of-file (the reference warns, util/kaldi-io.cc
  // PipeInputImpl::Close; the Python reader raises): say so as soon as the stream is closed.
  ~InStream() {
    own_.reset(); buf_.reset();
    if (pipe_) {
      const int st = pclose(pipe_);
      if (st != 0) std::cerr << "WARNING (kaldi_tables) pipe '" << cmd_ << "' ended with status " << st << ": the table may be truncated" << std::endl;
    }
  }
  // to be called once the reader has hit end-of-file on a pipe it must be able to trust (feature and label tables of a trainer)
  void CloseChecked() {
    own_.reset(); buf_.reset();
    if (pipe_) {
      const int st = pclose(pipe_);
      pipe_ = nullptr;
      if (st != 0) throw std::runtime_error("pipe '" + cmd_ + "' ended with status " + std::to_string(st) + ": the table it produced is truncated");
    }
  }
  InStream(const InStream&) = delete;
  std::istream& get() { return *is_; }
 private:
  std::string cmd_;
  std::istream* is_ = nullptr;
  std::unique_ptr<std::istream> own_;
  std::unique_ptr<__gnu_cxx::stdio_filebuf<char>> buf_;
  FILE* pipe_ = nullptr;
};
class OutStream {
 public:
  explicit OutStream(const std::string& xname) {
    std::string n = xname;
    if (n == "-") { os_ = &std::cout; return; }
    if (!n.empty() && n.front() == '|') {
      pipe_ = popen(n.c_str() + 1, "w");
      if (!pipe_) throw std::runtime_error("cannot run '" + n.substr(1) + "'");
      buf_.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_, std::ios::out | std::ios::binary));
      own_.reset(new std::ostream(buf_.get()));
      os_ = own_.get();
      return;
    }
    auto f = new std::ofstream(n, std::ios::binary);
    own_.reset(f);
    if (!*f) throw std::runtime_error("cannot open " + n + " for writing");
    os_ = f;
  }
  ~OutStream() {
    if (os_) os_->flush();
    own_.reset(); buf_.reset();
    if (pipe_) {
      const int st = pclose(pipe_);
      if (st != 0) std::cerr << "WARNING (kaldi_tables) output pipe ended with status " << st << std::endl;
    }
  }
  OutStream(const OutStream&) = delete;
  std::ostream& get() { return *os_; }
 private:
  std::ostream* os_ = nullptr;
  std::unique_ptr<std::ostream> own_;
  std::unique_ptr<__gnu_cxx::stdio_filebuf<char>> buf_;
  FILE* pipe_ = nullptr;
};

struct Mat {
  std::vector<float> v;
  int rows = 0, cols = 0;
};

inline void need(std::istream& is, const char* what) {
  if (!is) throw std::runtime_error(std::string("table read error: ") + what);
}
inline int32_t read_sized_int(std::istream& is) {  // io-funcs-inl.h:32-41: size byte 4 + little-endian payload
  char sz = 0;
  is.get(sz);
  int32_t v = 0;
  is.read(reinterpret_cast<char*>(&v), 4);
  if (!is || sz != 4) throw std::runtime_error("bad binary int32 in table");
  return v;
}
inline std::string read_key(std::istream& is) {  // bytes up to the first space; "" at EOF; leading newlines skipped (text mode)
  std::string k;
  char c;
  while (is.get(c)) {
    if (c == ' ' || c == '\t') { if (!k.empty()) return k; continue; }
    if ((c == '\n' || c == '\r') && k.empty()) continue;
    k.push_back(c);
  }
  return k;
}
inline float u16_to_float(float min_value, float range, uint16_t v) {  // compressed-matrix.cc:244-250
  return min_value + range * 1.52590218966964e-05F * v;
}
inline float char_to_float(float p0, float p25, float p75, float p100, unsigned char value) {  // compressed-matrix.cc:363-373
  if (value <= 64) return p0 + (p25 - p0) * value * (1 / 64.0);
  if (value <= 192) return p25 + (p75 - p25) * (value - 64) * (1 / 128.0);
  return p75 + (p100 - p75) * (value - 192) * (1 / 63.0);
}
inline Mat read_compressed(std::istream& is, int format) {  // compressed-matrix.cc:437-470 + CopyToMat :485-520
  struct { float min_value, range; int32_t num_rows, num_cols; } h;
  is.read(reinterpret_cast<char*>(&h), sizeof(h));
  need(is, "compressed-matrix header");
  Mat m;
  m.rows = h.num_rows; m.cols = h.num_cols;
  if (h.num_cols == 0) { m.rows = 0; return m; }
  m.v.resize((size_t)m.rows * m.cols);
  if (format == 2) {
    std::vector<uint16_t> d((size_t)m.rows * m.cols);
    is.read(reinterpret_cast<char*>(d.data()), d.size() * 2);
    need(is, "CM2 data");
    for (size_t i = 0; i < d.size(); ++i) m.v[i] = u16_to_float(h.min_value, h.range, d[i]);
    return m;
  }
  std::vector<uint16_t> pc((size_t)4 * m.cols);
  is.read(reinterpret_cast<char*>(pc.data()), pc.size() * 2);
  std::vector<unsigned char> bytes((size_t)m.rows * m.cols);
  is.read(reinterpret_cast<char*>(bytes.data()), bytes.size());
  need(is, "CM data");
  for (int c = 0; c < m.cols; ++c) {
    const float p0 = u16_to_float(h.min_value, h.range, pc[4 * c]), p25 = u16_to_float(h.min_value, h.range, pc[4 * c + 1]),
                p75 = u16_to_float(h.min_value, h.range, pc[4 * c + 2]), p100 = u16_to_float(h.min_value, h.range, pc[4 * c + 3]);
    for (int r = 0; r < m.rows; ++r) m.v[(size_t)r * m.cols + c] = char_to_float(p0, p25, p75, p100, bytes[(size_t)c * m.rows + r]);
  }
  return m;
}
inline Mat read_matrix(std::istream& is) {
  Mat m;
  if (is.peek() == '\0') {  // binary: \0B then a token
    is.get(); is.get();
    std::string tok;
    is >> tok;
    is.get();  // the space after the token
    if (tok == "CM") return read_compressed(is, 1);
    if (tok == "CM2") return read_compressed(is, 2);
    if (tok == "DM") throw std::runtime_error("double-precision matrices are not supported (the path is BaseFloat = float)");
    if (tok != "FM") throw std::runtime_error("expected FM / CM / CM2, got " + tok);
    m.rows = read_sized_int(is); m.cols = read_sized_int(is);
    m.v.resize((size_t)m.rows * m.cols);
    is.read(reinterpret_cast<char*>(m.v.data()), m.v.size() * 4);
    need(is, "matrix data");
    return m;
  }
  // text: " [" rows separated by newlines "]"
  char c;
  do { need(is.get(c), "text matrix"); } while (c != '[');
  std::string body;
  std::getline(is, body, ']');
  std::istringstream rows(body);
  std::string line;
  while (std::getline(rows, line)) {
    std::istringstream ls(line);
    float f;
    int n = 0;
    while (ls >> f) { m.v.push_back(f); ++n; }
    if (n) { if (m.cols && n != m.cols) throw std::runtime_error("ragged text matrix"); m.cols = n; ++m.rows; }
  }
  std::getline(is, line);  // rest of the closing line
  return m;
}
inline std::vector<int32_t> read_int_vector(std::istream& is) {
  std::vector<int32_t> v;
  if (is.peek() == '\0') {
    is.get(); is.get();
    const int32_t n = read_sized_int(is);
    v.resize(n);
    for (int32_t i = 0; i < n; ++i) v[i] = read_sized_int(is);
    return v;
  }
  std::string line;
  std::getline(is, line);
  std::istringstream ls(line);
  int32_t x;
  while (ls >> x) v.push_back(x);
  return v;
}
struct Spec { std::string kind, path; };
inline Spec parse_spec(const std::string& s) {
  const size_t c = s.find(':');
  if (c == std::string::npos) throw std::runtime_error("bad table specifier '" + s + "' (expected ark:... or scp:...)");
  Spec sp{s.substr(0, s.find_first_of(",:")), s.substr(c + 1)};
  if (sp.kind != "ark" && sp.kind != "scp") throw std::runtime_error("unsupported table kind in '" + s + "'");
  return sp;
}
// SequentialBaseFloatMatrixReader (train-ctc-parallel.cc:124)
class FeatureReader {
 public:
  explicit FeatureReader(const std::string& rspecifier) : sp_(parse_spec(rspecifier)), in_(sp_.path), f_(in_.get()) { Next(); }
  bool Done() const { return done_; }
  const std::string& Key() const { return key_; }
  Mat& Value() { return val_; }
  void Next() {
    if (done_) return;
    if (sp_.kind == "ark") {
      key_ = read_key(f_);
      if (key_.empty()) { done_ = true; in_.CloseChecked(); return; }
      val_ = read_matrix(f_);
      return;
    }
    std::string line;
    while (std::getline(f_, line)) {
      // `key<whitespace>rest-of-line` (util/kaldi-table.cc ReadScriptFile): the entry is everything behind the first run of
      // whitespace that follows the key -- NOT the first occurrence of its first word, which may sit inside the key itself
      const size_t k0 = line.find_first_not_of(" \t");
      if (k0 == std::string::npos) continue;
      const size_t k1 = line.find_first_of(" \t", k0);
      if (k1 == std::string::npos) continue;
      const size_t r0 = line.find_first_not_of(" \t", k1);
      if (r0 == std::string::npos) continue;
      key_ = line.substr(k0, k1 - k0);
      const std::string whole = line.substr(r0);
      std::string loc = whole.substr(0, whole.find_first_of(" \t"));
      std::streamoff off = 0;
      const size_t c = loc.rfind(':');
      if (c != std::string::npos && c + 1 < loc.size() && loc.find_first_not_of("0123456789", c + 1) == std::string::npos) {
        off = std::stoll(loc.substr(c + 1));
        loc = loc.substr(0, c);
      }
      // a script entry may itself be a command: `key cmd args |`
      if (!whole.empty() && whole.find_last_not_of(" \t") != std::string::npos && whole[whole.find_last_not_of(" \t")] == '|') {
        InStream a(whole);
        val_ = read_matrix(a.get());
        return;
      }
      std::ifstream a(loc, std::ios::binary);
      if (!a) throw std::runtime_error("cannot open " + loc);
      a.seekg(off);
      val_ = read_matrix(a);
      return;
    }
    done_ = true;
  }
 private:
  Spec sp_;
  InStream in_;
  std::istream& f_;
  std::string key_;
  Mat val_;
  bool done_ = false;
};
// RandomAccessInt32VectorReader (train-ctc-parallel.cc:125): the whole table in memory
inline std::map<std::string, std::vector<int32_t>> read_targets(const std::string& rspecifier) {
  const Spec sp = parse_spec(rspecifier);
  if (sp.kind != "ark") throw std::runtime_error("labels: only ark: tables are supported");
  InStream in(sp.path);
  std::istream& f = in.get();
  std::map<std::string, std::vector<int32_t>> t;
  for (;;) {
    const std::string k = read_key(f);
    if (k.empty()) break;
    t[k] = read_int_vector(f);
  }
  in.CloseChecked();   // `ark:gunzip -c labels.tr.gz|` that failed must not leave the trainer with half the transcripts
  return t;
}


// BaseFloatMatrixWriter to `ark:file` (binary) or `ark,t:file` (text)
class MatrixWriter {
 public:
  explicit MatrixWriter(const std::string& wspecifier) {
    const Spec sp = parse_spec(wspecifier);
    if (sp.kind != "ark") throw std::runtime_error("only ark: output is supported");
    text_ = wspecifier.substr(0, wspecifier.find(':')).find(",t") != std::string::npos;
    out_.reset(new OutStream(sp.path));
  }
  void Write(const std::string& key, const float* data, int rows, int cols, int ld) {
    std::ostream& f_ = out_->get();
    f_ << key << ' ';
    if (text_) {
      f_ << " [";
      f_.precision(9);
      for (int r = 0; r < rows; ++r) {
        f_ << "\n  ";
        for (int c = 0; c < cols; ++c) f_ << data[(size_t)r * ld + c] << ' ';
      }
      f_ << "]\n";
    } else {
      f_.write("\0BFM ", 5);
      const char four = 4;
      const int32_t rc[2] = {rows, cols};
      for (int i = 0; i < 2; ++i) { f_.write(&four, 1); f_.write(reinterpret_cast<const char*>(&rc[i]), 4); }
      for (int r = 0; r < rows; ++r) f_.write(reinterpret_cast<const char*>(data + (size_t)r * ld), (size_t)cols * 4);
    }
    f_.flush();   // every utterance leaves at once: downstream tools of a pipe (`... ark:- | latgen-faster ...`) start on it
    if (!f_) throw std::runtime_error("write error");
  }
 private:
  std::unique_ptr<OutStream> out_;
  bool text_ = false;
};

}  // namespace ktab
