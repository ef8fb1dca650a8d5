// feat_pipeline.h -- host side of the device feature front end (include/eesen_hip.h, eesen_feeder_set_pipeline), shared by
// train_ctc_parallel.cc and net_output_extract.cc.
//
// The recipes hand both tools their features as an rspecifier that is a pipe of the reference's own filters
// (asr_egs/wsj/steps/train_ctc_parallel.sh:95-110, decode_ctc_lat.sh:92-95, librispeech/steps/train_ctc_parallel_mult.sh:110-133):
//   ark,s,cs:apply-cmvn --norm-vars=true --utt2spk=ark:data/utt2spk scp:data/cmvn.scp scp:exp/train.scp ark:- |
//            splice-feats --left-context=1 --right-context=1 ark:- ark:- | subsample-feats --n=3 --offset=0 ark:- ark:- | add-deltas ark:- ark:- |
// parse_feature_pipeline recognises exactly such command lines (the option sets of src/featbin/{apply-cmvn,copy-feats,splice-feats,
// subsample-feats,add-deltas}.cc); the tool then reads the RAW table itself, looks the CMVN statistics up per utterance, and the
// filters run on the GPU inside the batch assembly.  Anything else (another tool, an option not implemented, e.g. --skip-dims)
// is left to the shell: the rspecifier is opened as the pipe it is.  Same logic as eesen_amd/frontend.py.
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "../../../include/eesen_hip.h"
#include "kaldi_tables.h"

namespace ktab {

struct Pipeline {
  std::string source;                       // rspecifier of the raw features
  std::vector<eesen_feat_stage_t> stages;   // in order
  std::string cmvn, utt2spk;                // statistics rspecifier / rxfilename ("" = no CMVN stage), utt2spk rspecifier
  bool norm_vars = false;
};

inline bool shell_split(const std::string& s, std::vector<std::string>* out) {  // whitespace-separated words, '...' and "..." quoting
  out->clear();
  std::string cur;
  bool have = false;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (c == '\'' || c == '"') {
      const size_t e = s.find(c, i + 1);
      if (e == std::string::npos) return false;
      cur += s.substr(i + 1, e - i - 1);
      have = true;
      i = e;
    } else if (c == ' ' || c == '\t' || c == '\n') {
      if (have) { out->push_back(cur); cur.clear(); have = false; }
    } else {
      cur += c;
      have = true;
    }
  }
  if (have) out->push_back(cur);
  return true;
}

inline bool to_bool(const std::string& v, bool* b) {  // ParseOptions::ToBool (src/util/parse-options.cc)
  std::string l;
  for (char c : v) l += (char)tolower(c);
  if (l.empty() || l == "true" || l == "t" || l == "1") { *b = true; return true; }
  if (l == "false" || l == "f" || l == "0") { *b = false; return true; }
  return false;
}
inline bool to_int(const std::string& v, int* x) {
  if (v.empty()) return false;
  char* end = nullptr;
  const long r = strtol(v.c_str(), &end, 10);
  if (*end) return false;
  *x = (int)r;
  return true;
}

// `--name=value` options (names with '_' for '-' accepted, as ParseOptions does) and positional arguments
inline bool split_options(const std::vector<std::string>& argv, size_t from, std::map<std::string, std::string>* opts,
                          std::vector<std::string>* pos) {
  for (size_t i = from; i < argv.size(); ++i) {
    const std::string& a = argv[i];
    if (a.rfind("--", 0) == 0) {
      const size_t eq = a.find('=');
      std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      for (char& c : name) if (c == '_') c = '-';
      (*opts)[name] = eq == std::string::npos ? "" : a.substr(eq + 1);
    } else {
      pos->push_back(a);
    }
  }
  return true;
}

inline bool table_kind(const std::string& spec) {  // ark:... / scp:... (with options), not itself a command
  const size_t c = spec.find(':');
  if (c == std::string::npos) return false;
  const std::string k = spec.substr(0, spec.find_first_of(",:"));
  return k == "ark" || k == "scp";
}

inline bool parse_feature_pipeline(const std::string& rspecifier, Pipeline* out) {
  const size_t colon = rspecifier.find(':');
  if (colon == std::string::npos || rspecifier.substr(0, rspecifier.find_first_of(",:")) != "ark") return false;
  std::string cmd = rspecifier.substr(colon + 1);
  while (!cmd.empty() && isspace((unsigned char)cmd.back())) cmd.pop_back();
  if (cmd.empty() || cmd.back() != '|') return false;
  cmd.pop_back();
  std::vector<std::string> segs;
  {
    size_t b = 0;
    bool in_q = false;
    char q = 0;
    for (size_t i = 0; i <= cmd.size(); ++i) {
      if (i < cmd.size() && (cmd[i] == '\'' || cmd[i] == '"')) { if (!in_q) { in_q = true; q = cmd[i]; } else if (q == cmd[i]) in_q = false; }
      if (i == cmd.size() || (cmd[i] == '|' && !in_q)) { segs.push_back(cmd.substr(b, i - b)); b = i + 1; }
    }
  }
  Pipeline p;
  for (size_t si = 0; si < segs.size(); ++si) {
    std::vector<std::string> argv, pos;
    std::map<std::string, std::string> o;
    if (!shell_split(segs[si], &argv) || argv.empty()) return false;
    std::string tool = argv[0];
    if (tool.find('/') != std::string::npos) tool = tool.substr(tool.rfind('/') + 1);
    split_options(argv, 1, &o, &pos);
    auto only = [&](std::initializer_list<const char*> names) {
      for (const auto& kv : o) { bool ok = false; for (const char* n : names) ok = ok || kv.first == n; if (!ok) return false; }
      return true;
    };
    if (si == 0) {
      if (tool == "apply-cmvn") {                         // featbin/apply-cmvn.cc:36-47
        if (!only({"utt2spk", "norm-vars", "norm-means"}) || pos.size() != 3 || pos[2] != "ark:-") return false;
        bool norm_means = true, norm_vars = false;
        if (o.count("norm-means") && !to_bool(o["norm-means"], &norm_means)) return false;
        if (o.count("norm-vars") && !to_bool(o["norm-vars"], &norm_vars)) return false;
        if (norm_vars && !norm_means) return false;       // the tool itself refuses this (:55-56): let it say so
        p.source = pos[1];
        if (norm_means) {
          p.cmvn = pos[0];
          p.utt2spk = o.count("utt2spk") ? o["utt2spk"] : "";
          p.norm_vars = norm_vars;
          p.stages.push_back({EESEN_FEAT_CMVN, norm_vars ? 1 : 0, 0});
        }
      } else if (tool == "copy-feats") {
        if (!o.empty() || pos.size() != 2 || pos[1] != "ark:-") return false;
        p.source = pos[0];
      } else {
        return false;
      }
      std::string src = p.source;
      while (!src.empty() && isspace((unsigned char)src.back())) src.pop_back();
      if (!table_kind(src) || src.back() == '|') return false;
      continue;
    }
    if (pos.size() != 2 || pos[0] != "ark:-" || pos[1] != "ark:-") return false;
    if (tool == "splice-feats") {                          // featbin/splice-feats.cc:36-40: both contexts default to 4
      int L = 4, R = 4;
      if (!only({"left-context", "right-context"})) return false;
      if (o.count("left-context") && !to_int(o["left-context"], &L)) return false;
      if (o.count("right-context") && !to_int(o["right-context"], &R)) return false;
      if (L < 0 || R < 0) return false;
      p.stages.push_back({EESEN_FEAT_SPLICE, L, R});
    } else if (tool == "subsample-feats") {                // featbin/subsample-feats.cc:46-55
      int n = 1, off = 0;
      if (!only({"n", "offset"})) return false;
      if (o.count("n") && !to_int(o["n"], &n)) return false;
      if (o.count("offset") && !to_int(o["offset"], &off)) return false;
      if (n == 0 || off < 0 || (n < 0 && off != 0)) return false;
      p.stages.push_back({EESEN_FEAT_SUBSAMPLE, n, off});
    } else if (tool == "add-deltas") {                     // featbin/add-deltas.cc:33-38; DeltaFeaturesOptions: order 2, window 2
      int order = 2, window = 2, truncate = 0;
      if (!only({"delta-order", "delta-window", "truncate"})) return false;
      if (o.count("delta-order") && !to_int(o["delta-order"], &order)) return false;
      if (o.count("delta-window") && !to_int(o["delta-window"], &window)) return false;
      if (o.count("truncate") && !to_int(o["truncate"], &truncate)) return false;
      if (truncate != 0 || order < 0 || order > 8 || window <= 0 || window > 16) return false;
      p.stages.push_back({EESEN_FEAT_DELTAS, order, window});
    } else {
      return false;
    }
  }
  *out = p;
  return true;
}

// Matrix<double>::Read (src/cpucompute/matrix.cc:1012-1100): DM, or FM converted, or text.  CMVN statistics are doubles.
struct Mat64 { std::vector<double> v; int rows = 0, cols = 0; };
inline Mat64 read_matrix64(std::istream& is) {
  Mat64 m;
  if (is.peek() == '\0') {
    is.get(); is.get();
    std::string tok;
    is >> tok;
    is.get();
    if (tok != "DM" && tok != "FM") throw std::runtime_error("expected DM or FM, got " + tok);
    m.rows = read_sized_int(is); m.cols = read_sized_int(is);
    m.v.resize((size_t)m.rows * m.cols);
    if (tok == "DM") {
      is.read(reinterpret_cast<char*>(m.v.data()), m.v.size() * 8);
    } else {
      std::vector<float> f(m.v.size());
      is.read(reinterpret_cast<char*>(f.data()), f.size() * 4);
      for (size_t i = 0; i < f.size(); ++i) m.v[i] = f[i];
    }
    need(is, "matrix data");
    return m;
  }
  char c;
  do { need(is.get(c), "text matrix"); } while (c != '[');
  std::string body;
  std::getline(is, body, ']');
  std::istringstream rows(body);
  std::string line;
  while (std::getline(rows, line)) {
    std::istringstream ls(line);
    double f;
    int n = 0;
    while (ls >> f) { m.v.push_back(f); ++n; }
    if (n) { if (m.cols && n != m.cols) throw std::runtime_error("ragged text matrix"); m.cols = n; ++m.rows; }
  }
  std::getline(is, line);
  return m;
}

// RandomAccessDoubleMatrixReaderMapped(cmvn_rspecifier, utt2spk_rspecifier) of apply-cmvn.cc:80-81, or the single matrix of its
// rxfilename form (:115-122).  The normaliser ([2 x dim] floats: offsets, scales) is computed once per speaker by the library's
// eesen_cmvn_norm (ApplyCmvn's own arithmetic, src/feat/cmvn.cc:78-108).
class CmvnTable {
 public:
  CmvnTable(const std::string& spec, const std::string& utt2spk, bool norm_vars) : norm_vars_(norm_vars) {
    if (table_kind(spec)) {
      const Spec sp = parse_spec(spec);
      InStream in(sp.path);
      std::istream& f = in.get();
      if (sp.kind == "ark") {
        for (std::string key = read_key(f); !key.empty(); key = read_key(f)) stats_[key] = read_matrix64(f);
      } else {
        std::string line;
        while (std::getline(f, line)) {
          std::istringstream ls(line);
          std::string key, loc;
          if (!(ls >> key >> loc)) continue;
          std::streamoff off = 0;
          const size_t c = loc.rfind(':');
          if (c != std::string::npos && c + 1 < loc.size() && loc.find_first_not_of("0123456789", c + 1) == std::string::npos) {
            off = std::stoll(loc.substr(c + 1));
            loc = loc.substr(0, c);
          }
          std::ifstream a(loc, std::ios::binary);
          if (!a) throw std::runtime_error("cannot open " + loc);
          a.seekg(off);
          stats_[key] = read_matrix64(a);
        }
      }
      if (!utt2spk.empty()) {
        const Spec us = parse_spec(utt2spk);
        if (us.kind != "ark") throw std::runtime_error("utt2spk: only ark: tables are supported");
        InStream uin(us.path);
        std::string line;
        while (std::getline(uin.get(), line)) {
          std::istringstream ls(line);
          std::string u, s;
          if (ls >> u >> s) map_[u] = s;
        }
        mapped_ = true;
      }
    } else {
      if (!utt2spk.empty()) throw std::runtime_error("--utt2spk option not compatible with rxfilename as input (did you forget ark:?)");
      std::string loc = spec;
      std::streamoff off = 0;
      const size_t c = loc.rfind(':');
      if (loc.find('|') == std::string::npos && c != std::string::npos && c + 1 < loc.size() &&
          loc.find_first_not_of("0123456789", c + 1) == std::string::npos) {
        off = std::stoll(loc.substr(c + 1));
        loc = loc.substr(0, c);
      }
      InStream in(loc);
      if (off) in.get().seekg(off);
      global_ = norm(read_matrix64(in.get()));
      is_global_ = true;
    }
  }
  // nullptr: no statistics for this utterance (apply-cmvn then writes nothing for it, :87-92)
  const std::vector<float>* lookup(const std::string& utt) {
    if (is_global_) return &global_;
    std::string key = utt;
    if (mapped_) {
      auto m = map_.find(utt);
      if (m == map_.end()) return nullptr;
      key = m->second;
    }
    auto c = cache_.find(key);
    if (c != cache_.end()) return &c->second;
    auto s = stats_.find(key);
    if (s == stats_.end()) return nullptr;
    return &(cache_[key] = norm(s->second));
  }
  static int dim(const std::vector<float>& n) { return (int)n.size() / 2; }

 private:
  std::vector<float> norm(const Mat64& m) const {
    std::vector<float> out((size_t)2 * std::max(m.cols - 1, 0));
    if (eesen_cmvn_norm(m.v.data(), m.rows, m.cols, norm_vars_ ? 1 : 0, out.data()) != EESEN_OK) throw std::runtime_error(eesen_last_error());
    return out;
  }
  bool norm_vars_, mapped_ = false, is_global_ = false;
  std::map<std::string, Mat64> stats_;
  std::map<std::string, std::string> map_;
  std::map<std::string, std::vector<float>> cache_;
  std::vector<float> global_;
};

}  // namespace ktab
