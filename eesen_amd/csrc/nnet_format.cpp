// nnet_format.cpp -- Eesen's <Nnet> model files (Kaldi stream format, text and binary) for the layer
// kinds on the hot path.  Format facts (all from the reference):
//   * stream header "\0B" when binary (/root/reference/src/base/io-funcs-inl.h:183-187), absent in text;
//   * <Nnet> ... </Nnet> wrapper (src/net/net.cc:325-334); per layer the marker, <InputDim> d,
//     <CellDim> (LSTM kinds) or <OutputDim> d (src/net/layer.cc:138-222);
//   * layer data: optional <LearnRateCoef>, <MaxGrad>, nine dropout tokens, optional accumulators, then the
//     tensors (src/net/bilstm-layer.h:317-493, lstm-layer.h:106-172, affine-trans-layer.h:83-128);
//   * tensors: text " [ ... ]" (rows end in newlines), binary "FM " + rows + cols + raw fp32, "FV " + dim +
//     raw fp32 (src/cpucompute/matrix.cc:968-994, vector.cc:1114-1134); binary int32 / float are a size
//     byte followed by the little-endian payload, bool is a bare 'T' / 'F' (src/base/io-funcs.cc:26-55).
// The parser below is a cursor over the whole file in memory; nothing of the reference's stream classes
// is used.
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>

#include "net.h"

namespace eesen {
namespace {

struct Cursor {
  const std::string& b;
  size_t i = 0;
  bool binary = false;
  explicit Cursor(const std::string& buf) : b(buf) {
    if (b.size() >= 2 && b[0] == '\0' && b[1] == 'B') { binary = true; i = 2; }
  }
  [[noreturn]] void fail(const std::string& what) const {
    throw Error(EESEN_ERR_IO, "model file: " + what + " at byte " + std::to_string(i));
  }
  void ws() { while (i < b.size() && isspace((unsigned char)b[i])) ++i; }
  int peek() {
    if (!binary) ws();
    return i < b.size() ? (unsigned char)b[i] : -1;
  }
  std::string token() {
    if (!binary) ws();
    size_t j = i;
    while (j < b.size() && !isspace((unsigned char)b[j])) ++j;
    if (j == i) fail("expected a token");
    std::string t = b.substr(i, j - i);
    i = j;
    if (i < b.size()) ++i;  // the single separator after a token
    return t;
  }
  void expect(const char* tok) {
    const std::string t = token();
    if (t != tok) fail(std::string("expected ") + tok + ", got " + t);
  }
  template <typename V>
  V basic() {
    V v;
    if (binary) {
      if (i + 1 + sizeof(V) > b.size() || (unsigned char)b[i] != sizeof(V)) fail("bad binary scalar");
      memcpy(&v, b.data() + i + 1, sizeof(V));
      i += 1 + sizeof(V);
    } else {
      const std::string t = token();
      char* end = nullptr;
      errno = 0;
      const double d = strtod(t.c_str(), &end);
      if (end == t.c_str() || *end) fail("bad number '" + t + "'");
      v = (V)d;
    }
    return v;
  }
  bool boolean() {
    const int c = peek();
    if (c != 'T' && c != 'F') fail("expected T or F");
    ++i;
    return c == 'T';
  }
  // reads rows*cols (cols == 0: a vector of `rows`) floats into dst
  void tensor(int rows, int cols, float* dst) {
    const size_t n = (size_t)rows * (cols ? cols : 1);
    if (binary) {
      const std::string t = token();
      if (cols) {
        if (t != "FM") fail("expected FM, got " + t);
        const int r = basic<int32_t>(), c = basic<int32_t>();
        if (r != rows || c != cols) fail("matrix is " + std::to_string(r) + "x" + std::to_string(c) + ", expected " + std::to_string(rows) + "x" + std::to_string(cols));
      } else {
        if (t != "FV") fail("expected FV, got " + t);
        const int d = basic<int32_t>();
        if (d != rows) fail("vector has " + std::to_string(d) + " elements, expected " + std::to_string(rows));
      }
      if (i + n * 4 > b.size()) fail("truncated tensor");
      memcpy(dst, b.data() + i, n * 4);
      i += n * 4;
    } else {
      ws();
      if (i >= b.size() || b[i] != '[') fail("expected '['");
      ++i;
      for (size_t k = 0; k < n; ++k) {
        ws();
        char* end = nullptr;
        const float v = strtof(b.c_str() + i, &end);
        if (end == b.c_str() + i) fail("bad float in tensor");
        dst[k] = v;
        i = end - b.c_str();
      }
      ws();
      if (i >= b.size() || b[i] != ']') fail("tensor has more elements than its layer dimensions allow");
      ++i;
    }
  }
};

struct ParsedLayer {
  int kind, din, dout;
  float coef = 1.f, max_grad = 0.f;
  std::vector<float> flat;  // Net::GetParams order
  std::vector<float> accu;  // same order; empty unless the file carries <...Accus>
  float drop[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // dropout options, token order (Layer::drop)
  bool nonparallel = false;  // the file said <BiLstm> / <Lstm> (layer.cc:39-42): same arithmetic here, and the marker is written back
};

int marker_kind(const std::string& m) {
  // <BiLstm> / <Lstm> are the single-sequence twins with identical parameters (layer.cc:164-170)
  if (m == "<BiLstmParallel>" || m == "<BiLstm>") return EESEN_LAYER_BILSTM_PARALLEL;
  if (m == "<LstmParallel>" || m == "<Lstm>") return EESEN_LAYER_LSTM_PARALLEL;
  if (m == "<AffineTransform>") return EESEN_LAYER_AFFINE;
  if (m == "<Softmax>") return EESEN_LAYER_SOFTMAX;
  if (m == "<Sigmoid>") return EESEN_LAYER_SIGMOID;
  if (m == "<Tanh>") return EESEN_LAYER_TANH;
  return 0;
}

const char* kDropoutTokens[] = {"<ForwardDropoutFactor>", "<ForwardTimeStepDropout>", "<ForwardSequenceDropout>",
                                "<RecurrentTimeStepDropout>", "<RecurrentSequenceDropout>", "<RNNDrop>",
                                "<NoMemLossDropout>", "<RecurrentDropoutFactor>", "<TwiddleForward>"};
const bool kDropoutIsFloat[] = {true, false, false, false, false, false, false, true, false};

}  // namespace

void Net::read(const std::string& path) {
  EESEN_REQUIRE(!finalized && layers.empty(), EESEN_ERR_STATE, "Read needs an empty net");
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error(EESEN_ERR_IO, "cannot open model file " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string buf = ss.str();
  Cursor c(buf);
  std::vector<ParsedLayer> parsed;
  while (true) {
    if (c.peek() < 0) break;
    std::string tok = c.token();
    if (tok == "</Nnet>") break;
    if (tok == "<Nnet>") {
      if (c.peek() < 0) break;
      tok = c.token();
      if (tok == "</Nnet>") break;
    }
    ParsedLayer P;
    P.kind = marker_kind(tok);
    P.nonparallel = tok == "<BiLstm>" || tok == "<Lstm>";
    if (!P.kind) throw Error(EESEN_ERR_INVALID, "layer kind " + tok + " is outside the MI355X hot path (supported: BiLstmParallel, LstmParallel, AffineTransform, Softmax, Sigmoid, Tanh)");
    c.expect("<InputDim>");
    P.din = c.basic<int32_t>();
    const bool lstm = P.kind == EESEN_LAYER_BILSTM_PARALLEL || P.kind == EESEN_LAYER_LSTM_PARALLEL;
    c.expect(lstm ? "<CellDim>" : "<OutputDim>");
    P.dout = c.basic<int32_t>();
    if (lstm || P.kind == EESEN_LAYER_AFFINE) {   // the other kinds carry nothing but their dimensions (layer.cc:195-205)
      const int nd = P.kind == EESEN_LAYER_BILSTM_PARALLEL ? 2 : 1;
      if (lstm && P.dout % nd) c.fail("odd <CellDim> for a BiLstm layer");
      const int H = lstm ? P.dout / nd : 0;
      const size_t count = lstm ? (size_t)nd * ((size_t)4 * H * P.din + (size_t)4 * H * H + 7 * H) : (size_t)P.dout * P.din + P.dout;
      auto read_tensors = [&](std::vector<float>& dst) {  // the layer's tensor list, identical for weights and accumulators
        dst.resize(count);
        float* p = dst.data();
        if (lstm) {
          for (int d = 0; d < nd; ++d) {
            c.tensor(4 * H, P.din, p); p += (size_t)4 * H * P.din;
            c.tensor(4 * H, H, p);     p += (size_t)4 * H * H;
            c.tensor(4 * H, 0, p);     p += 4 * H;
            for (int g = 0; g < 3; ++g) { c.tensor(H, 0, p); p += H; }
          }
        } else {
          c.tensor(P.dout, P.din, p);
          c.tensor(P.dout, 0, p + (size_t)P.dout * P.din);
        }
      };
      while (c.peek() == '<') {
        const std::string t = c.token();
        if (t == "<LearnRateCoef>") P.coef = c.basic<float>();
        else if (t == "<MaxGrad>") P.max_grad = c.basic<float>();
        else if (t == "<BiLstmAccus>" || t == "<LstmAccus>" || t == "<AffineAccus>") {
          // Adagrad / RMSProp accumulators precede the weights (bilstm-layer.h:376-395, lstm-layer.h:118-128,
          // affine-trans-layer.h:99-106)
          read_tensors(P.accu);
          break;
        } else {
          bool known = false;
          for (int k = 0; k < 9; ++k)
            if (t == kDropoutTokens[k]) {
              known = true;
              P.drop[k] = kDropoutIsFloat[k] ? c.basic<float>() : (c.boolean() ? 1.f : 0.f);
              if (P.kind != EESEN_LAYER_BILSTM_PARALLEL && P.drop[k] != 0.f)   // only BiLstm reads these tokens (lstm-layer.h has none)
                throw Error(EESEN_ERR_INVALID, "dropout option " + t + " on a layer that is not a BiLstm");
            }
          if (!known) c.fail("unexpected token " + t);
        }
      }
      read_tensors(P.flat);
    }
    parsed.push_back(std::move(P));
  }
  if (parsed.empty()) throw Error(EESEN_ERR_IO, "model file " + path + " holds no layers");
  for (const ParsedLayer& P : parsed) {
    add_layer(P.kind, P.din, P.dout, P.coef, P.max_grad);
    for (int k = 0; k < 9; ++k) layers.back().drop[k] = P.drop[k];
    layers.back().nonparallel = P.nonparallel;
  }
  finalize();
  std::vector<float> all;
  for (const ParsedLayer& P : parsed) all.insert(all.end(), P.flat.begin(), P.flat.end());
  for (float v : all)  // Net::Check, net.cc:459-468
    if (!std::isfinite(v)) throw Error(EESEN_ERR_INVALID, "model holds NaN/Inf parameters");
  set_params(all.data(), (long)all.size());
  bool any_accu = false;
  for (const ParsedLayer& P : parsed) any_accu = any_accu || !P.accu.empty();
  if (any_accu) {
    std::vector<float> acc;
    for (const ParsedLayer& P : parsed) {
      if (!P.accu.empty()) acc.insert(acc.end(), P.accu.begin(), P.accu.end());
      else acc.insert(acc.end(), P.flat.size(), 0.f);
    }
    set_accu(acc.data(), (long)acc.size());
  }
  lr = 0.f;  // net.cc:294
}

namespace {
void put_token(std::ostream& os, const char* t) { os << t << " "; }
void put_int(std::ostream& os, bool binary, int32_t v) {
  if (binary) { os.put((char)4); os.write(reinterpret_cast<const char*>(&v), 4); }
  else os << v << " ";
}
void put_float(std::ostream& os, bool binary, float v) {
  if (binary) { os.put((char)4); os.write(reinterpret_cast<const char*>(&v), 4); }
  else { char s[32]; snprintf(s, sizeof s, "%.9g", v); os << s << " "; }
}
void put_bool(std::ostream& os, bool binary, bool v) { os << (v ? "T" : "F"); if (!binary) os << " "; }
void put_tensor(std::ostream& os, bool binary, const float* p, int rows, int cols) {
  const size_t n = (size_t)rows * (cols ? cols : 1);
  if (binary) {
    put_token(os, cols ? "FM" : "FV");
    put_int(os, true, rows);
    if (cols) put_int(os, true, cols);
    os.write(reinterpret_cast<const char*>(p), n * 4);
  } else if (cols) {
    os << " [";
    for (int r = 0; r < rows; ++r) {
      os << "\n  ";
      for (int c = 0; c < cols; ++c) { char s[32]; snprintf(s, sizeof s, "%.9g", p[(size_t)r * cols + c]); os << s << " "; }
    }
    os << "]\n";
  } else {
    os << " [ ";
    for (int r = 0; r < rows; ++r) { char s[32]; snprintf(s, sizeof s, "%.9g", p[r]); os << s << " "; }
    os << "]\n";
  }
}
}  // namespace

void Net::write(const std::string& path, bool binary) {
  EESEN_REQUIRE(finalized, EESEN_ERR_STATE, "net not finalized");
  std::vector<float> all(num_params());
  get_flat(params, all.data(), (long)all.size());
  for (float v : all)
    if (!std::isfinite(v)) throw Error(EESEN_ERR_INVALID, "refusing to write NaN/Inf parameters");  // net.cc:459-468
  std::ofstream os(path, std::ios::binary);
  if (!os) throw Error(EESEN_ERR_IO, "cannot open " + path + " for writing");
  if (binary) { os.put('\0'); os.put('B'); }
  put_token(os, "<Nnet>");
  if (!binary) os << "\n";
  std::vector<float> acc;
  if (accu_init) {  // adaBuffersInitialized (bilstm-layer.h:457-476): the accumulators travel with the model
    acc.resize(all.size());
    get_flat(accu, acc.data(), (long)acc.size());
  }
  const float* p = all.data();
  const float* pa = acc.empty() ? nullptr : acc.data();
  for (const Layer& L : layers) {
    const char* marker = L.kind == EESEN_LAYER_BILSTM_PARALLEL ? (L.nonparallel ? "<BiLstm>" : "<BiLstmParallel>")
                         : L.kind == EESEN_LAYER_LSTM_PARALLEL ? (L.nonparallel ? "<Lstm>" : "<LstmParallel>")
                         : L.kind == EESEN_LAYER_AFFINE        ? "<AffineTransform>"
                         : L.kind == EESEN_LAYER_SIGMOID       ? "<Sigmoid>"
                         : L.kind == EESEN_LAYER_TANH          ? "<Tanh>"
                                                               : "<Softmax>";
    put_token(os, marker);
    put_token(os, "<InputDim>");
    put_int(os, binary, L.din_f);    // the file's dimensions (Layer::din_f: an LSTM layer padded to a multiple of 4 cells is written unpadded)
    put_token(os, L.is_lstm() ? "<CellDim>" : "<OutputDim>");
    put_int(os, binary, L.dout_f);
    if (!binary) os << "\n";
    if (!L.trainable()) continue;
    put_token(os, "<LearnRateCoef>");
    put_float(os, binary, L.coef);
    put_token(os, "<MaxGrad>");
    put_float(os, binary, L.max_grad);
    if (L.kind == EESEN_LAYER_BILSTM_PARALLEL)  // bilstm-layer.h:435-455; the uni-LSTM writes none (lstm-layer.h:147-151)
      for (int k = 0; k < 9; ++k) {
        put_token(os, kDropoutTokens[k]);
        if (kDropoutIsFloat[k]) put_float(os, binary, L.drop[k]);
        else put_bool(os, binary, L.drop[k] != 0.f);
      }
    auto write_tensors = [&](const float*& q) {
      if (L.is_lstm()) {
        const int H = L.Hf;
        for (int d = 0; d < L.ndir; ++d) {
          put_tensor(os, binary, q, 4 * H, L.din_f); q += (size_t)4 * H * L.din_f;
          put_tensor(os, binary, q, 4 * H, H);     q += (size_t)4 * H * H;
          put_tensor(os, binary, q, 4 * H, 0);     q += 4 * H;
          for (int g = 0; g < 3; ++g) { put_tensor(os, binary, q, H, 0); q += H; }
        }
      } else {
        put_tensor(os, binary, q, L.dout_f, L.din_f); q += (size_t)L.dout_f * L.din_f;
        put_tensor(os, binary, q, L.dout_f, 0);       q += L.dout_f;
      }
    };
    if (pa) {
      // NB: the reference's uni-directional Lstm::WriteData writes the WEIGHTS under <LstmAccus> (lstm-layer.h:153-163, a
      // bug); this writer stores the real accumulators there, which is what its own ReadData expects.
      put_token(os, L.kind == EESEN_LAYER_BILSTM_PARALLEL ? "<BiLstmAccus>" : L.kind == EESEN_LAYER_LSTM_PARALLEL ? "<LstmAccus>" : "<AffineAccus>");
      write_tensors(pa);
    }
    write_tensors(p);
  }
  put_token(os, "</Nnet>");
  if (!binary) os << "\n";
  os.close();
  if (!os) throw Error(EESEN_ERR_IO, "write failure on " + path);
}

}  // namespace eesen
