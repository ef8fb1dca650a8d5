// Test harness (CPU only) for eesen_amd/csrc/tools/kaldi_tables.h: `read_tables feats <rspecifier>` prints `key rows cols` per
// entry of a feature table, `read_tables labels <rspecifier>` prints `key n` per label vector; exit code 3 + the message on
// stderr when the reader throws (e.g. a table pipe that ended with a non-zero status).
#include <cstdio>
#include <cstring>
#include "../../eesen_amd/csrc/tools/kaldi_tables.h"
int main(int argc, char** argv) {
  if (argc != 3) return 1;
  try {
    if (!std::strcmp(argv[1], "feats")) {
      for (ktab::FeatureReader r(argv[2]); !r.Done(); r.Next()) std::printf("%s %d %d\n", r.Key().c_str(), r.Value().rows, r.Value().cols);
    } else {
      for (const auto& kv : ktab::read_targets(argv[2])) std::printf("%s %zu\n", kv.first.c_str(), kv.second.size());
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
