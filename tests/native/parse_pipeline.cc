// Test harness (CPU only): prints what eesen_amd/csrc/tools/feat_pipeline.h makes of the feature rspecifiers on its command line,
// one line per argument: `source|cmvn|utt2spk|norm_vars|kind:a:b,kind:a:b,...` or `NONE`.  tests/test_frontend.py compares
// it with eesen_amd/frontend.py (the two hosts must recognise exactly the same pipelines).
#include <cstdio>
#include "../../eesen_amd/csrc/tools/feat_pipeline.h"
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    ktab::Pipeline p;
    if (!ktab::parse_feature_pipeline(argv[i], &p)) { std::printf("NONE\n"); continue; }
    std::printf("%s|%s|%s|%d|", p.source.c_str(), p.cmvn.c_str(), p.utt2spk.c_str(), p.norm_vars ? 1 : 0);
    for (size_t k = 0; k < p.stages.size(); ++k) std::printf("%s%d:%d:%d", k ? "," : "", p.stages[k].kind, p.stages[k].a, p.stages[k].b);
    std::printf("\n");
  }
  return 0;
}
