// seam2_check.cc -- TEST INFRASTRUCTURE: Seam 2 compiled.  The reference's OWN eesen::Net (src/net/net.cc, layer.cc and the layer
// headers, compiled unmodified in CPU mode by this Makefile) runs the same model twice: once with its own BiLstmParallel layers
// (the reference arithmetic, on the host) and once with every BiLstmParallel swapped for include/eesen_hip_layer.h's
// HipBiLstmParallel, whose PropagateFnc / BackpropagateFnc / Update run in libeesen_hip.so on the MI355X.  Same random input,
// same random out_diff, two trainer steps with momentum and <MaxGrad>: layer-chain outputs, in_diff and the updated
// parameters of the two nets must agree within 1e-4 (max-norm relative).  Needs a GPU; tests/test_gpu_cli.py runs it.
//
//   seam2_check <model file> <S> <T> [steps]      exit 0 = agree; prints one line per quantity
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "net/net.h"
#include "eesen_hip_layer.h"

using namespace eesen;

static double rel(const MatrixBase<BaseFloat>& a, const MatrixBase<BaseFloat>& b) {
  double num = 0, den = 1e-30;
  for (int32 r = 0; r < a.NumRows(); ++r)
    for (int32 c = 0; c < a.NumCols(); ++c) {
      num = std::max(num, (double)std::fabs(a(r, c) - b(r, c)));
      den = std::max(den, (double)std::fabs(b(r, c)));
    }
  return num / den;
}
static double relv(const VectorBase<BaseFloat>& a, const VectorBase<BaseFloat>& b) {
  double num = 0, den = 1e-30;
  for (int32 i = 0; i < a.Dim(); ++i) { num = std::max(num, (double)std::fabs(a(i) - b(i))); den = std::max(den, (double)std::fabs(b(i))); }
  return num / den;
}

int main(int argc, char** argv) {
  try {
    if (argc < 4) { std::cerr << "usage: seam2_check <model> <S> <T> [steps]\n"; return 2; }
    const std::string model = argv[1];
    const int S = std::atoi(argv[2]), T = std::atoi(argv[3]), steps = argc > 4 ? std::atoi(argv[4]) : 2;
    Net ref, hip;
    ref.Read(model);
    hip.Read(model);
    const int swapped = HipBiLstmParallel::Adopt(&hip);
    std::printf("layers %d, BiLstmParallel layers running in libeesen_hip.so: %d\n", hip.NumLayers(), swapped);
    if (swapped == 0) return 3;
    NetTrainOptions opts;
    opts.learn_rate = 0.05; opts.momentum = 0.9;
    ref.SetTrainOptions(opts); hip.SetTrainOptions(opts);
    ref.SetUpdateAlgorithm("SGD"); hip.SetUpdateAlgorithm("SGD");
    ref.SetTrainMode(); hip.SetTrainMode();
    std::vector<int> lens(S, T);   // equal lengths: the reference computes on padding in its forward direction (garbage nobody reads)
    double worst = 0;
    std::srand(777);
    for (int step = 0; step < steps; ++step) {
      Matrix<BaseFloat> in(T * S, ref.InputDim()), od(T * S, ref.OutputDim());
      for (int32 r = 0; r < in.NumRows(); ++r) for (int32 c = 0; c < in.NumCols(); ++c) in(r, c) = (std::rand() / (float)RAND_MAX - 0.5f) * 2.f;
      for (int32 r = 0; r < od.NumRows(); ++r) for (int32 c = 0; c < od.NumCols(); ++c) od(r, c) = (std::rand() / (float)RAND_MAX - 0.5f) * 0.02f;
      CuMatrix<BaseFloat> cin(in), cod(od), out_r, out_h, id_r, id_h;
      ref.SetSeqLengths(lens); hip.SetSeqLengths(lens);
      ref.Propagate(cin, &out_r); hip.Propagate(cin, &out_h);
      ref.Backpropagate(cod, &id_r); hip.Backpropagate(cod, &id_h);
      Matrix<BaseFloat> a(out_r), b(out_h), c(id_r), d(id_h);
      Vector<BaseFloat> pr, ph;
      ref.GetParams(&pr); hip.GetParams(&ph);
      const double e_out = rel(b, a), e_id = rel(d, c), e_par = relv(ph, pr);
      std::printf("step %d: net_out %.3e  in_diff %.3e  parameters after the update %.3e\n", step, e_out, e_id, e_par);
      worst = std::max(worst, std::max(e_out, std::max(e_id, e_par)));
    }
    // the swapped net writes a model the reference reads back (WriteData is the reference's own)
    hip.Write(model + ".seam2", true);
    Net back; back.Read(model + ".seam2");
    Vector<BaseFloat> p1, p2; hip.GetParams(&p1); back.GetParams(&p2);
    std::printf("model written by the swapped net and read back: %.3e\n", relv(p2, p1));
    std::remove((model + ".seam2").c_str());
    std::printf("SEAM2 %s (worst %.3e)\n", worst < 1e-4 ? "OK" : "MISMATCH", worst);
    return worst < 1e-4 ? 0 : 1;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 255;
  }
}
