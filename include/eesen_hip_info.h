/* eesen_hip_info.h -- the strings of Net::Info() / Net::InfoGradient() over the C-ABI of eesen_hip.h.
 *
 * The reference's trainer logs both when training finishes (/root/reference/src/netbin/train-ctc-parallel.cc:236-240):
 *   Net::Info          src/net/net.cc:336-354   topology line per layer + the layer's Info()
 *   Net::InfoGradient  src/net/net.cc:356-366   the layer's InfoGradient()
 * whose per-layer parts list MomentStatistics (src/net/utils-functions.h:50-82: min, max, mean, variance, skewness, kurtosis)
 * of every tensor: bilstm-layer.h:496-560, lstm-layer.h:175-196, affine-trans-layer.h:145-159.  The figures come from
 * eesen_net_tensor_moments (device-side reductions); this header only arranges them the reference's way.  Plain C++ over the
 * C-ABI: used by the seam (eesen_hip_net.h) and by the native trainer (eesen_amd/csrc/tools/train_ctc_parallel.cc).
 */
#ifndef EESEN_HIP_INFO_H_
#define EESEN_HIP_INFO_H_

#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "eesen_hip.h"

namespace eesen_hip {

inline void InfoCheck(int rc) {
  if (rc != EESEN_OK) throw std::runtime_error(std::string("eesen_hip: ") + eesen_last_error());
}

/* MomentStatistics' own rendering (utils-functions.h:74-81) */
inline std::string MomentString(const double* m) {
  std::ostringstream o;
  o << " ( min " << m[0] << ", max " << m[1] << ", mean " << m[2] << ", variance " << m[3] << ", skewness " << m[4] << ", kurtosis "
    << m[5] << " ) ";
  return o.str();
}

/* which = 0: Net::Info, which = 1: Net::InfoGradient (the adaptive accumulators follow the momentum buffers when
 * with_accumulators is set, as the reference does once adaBuffersInitialized) */
inline std::string NetInfo(eesen_net_t* net, int which, bool with_accumulators = false) {
  static const char* const kLstm[6] = {"wei_gifo_x", "wei_gifo_m", "bias", "phole_i_c", "phole_f_c", "phole_o_c"};
  std::ostringstream os;
  int n = 0, din = 0, dout = 0;
  long np = 0;
  InfoCheck(eesen_net_num_layers(net, &n));
  if (which == 0) {
    InfoCheck(eesen_net_input_dim(net, &din));
    InfoCheck(eesen_net_output_dim(net, &dout));
    InfoCheck(eesen_net_num_params(net, &np));
    os << "num-layers " << n << "\ninput-dim " << din << "\noutput-dim " << dout << "\nnumber-of-parameters "
       << static_cast<float>(np) / 1e6 << " millions\n";
  } else {
    os << "### Gradient stats :\n";
  }
  for (int i = 0; i < n; ++i) {
    int kind = 0, di = 0, dd = 0;
    char marker[64];
    InfoCheck(eesen_net_layer_info(net, i, &kind, &di, &dd, NULL, NULL));
    InfoCheck(eesen_net_layer_marker(net, i, marker, (int)sizeof(marker)));
    if (which == 0) os << "layer " << i + 1 << " : " << marker << ", input-dim " << di << ", output-dim " << dd << ", ";
    else os << "Layer " << i + 1 << " : " << marker << ", ";
    int nt = 0;
    InfoCheck(eesen_net_tensor_moments(net, which, i, NULL, 0, &nt));
    std::vector<double> m((size_t)(nt > 0 ? nt : 1) * 6), a;
    if (nt) InfoCheck(eesen_net_tensor_moments(net, which, i, m.data(), nt, &nt));
    if (nt && which == 1 && with_accumulators) {
      a.resize((size_t)nt * 6);
      InfoCheck(eesen_net_tensor_moments(net, 2, i, a.data(), nt, &nt));
    }
    const bool bi = kind == EESEN_LAYER_BILSTM_PARALLEL, lstm = bi || kind == EESEN_LAYER_LSTM_PARALLEL;
    if (lstm) {
      os << "    ";
      for (int t = 0; t < nt; ++t)
        os << "\n  " << kLstm[t % 6] << (bi ? (t < 6 ? "_fw_" : "_bw_") : "_") << (which ? "corr_" : "") << "  " << MomentString(&m[(size_t)t * 6]);
      if (!a.empty() && bi)  /* bilstm-layer.h:515-531 builds these lines (lstm-layer.h has none) */
        for (int t = 0; t < nt; ++t)
          os << "\n  " << kLstm[t % 6] << (t < 6 ? "_fw_" : "_bw_") << "corr_accu  " << MomentString(&a[(size_t)t * 6]);
    } else if (kind == EESEN_LAYER_AFFINE && nt == 2) {
      os << "\n  " << (which ? "linearity_corr_" : "linearity") << MomentString(&m[0]) << "\n  " << (which ? "bias_corr_" : "bias")
         << MomentString(&m[6]);
      if (!a.empty()) os << "\n  linearity_grad_accu" << MomentString(&a[0]) << "\n  bias_grad_accu" << MomentString(&a[6]);
    }
    os << "\n";
  }
  return os.str();
}

}  /* namespace eesen_hip */
#endif /* EESEN_HIP_INFO_H_ */
