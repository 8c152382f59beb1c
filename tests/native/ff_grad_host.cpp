// Host build of nvmolkit_amd/csrc/ff_grad.h for tests/test_ff_grad_host.py (no GPU needed): the gradient of every angular
// term group through exactly the functions the fused BFGS kernels call, with a plain-add accumulator.  Test infrastructure.
#include "../../nvmolkit_amd/csrc/ff_grad.h"

using namespace nvmk::ffg;

namespace {
template <int DIM> struct Adder {
  double* g;
  void    operator()(const int a, const V3 f) {
    g[a * DIM] += f.x;
    g[a * DIM + 1] += f.y;
    g[a * DIM + 2] += f.z;
  }
};
}  // namespace

// kind codes of the checks: one per angular term group
extern "C" void chk_term_gradient(int code, int dim, const double* pos, const int* idx, int n_terms, int n_idx, const double* par,
                                  int n_par, double w0, double* grad) {
  for (int t = 0; t < n_terms; ++t) {
    const int*    a = idx + static_cast<long>(n_idx) * t;
    const double* p = par + static_cast<long>(n_par) * t;
    if (dim == 4) {
      Adder<4> acc{grad};
      if (code == 0) grad_dg_chiral<4>(pos, a, p[0], p[1], w0, acc);
    } else {
      Adder<3> acc{grad};
      switch (code) {
        case 1: grad_etk_torsion<3>(pos, a, p, acc); break;
        case 2: grad_inversion<3>(pos, a, p[1], p[2], p[3], false, acc); break;            // ETK: C0, C1, C2, k
        case 3: grad_angle_window<3>(pos, a, p[0], p[1], 1.0, acc); break;                 // ETK 1-3 angle, k = 1
        case 4: grad_mmff_angle<3>(pos, a, p[0], p[1], p[2] != 0.0, acc); break;
        case 5: grad_mmff_stretch_bend<3>(pos, a, p, acc); break;
        case 6: grad_mmff_oop<3>(pos, a, p[0], acc); break;
        case 7: grad_mmff_torsion<3>(pos, a, p[0], p[1], p[2], acc); break;
        case 8: grad_uff_angle<3>(pos, a, p, acc); break;
        case 9: grad_uff_torsion<3>(pos, a, p[0], static_cast<int>(p[1]), p[2], acc); break;
        case 10: grad_inversion<3>(pos, a, p[2], p[3], p[0], true, acc); break;            // UFF: k, C0, C1, C2
        default: break;
      }
    }
  }
}
