// Hand-derived gradients of the angular force-field terms (device code; also compiles for the host so that
// tests/test_ff_grad_host.py can check every function against the CPU oracle without a GPU).
//
// Round 1 differentiated the angular terms (bends, stretch-bends, Wilson angles, torsions, inversions, chiral volumes)
// with forward-mode dual numbers: correct by construction, but a Dual<12> carries 13 doubles through every operation —
// 3-4x the arithmetic and most of the 250 VGPRs of the fused BFGS kernels.  Here every term is E = f(q) of ONE geometric
// primitive q (cosine of an angle, cosine of a dihedral, Wilson sine, chiral volume) or of a primitive and two bond
// lengths; the gradient is f'(q) times the primitive's gradient, which is a handful of cross products.  The energy forms
// stay in ff_terms.h (templated; instantiated with double); the dual-number gradients remain available as the test oracle
// of this file (-DNVMK_FF_DUAL_GRAD, and the C oracle derives the same primitives independently: oracle/oracle_ff.c).
//
// Conventions kept from the reference (src/forcefields/*_kernels_device.cuh, i.e. RDKit's contribs): a cosine clamped to
// +-1 has zero gradient; degenerate geometry (zero-length arm, collinear dihedral) contributes no gradient; acos / asin
// derivatives are dropped / capped at |q| = 1 exactly as ff_terms.h does; the chiral-volume "gradient" is half the
// derivative; the UFF inversion gradient carries the reference's sign for its C2 part.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NVMK_HD __host__ __device__ __forceinline__
#else
#define NVMK_HD inline
#endif

namespace nvmk {
namespace ffg {

constexpr double kPi      = 3.14159265358979323846;
constexpr double kRad2Deg = 180.0 / kPi;
constexpr double kDeg2Rad = kPi / 180.0;
constexpr double kMdyne   = 143.9325;

struct V3 {
  double x, y, z;
};
NVMK_HD V3     sub(const V3 a, const V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
NVMK_HD V3     add(const V3 a, const V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
NVMK_HD V3     scale(const V3 a, const double s) { return {a.x * s, a.y * s, a.z * s}; }
NVMK_HD V3     neg(const V3 a) { return {-a.x, -a.y, -a.z}; }
NVMK_HD double dot(const V3 a, const V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NVMK_HD V3     cross(const V3 a, const V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// ---- geometric primitives with their gradients ----------------------------------------------------------------------

// cos of the angle 1-2-3 (vertex 2).  false: a zero-length arm (no energy, no gradient).  A clamped cosine has g = 0.
NVMK_HD bool cos_angle_grad(const V3 p1, const V3 p2, const V3 p3, double& c, V3 (&g)[3]) {
  const V3     r1 = sub(p1, p2), r2 = sub(p3, p2);
  const double l1 = dot(r1, r1), l2 = dot(r2, r2);
  g[0] = g[1] = g[2] = V3{0.0, 0.0, 0.0};
  c                  = 0.0;
  if (!(l1 > 1.0e-16 && l2 > 1.0e-16)) return false;
  const double inv = 1.0 / sqrt(l1 * l2);
  const double cc  = dot(r1, r2) * inv;
  if (cc > 1.0) {
    c = 1.0;
    return true;
  }
  if (cc < -1.0) {
    c = -1.0;
    return true;
  }
  c    = cc;
  g[0] = sub(scale(r2, inv), scale(r1, cc / l1));
  g[2] = sub(scale(r1, inv), scale(r2, cc / l2));
  g[1] = neg(add(g[0], g[2]));
  return true;
}

// cos of the dihedral 1-2-3-4.  false: collinear (callers use cos = 0, no gradient).
NVMK_HD bool cos_dihedral_grad(const V3 p1, const V3 p2, const V3 p3, const V3 p4, double& c, V3 (&g)[4]) {
  const V3     r1 = sub(p1, p2), r2 = sub(p3, p2), r4 = sub(p4, p3);
  const V3     t1 = cross(r1, r2), t2 = cross(neg(r2), r4);
  const double a = dot(t1, t1), b = dot(t2, t2), d = a * b;
  g[0] = g[1] = g[2] = g[3] = V3{0.0, 0.0, 0.0};
  c                         = 0.0;
  if (!(d > 1.0e-16)) return false;
  const double inv = 1.0 / sqrt(d);
  const double cc  = dot(t1, t2) * inv;
  if (cc > 1.0) {
    c = 1.0;
    return true;
  }
  if (cc < -1.0) {
    c = -1.0;
    return true;
  }
  c          = cc;
  const V3 v = sub(scale(t2, inv), scale(t1, cc / a));  // dc / dt1
  const V3 w = sub(scale(t1, inv), scale(t2, cc / b));  // dc / dt2
  // t1 = r1 x r2, t2 = r4 x r2:  d(a x b) . v = da . (b x v) + db . (v x a)
  const V3 gr1 = cross(r2, v);
  const V3 gr2 = add(cross(v, r1), cross(w, r4));
  const V3 gr4 = cross(r2, w);
  g[0]         = gr1;
  g[1]         = neg(add(gr1, gr2));
  g[2]         = sub(gr2, gr4);
  g[3]         = gr4;
  return true;
}

// s = n . rJL / (|n| |rJL|), n = (p1 - p2) x (p3 - p2), rJL = p4 - p2: Wilson sine / cosY of the inversion.
// false when |n|^2 < nTol or rJL vanishes.
NVMK_HD bool wilson_grad(const V3 p1, const V3 p2, const V3 p3, const V3 p4, const double nTol, double& s, V3 (&g)[4]) {
  const V3     rI = sub(p1, p2), rK = sub(p3, p2), rL = sub(p4, p2);
  const V3     n  = cross(rI, rK);
  const double ln = dot(n, n), lL = dot(rL, rL);
  g[0] = g[1] = g[2] = g[3] = V3{0.0, 0.0, 0.0};
  s                         = 0.0;
  if (ln < nTol || lL < 1.0e-16) return false;
  const double inv = 1.0 / sqrt(ln * lL);
  const double ss  = dot(n, rL) * inv;
  if (ss > 1.0) {
    s = 1.0;
    return true;
  }
  if (ss < -1.0) {
    s = -1.0;
    return true;
  }
  s           = ss;
  const V3 dn = sub(scale(rL, inv), scale(n, ss / ln));
  const V3 dL = sub(scale(n, inv), scale(rL, ss / lL));
  const V3 gI = cross(rK, dn);
  const V3 gK = cross(dn, rI);
  g[0]        = gI;
  g[2]        = gK;
  g[3]        = dL;
  g[1]        = neg(add(add(gI, gK), dL));
  return true;
}

// d(acos c in degrees) / dc with the derivative dropped at |c| = 1 (ff_terms.h acos_)
NVMK_HD double dtheta_deg_dc(const double c) {
  const double s2 = 1.0 - c * c;
  return s2 > 1.0e-16 ? -kRad2Deg / sqrt(s2) : 0.0;
}

// ---- f'(q) of every angular term (energy forms: ff_terms.h) -------------------------------------------------------

// MMFF angle bend: dE/dc
NVMK_HD double mmff_angle_dE(const double c, const double theta0, const double ka, const bool isLinear) {
  if (isLinear) return kMdyne * ka;
  constexpr double cb  = -0.4 * kDeg2Rad;
  const double     dt  = kRad2Deg * acos(c) - theta0;
  const double     pre = 0.5 * kMdyne * kDeg2Rad * kDeg2Rad * ka;
  return pre * (2.0 * dt + 3.0 * cb * dt * dt) * dtheta_deg_dc(c);
}

// MMFF stretch-bend: the whole gradient (depends on the angle AND both bond lengths).  false: a zero-length arm.
NVMK_HD bool mmff_stretch_bend_grad(const V3 p1, const V3 p2, const V3 p3, const double theta0, const double r0ij, const double r0kj,
                                    const double kbaIJK, const double kbaKJI, V3 (&g)[3]) {
  const V3     r1 = sub(p1, p2), r2 = sub(p3, p2);
  const double d1 = sqrt(dot(r1, r1)), d2 = sqrt(dot(r2, r2));
  g[0] = g[1] = g[2] = V3{0.0, 0.0, 0.0};
  if (d1 < 1.0e-8 || d2 < 1.0e-8) return false;
  double c;
  V3     gc[3];
  cos_angle_grad(p1, p2, p3, c, gc);
  const double dt  = kRad2Deg * acos(c) - theta0;
  const double sb  = (d1 - r0ij) * kbaIJK + (d2 - r0kj) * kbaKJI;
  const double dth = 2.51210 * sb * dtheta_deg_dc(c);
  const double f1 = 2.51210 * dt * kbaIJK / d1, f2 = 2.51210 * dt * kbaKJI / d2;
  g[0] = add(scale(gc[0], dth), scale(r1, f1));
  g[2] = add(scale(gc[2], dth), scale(r2, f2));
  g[1] = sub(scale(gc[1], dth), add(scale(r1, f1), scale(r2, f2)));
  return true;
}

// MMFF Wilson out-of-plane: dE/ds (asin derivative capped like ff_terms.h asin_)
NVMK_HD double mmff_oop_dE(const double s, const double koop) {
  const double chi = kRad2Deg * asin(s);
  const double c2  = 1.0 - s * s;
  return 0.5 * kMdyne * kDeg2Rad * kDeg2Rad * koop * 2.0 * chi * kRad2Deg * (c2 > 1.0e-16 ? 1.0 / sqrt(c2) : 1.0e8);
}

NVMK_HD double mmff_torsion_dE(const double c, const double V1, const double V2, const double V3_) {
  return 0.5 * (V1 - 4.0 * V2 * c + V3_ * (12.0 * c * c - 3.0));
}

// ETK 6-term cosine series: dE/dc (derivatives of the Chebyshev polynomials)
NVMK_HD double torsion_m6_dE(const double c, const double* fc, const double* sg) {
  const double c2 = c * c, c3 = c2 * c, c4 = c3 * c, c5 = c4 * c;
  return fc[0] * sg[0] + fc[1] * sg[1] * (4.0 * c) + fc[2] * sg[2] * (12.0 * c2 - 3.0) + fc[3] * sg[3] * (32.0 * c3 - 16.0 * c) +
         fc[4] * sg[4] * (80.0 * c4 - 60.0 * c2 + 5.0) + fc[5] * sg[5] * (192.0 * c5 - 192.0 * c3 + 36.0 * c);
}

// Inversion k (C0 + C1 sinY + C2 cos2W) as a function of cosY: dE/dcosY.  `ok` false: sinY sits on its floor (no gradient).
// uffConvention: the reference's UFF gradient has the opposite sign on its C2 part (uff_kernels_device.cuh:497).
NVMK_HD double inversion_dE(const double cosY, const double C1, const double C2, const double k, const bool uffConvention, bool& ok) {
  const double sinYSq = 1.0 - cosY * cosY;
  ok                  = sinYSq >= 1.0e-16;
  if (!ok) return 0.0;
  const double sinY = sqrt(sinYSq);
  const double dsin = -cosY / sinY;  // d sinY / d cosY
  return uffConvention ? k * (C1 - 4.0 * C2 * sinY) * dsin : k * (C1 + 4.0 * C2 * sinY) * dsin;
}

// flat-bottom angle restraint in degrees, E = k (theta - bound)^2: dE/dc
NVMK_HD double angle_window_dE(const double c, const double minDeg, const double maxDeg, const double k) {
  const double theta = kRad2Deg * acos(c);
  double       diff  = 0.0;
  if (theta < minDeg) {
    diff = theta - minDeg;
  } else if (theta > maxDeg) {
    diff = theta - maxDeg;
  }
  return diff == 0.0 ? 0.0 : 2.0 * k * diff * dtheta_deg_dc(c);
}

// UFF angle bend: dE/dc (orders 0..4, plus the near-zero-angle correction, uff_kernels_device.cuh:78-170)
NVMK_HD double uff_angle_dE(const double c, const double theta0, const double k, const int order, const double C1, const double C2) {
  const double c2 = c * c;
  double       dE;
  if (order == 0) {
    dE = k * (C1 + 4.0 * C2 * c);
  } else {
    double df;
    switch (order) {
      case 1: df = -1.0; break;
      case 2: df = 4.0 * c; break;
      case 3: df = 12.0 * c2 - 3.0; break;
      case 4: df = 32.0 * c2 * c - 16.0 * c; break;
      default: df = 0.0; break;
    }
    dE = -(k / static_cast<double>(order * order)) * df;
    if (order < 5 && c > 0.8660) {
      const double s2 = 1.0 - c2;
      const double ex = exp(-20.0 * (acos(c) - theta0 + 0.25));
      dE += -20.0 * ex * (s2 > 1.0e-16 ? -1.0 / sqrt(s2) : 0.0);
    }
  }
  return dE;
}

// UFF Fourier torsion k/2 (1 - cosTerm cos(n phi)), n in {2, 3, 6}: dE/dc; other orders 0
NVMK_HD double uff_torsion_dE(const double c, const double k, const int order, const double cosTerm) {
  const double c2 = c * c, s2 = 1.0 - c2;
  double       dcn;
  switch (order) {
    case 2: dcn = 4.0 * c; break;
    case 3: dcn = 12.0 * c2 - 3.0; break;
    case 6: dcn = (-96.0 * s2 * s2 + 96.0 * s2 - 18.0) * (-2.0 * c); break;
    default: return 0.0;
  }
  return -0.5 * k * cosTerm * dcn;
}

// chiral volume (p1 - p4) . ((p2 - p4) x (p3 - p4)) with its gradient
NVMK_HD double chiral_volume_grad(const V3 p1, const V3 p2, const V3 p3, const V3 p4, V3 (&g)[4]) {
  const V3 a = sub(p1, p4), b = sub(p2, p4), c = sub(p3, p4);
  g[0]       = cross(b, c);
  g[1]       = cross(c, a);
  g[2]       = cross(a, b);
  g[3]       = neg(add(add(g[0], g[1]), g[2]));
  return dot(a, g[0]);
}

// ---- whole gradient of one term, accumulated through `acc(atom, V3)` ----------------------------------------------
// DIM = coordinate stride of `pos` (3, or 4 for the distance-geometry field).  The same functions run inside the fused BFGS
// kernels (acc = LDS atomics into the wave's gradient slab) and in the host check (acc = plain adds).

template <int DIM> NVMK_HD V3 load(const double* pos, const int a) { return {pos[a * DIM], pos[a * DIM + 1], pos[a * DIM + 2]}; }

template <int NA, class Acc> NVMK_HD void push_scaled(Acc& acc, const int* a, const V3 (&g)[NA], const double f) {
#pragma unroll
  for (int m = 0; m < NA; ++m) acc(a[m], scale(g[m], f));
}

// DG chiral volume term, weight w; RDKit's gradient is HALF the derivative of w (vol - bound)^2
template <int DIM, class Acc> NVMK_HD void grad_dg_chiral(const double* pos, const int* a, const double lo, const double hi, const double w, Acc& acc) {
  V3           g[4];
  const double vol = chiral_volume_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), load<DIM>(pos, a[3]), g);
  double       dv  = 0.0;
  if (vol < lo) {
    dv = vol - lo;
  } else if (vol > hi) {
    dv = vol - hi;
  }
  if (dv != 0.0) push_scaled<4>(acc, a, g, w * dv);
}

template <int DIM, class Acc> NVMK_HD void grad_etk_torsion(const double* pos, const int* a, const double* fc12, Acc& acc) {
  double c;
  V3     g[4];
  if (cos_dihedral_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), load<DIM>(pos, a[3]), c, g)) {
    push_scaled<4>(acc, a, g, torsion_m6_dE(c, fc12, fc12 + 6));
  }
}

// inversion at centre a[1]: par = C0, C1, C2, k (ETK order) — or k, C0, C1, C2 with uffConvention (UFF order)
template <int DIM, class Acc> NVMK_HD void grad_inversion(const double* pos, const int* a, const double C1, const double C2, const double k,
                                                          const bool uffConvention, Acc& acc) {
  const V3     p1 = load<DIM>(pos, a[0]), p2 = load<DIM>(pos, a[1]), p3 = load<DIM>(pos, a[2]), p4 = load<DIM>(pos, a[3]);
  const V3     rI = sub(p1, p2), rK = sub(p3, p2), rL = sub(p4, p2);
  const double lI = dot(rI, rI), lK = dot(rK, rK), lL = dot(rL, rL);
  if (lI < 1.0e-16 || lK < 1.0e-16 || lL < 1.0e-16) return;
  double cosY;
  V3     g[4];
  if (!wilson_grad(p1, p2, p3, p4, 1.0e-16 * lI * lK, cosY, g)) return;
  bool         ok;
  const double dE = inversion_dE(cosY, C1, C2, k, uffConvention, ok);
  if (ok) push_scaled<4>(acc, a, g, dE);
}

template <int DIM, class Acc> NVMK_HD void grad_angle_window(const double* pos, const int* a, const double lo, const double hi, const double k, Acc& acc) {
  double c;
  V3     g[3];
  if (cos_angle_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), c, g)) {
    const double dE = angle_window_dE(c, lo, hi, k);
    if (dE != 0.0) push_scaled<3>(acc, a, g, dE);
  }
}

template <int DIM, class Acc> NVMK_HD void grad_mmff_angle(const double* pos, const int* a, const double theta0, const double ka, const bool lin, Acc& acc) {
  double c;
  V3     g[3];
  if (cos_angle_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), c, g)) {
    push_scaled<3>(acc, a, g, mmff_angle_dE(c, theta0, ka, lin));
  }
}

template <int DIM, class Acc> NVMK_HD void grad_mmff_stretch_bend(const double* pos, const int* a, const double* p, Acc& acc) {
  V3 g[3];
  if (mmff_stretch_bend_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), p[0], p[1], p[2], p[3], p[4], g)) {
    push_scaled<3>(acc, a, g, 1.0);
  }
}

template <int DIM, class Acc> NVMK_HD void grad_mmff_oop(const double* pos, const int* a, const double koop, Acc& acc) {
  double s;
  V3     g[4];
  if (wilson_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), load<DIM>(pos, a[3]), 1.0e-16, s, g)) {
    push_scaled<4>(acc, a, g, mmff_oop_dE(s, koop));
  }
}

template <int DIM, class Acc> NVMK_HD void grad_mmff_torsion(const double* pos, const int* a, const double V1, const double V2, const double V3_, Acc& acc) {
  double c;
  V3     g[4];
  if (cos_dihedral_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), load<DIM>(pos, a[3]), c, g)) {
    push_scaled<4>(acc, a, g, mmff_torsion_dE(c, V1, V2, V3_));
  }
}

template <int DIM, class Acc> NVMK_HD void grad_uff_angle(const double* pos, const int* a, const double* p, Acc& acc) {
  double c;
  V3     g[3];
  if (cos_angle_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), c, g)) {
    push_scaled<3>(acc, a, g, uff_angle_dE(c, p[0], p[1], static_cast<int>(p[2]), p[4], p[5]));
  }
}

template <int DIM, class Acc> NVMK_HD void grad_uff_torsion(const double* pos, const int* a, const double k, const int order, const double cosTerm, Acc& acc) {
  double c;
  V3     g[4];
  if (cos_dihedral_grad(load<DIM>(pos, a[0]), load<DIM>(pos, a[1]), load<DIM>(pos, a[2]), load<DIM>(pos, a[3]), c, g)) {
    const double dE = uff_torsion_dE(c, k, order, cosTerm);
    if (dE != 0.0) push_scaled<4>(acc, a, g, dE);
  }
}

}  // namespace ffg
}  // namespace nvmk
