// Force-field term arithmetic for the batched minimisers (device code, fp64 throughout).
//
// Functional forms restated from the reference's device headers (which are themselves ports of RDKit's
// ForceField contribs):
//   DG   : src/forcefields/dist_geom_kernels_device.cuh:37-231   (distance violation, chiral volume, 4th dimension)
//   ETK  : src/forcefields/dist_geom_kernels_device.cuh:237-830  (6-term cosine torsion, inversion, flat-bottom
//          distance and angle constraints)
//   MMFF : src/forcefields/mmff_kernels_device.cuh:28-660        (MMFF94 stretch, bend, stretch-bend, Wilson oop,
//          torsion, buffered 14-7 vdW, buffered Coulomb)
//   UFF  : src/forcefields/uff_kernels_device.cuh:37-580         (harmonic stretch, cosine-harmonic / Fourier bend
//          with the near-zero-angle correction, Fourier torsion, inversion, thresholded 12-6)
//
// Implementation is this build's own:
//   * the reference mixes float and double inside a term; here everything is double;
//   * radial pair terms (the O(N^2) majority) use hand-derived dE/dr;
//   * angular terms are written ONCE, templated on the scalar type, and differentiated with forward-mode
//     dual numbers, so energy and gradient cannot drift apart and the reference's 1/sin(phi) special cases
//     disappear (the cosine-series torsions are polynomials in cos(phi)).  This also uses force constant 5 for
//     the 6th ETK torsion term where the reference's gradient reuses index 4;
//   * two RDKit conventions are kept because coordinates must match RDKit's minimiser: the chiral-volume and
//     fourth-dimension "gradients" are HALF the true derivative (dist_geom_kernels_device.cuh:172-176, :229).
#pragma once

#include <hip/hip_runtime.h>

namespace nvmk {
namespace ff {

constexpr double kPi      = 3.14159265358979323846;
constexpr double kRad2Deg = 180.0 / kPi;
constexpr double kDeg2Rad = kPi / 180.0;

// ---- forward-mode dual numbers ------------------------------------------------------------------
template <int NP> struct Dual {
  double v;
  double d[NP];
};

template <int NP> __device__ __forceinline__ Dual<NP> make_const(const double c) {
  Dual<NP> r;
  r.v = c;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = 0.0;
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> make_var(const double x, const int slot) {
  Dual<NP> r = make_const<NP>(x);
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = (k == slot) ? 1.0 : 0.0;
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator+(const Dual<NP>& a, const Dual<NP>& b) {
  Dual<NP> r;
  r.v = a.v + b.v;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = a.d[k] + b.d[k];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator-(const Dual<NP>& a, const Dual<NP>& b) {
  Dual<NP> r;
  r.v = a.v - b.v;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = a.d[k] - b.d[k];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator-(const Dual<NP>& a) {
  Dual<NP> r;
  r.v = -a.v;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = -a.d[k];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(const Dual<NP>& a, const Dual<NP>& b) {
  Dual<NP> r;
  r.v = a.v * b.v;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(const double a, const Dual<NP>& b) {
  Dual<NP> r;
  r.v = a * b.v;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = a * b.d[k];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(const Dual<NP>& a, const double b) { return b * a; }
template <int NP> __device__ __forceinline__ Dual<NP> operator+(const Dual<NP>& a, const double b) {
  Dual<NP> r = a;
  r.v += b;
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator+(const double a, const Dual<NP>& b) { return b + a; }
template <int NP> __device__ __forceinline__ Dual<NP> operator-(const Dual<NP>& a, const double b) { return a + (-b); }
template <int NP> __device__ __forceinline__ Dual<NP> operator-(const double a, const Dual<NP>& b) { return (-b) + a; }
template <int NP> __device__ __forceinline__ Dual<NP> operator/(const Dual<NP>& a, const Dual<NP>& b) {
  const double inv = 1.0 / b.v;
  Dual<NP>     r;
  r.v = a.v * inv;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * inv;
  return r;
}
// f(a) with derivative fp: chain rule
template <int NP> __device__ __forceinline__ Dual<NP> chain(const Dual<NP>& a, const double f, const double fp) {
  Dual<NP> r;
  r.v = f;
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = fp * a.d[k];
  return r;
}
__device__ __forceinline__ double sqrt_(const double x) { return sqrt(x); }
template <int NP> __device__ __forceinline__ Dual<NP> sqrt_(const Dual<NP>& a) {
  const double s = sqrt(a.v);
  return chain(a, s, s > 0.0 ? 0.5 / s : 0.0);
}
// acos / asin with the derivative capped near |x| = 1 (the reference drops or caps these gradients,
// mmff_kernels_device.cuh:346-349, :46-47)
__device__ __forceinline__ double acos_(const double x) { return acos(x); }
template <int NP> __device__ __forceinline__ Dual<NP> acos_(const Dual<NP>& a) {
  const double s2 = 1.0 - a.v * a.v;
  return chain(a, acos(a.v), s2 > 1.0e-16 ? -1.0 / sqrt(s2) : 0.0);
}
__device__ __forceinline__ double asin_(const double x) { return asin(x); }
template <int NP> __device__ __forceinline__ Dual<NP> asin_(const Dual<NP>& a) {
  const double c2 = 1.0 - a.v * a.v;
  return chain(a, asin(a.v), c2 > 1.0e-16 ? 1.0 / sqrt(c2) : 1.0e8);
}
__device__ __forceinline__ double value(const double x) { return x; }
template <int NP> __device__ __forceinline__ double value(const Dual<NP>& a) { return a.v; }
__device__ __forceinline__ double clamp_unit(const double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); }
template <int NP> __device__ __forceinline__ Dual<NP> clamp_unit(const Dual<NP>& a) {
  if (a.v > 1.0) return make_const<NP>(1.0);
  if (a.v < -1.0) return make_const<NP>(-1.0);
  return a;
}

template <typename T> struct Vec3 {
  T x, y, z;
};
template <typename T> __device__ __forceinline__ Vec3<T> operator-(const Vec3<T>& a, const Vec3<T>& b) {
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename T> __device__ __forceinline__ T dot(const Vec3<T>& a, const Vec3<T>& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename T> __device__ __forceinline__ Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Load atom `m` (0-based slot inside the term) at local atom index `atom`; DIM = coordinate stride.
template <typename T, int DIM> struct Loader;
template <int DIM> struct Loader<double, DIM> {
  __device__ static __forceinline__ Vec3<double> get(const double* pos, const int atom, const int) {
    return {pos[atom * DIM + 0], pos[atom * DIM + 1], pos[atom * DIM + 2]};
  }
};
template <int NP, int DIM> struct Loader<Dual<NP>, DIM> {
  __device__ static __forceinline__ Vec3<Dual<NP>> get(const double* pos, const int atom, const int m) {
    return {make_var<NP>(pos[atom * DIM + 0], 3 * m + 0), make_var<NP>(pos[atom * DIM + 1], 3 * m + 1),
            make_var<NP>(pos[atom * DIM + 2], 3 * m + 2)};
  }
};

// ---- angular primitives (templated: double -> value, Dual -> value + gradient) ------------------

// cos of the angle 1-2-3 (vertex 2); degenerate (zero-length arm) -> returns `ok = false`.
template <typename T> __device__ __forceinline__ T cos_angle(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, bool& ok) {
  const Vec3<T> r1 = p1 - p2, r2 = p3 - p2;
  const T       l1 = dot(r1, r1), l2 = dot(r2, r2);
  ok                = value(l1) > 1.0e-16 && value(l2) > 1.0e-16;
  if (!ok) return l1 * 0.0;
  return clamp_unit(dot(r1, r2) / sqrt_(l1 * l2));
}

// cos of the dihedral 1-2-3-4; degenerate (collinear) -> ok = false.
template <typename T>
__device__ __forceinline__ T cos_dihedral(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const Vec3<T>& p4, bool& ok) {
  const Vec3<T> r1 = p1 - p2, r2 = p3 - p2, r4 = p4 - p3;
  const Vec3<T> t1 = cross(r1, r2);
  const Vec3<T> t2 = cross(Vec3<T>{-r2.x, -r2.y, -r2.z}, r4);
  const T       d  = dot(t1, t1) * dot(t2, t2);
  ok               = value(d) > 1.0e-16;
  if (!ok) return d * 0.0;
  return clamp_unit(dot(t1, t2) / sqrt_(d));
}

// ---- DG terms (dist_geom_kernels_device.cuh:37-231) ---------------------------------------------

// distance violation in DIM dimensions: E and dE/d(d^2)
__device__ __forceinline__ void dist_violation(const double d2, const double lb2, const double ub2, const double w, double& e,
                                               double& dE_dd2) {
  e      = 0.0;
  dE_dd2 = 0.0;
  if (d2 > ub2) {
    const double val = d2 / ub2 - 1.0;
    e                = w * val * val;
    dE_dd2           = 2.0 * w * val / ub2;
  } else if (d2 < lb2) {
    const double s   = lb2 + d2;
    const double val = 2.0 * lb2 / s - 1.0;
    e                = w * val * val;
    dE_dd2           = -4.0 * w * val * lb2 / (s * s);
  }
}

template <typename T> __device__ __forceinline__ T chiral_volume(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const Vec3<T>& p4) {
  return dot(p1 - p4, cross(p2 - p4, p3 - p4));
}
template <typename T> __device__ __forceinline__ T chiral_violation(const T vol, const double lb, const double ub, const double w) {
  if (value(vol) < lb) return w * (vol - lb) * (vol - lb);
  if (value(vol) > ub) return w * (vol - ub) * (vol - ub);
  return vol * 0.0;
}

// ---- ETK terms (dist_geom_kernels_device.cuh:237-830) -------------------------------------------

// 6-term cosine series: sum_k fc[k] (1 + sign[k] cos((k+1) phi)), Chebyshev polynomials of c = cos(phi)
template <typename T> __device__ __forceinline__ T torsion_m6(const T c, const double* fc, const double* sg) {
  const T c2 = c * c, c3 = c2 * c, c4 = c3 * c, c5 = c4 * c, c6 = c5 * c;
  const T k1 = c;
  const T k2 = 2.0 * c2 - 1.0;
  const T k3 = 4.0 * c3 - 3.0 * c;
  const T k4 = 8.0 * c4 - 8.0 * c2 + 1.0;
  const T k5 = 16.0 * c5 - 20.0 * c3 + 5.0 * c;
  const T k6 = 32.0 * c6 - 48.0 * c4 + 18.0 * c2 - 1.0;
  return fc[0] * (1.0 + sg[0] * k1) + fc[1] * (1.0 + sg[1] * k2) + fc[2] * (1.0 + sg[2] * k3) + fc[3] * (1.0 + sg[3] * k4) +
         fc[4] * (1.0 + sg[4] * k5) + fc[5] * (1.0 + sg[5] * k6);
}

// UFF-style inversion at centre 2 with arms 1, 3 and apex 4: E = k (C0 + C1 sinY + C2 cos2W),
// cosY = n . rJL / |rJL|, n = normal of plane (1, 2, 3); cos2W = 2 sinY^2 - 1.
template <typename T>
__device__ __forceinline__ T inversion(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const Vec3<T>& p4, const double C0,
                                       const double C1, const double C2, const double k) {
  const Vec3<T> rJI = p1 - p2, rJK = p3 - p2, rJL = p4 - p2;
  const T       lI = dot(rJI, rJI), lK = dot(rJK, rJK), lL = dot(rJL, rJL);
  // degenerate geometry: the reference treats cosY as 0 (sinY = 1, cos2W = 1), constant energy, no gradient (:300-321)
  if (value(lI) < 1.0e-16 || value(lK) < 1.0e-16 || value(lL) < 1.0e-16) return lI * 0.0 + k * (C0 + C1 + C2);
  const Vec3<T> n  = cross(rJI, rJK);
  const T       ln = dot(n, n);
  if (value(ln) < 1.0e-16 * value(lI) * value(lK)) return lI * 0.0 + k * (C0 + C1 + C2);
  const T cosY   = clamp_unit(dot(n, rJL) / sqrt_(ln * lL));
  T       sinYSq = 1.0 - cosY * cosY;
  if (value(sinYSq) < 1.0e-16) sinYSq = sinYSq * 0.0 + 1.0e-16;  // reference clamps sinY at 1e-8 (:646)
  const T sinY = sqrt_(sinYSq);
  return k * (C0 + C1 * sinY + C2 * (2.0 * sinYSq - 1.0));
}

// flat-bottom distance restraint: E = k/2 (d - bound)^2 outside [minLen, maxLen]; returns E and dE/dd
__device__ __forceinline__ void dist_constraint(const double d, const double minLen, const double maxLen, const double k, double& e,
                                                double& dE_dd) {
  double diff = 0.0;
  if (d < minLen) {
    diff = d - minLen;
  } else if (d > maxLen) {
    diff = d - maxLen;
  }
  e     = 0.5 * k * diff * diff;
  dE_dd = k * diff;
}

// flat-bottom angle restraint in degrees: E = k (theta - bound)^2 outside [minAngle, maxAngle]
template <typename T>
__device__ __forceinline__ T angle_constraint(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const double minAngle,
                                              const double maxAngle, const double k) {
  bool    ok;
  const T c = cos_angle(p1, p2, p3, ok);
  if (!ok) return c * 0.0;
  const T theta = kRad2Deg * acos_(c);
  if (value(theta) < minAngle) return k * (theta - minAngle) * (theta - minAngle);
  if (value(theta) > maxAngle) return k * (theta - maxAngle) * (theta - maxAngle);
  return theta * 0.0;
}

// ---- MMFF94 terms (mmff_kernels_device.cuh:28-660) ----------------------------------------------

constexpr double kMdyneA2Kcal = 143.9325;

// bond stretch: E(r) and dE/dr
__device__ __forceinline__ void mmff_bond(const double r, const double r0, const double kb, double& e, double& dE_dr) {
  constexpr double cs  = -2.0;
  constexpr double cs2 = 7.0 / 12.0 * cs * cs;
  const double     dr  = r - r0;
  e                    = 0.5 * kMdyneA2Kcal * kb * dr * dr * (1.0 + cs * dr + cs2 * dr * dr);
  dE_dr                = kMdyneA2Kcal * kb * dr * (1.0 + 1.5 * cs * dr + 2.0 * cs2 * dr * dr);
}

template <typename T>
__device__ __forceinline__ T mmff_angle(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const double theta0, const double ka,
                                        const bool isLinear) {
  bool    ok;
  const T c = cos_angle(p1, p2, p3, ok);
  if (!ok) return c * 0.0;
  if (isLinear) return kMdyneA2Kcal * ka * (1.0 + c);
  constexpr double cb = -0.4 * kDeg2Rad;
  const T          dt = kRad2Deg * acos_(c) - theta0;
  return 0.5 * kMdyneA2Kcal * kDeg2Rad * kDeg2Rad * ka * dt * dt * (1.0 + cb * dt);
}

template <typename T>
__device__ __forceinline__ T mmff_stretch_bend(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const double theta0,
                                               const double r0ij, const double r0kj, const double kbaIJK, const double kbaKJI) {
  const Vec3<T> r1 = p1 - p2, r2 = p3 - p2;
  const T       d1 = sqrt_(dot(r1, r1)), d2 = sqrt_(dot(r2, r2));
  if (value(d1) < 1.0e-8 || value(d2) < 1.0e-8) return d1 * 0.0;
  const T c  = clamp_unit(dot(r1, r2) / (d1 * d2));
  const T dt = kRad2Deg * acos_(c) - theta0;
  return 2.51210 * dt * ((d1 - r0ij) * kbaIJK + (d2 - r0kj) * kbaKJI);
}

// Wilson out-of-plane angle of arm 4 at centre 2 against plane (1, 2, 3)
template <typename T>
__device__ __forceinline__ T mmff_oop(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const Vec3<T>& p4, const double koop) {
  const Vec3<T> rJI = p1 - p2, rJK = p3 - p2, rJL = p4 - p2;
  const Vec3<T> n   = cross(rJI, rJK);
  const T       ln  = dot(n, n);
  const T       lL  = dot(rJL, rJL);
  if (value(ln) < 1.0e-16 || value(lL) < 1.0e-16) return ln * 0.0;
  const T s   = clamp_unit(dot(n, rJL) / sqrt_(ln * lL));
  const T chi = kRad2Deg * asin_(s);
  return 0.5 * kMdyneA2Kcal * kDeg2Rad * kDeg2Rad * koop * chi * chi;
}

template <typename T> __device__ __forceinline__ T mmff_torsion(const T c, const double V1, const double V2, const double V3) {
  const T c2 = c * c;
  return 0.5 * (V1 * (1.0 + c) + V2 * (1.0 - (2.0 * c2 - 1.0)) + V3 * (1.0 + (4.0 * c2 * c - 3.0 * c)));
}

// buffered 14-7: E(r), dE/dr
// ---- lean reciprocal / square root for the O(N^2) pair terms ------------------------------------------
// The pair loops are VALU bound (tools/probe_mmff_groups.py: 0.34 us per MMFF pair term and thread, about 70 % of it in
// two IEEE square roots and five IEEE divisions per van der Waals + electrostatic pair).  An IEEE f64 division is 11
// instructions with scaling and fix-up for subnormal / infinite operands, a square root about 20; the operands here are
// inter-atomic distances and sums of positive powers of them, so the hardware seed and two Newton steps (5 and 9
// instructions, <= 2 ulp) do.  A van der Waals term also takes ONE reciprocal for its two denominators.
__device__ __forceinline__ double rcp_lean(const double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y        = fma(y, e, y);
  e        = fma(-x, y, 1.0);
  return fma(y, e, y);
}
struct Root {
  double r, rinv;  // sqrt(x), 1 / sqrt(x)
};
__device__ __forceinline__ Root root_lean(const double x0) {
  const double x = fmax(x0, 1.0e-300);  // coincident atoms: r -> 1e-150, forces stay finite (0 * rinv = 0)
  const double y = __builtin_amdgcn_rsq(x);
  double       g = x * y, h = 0.5 * y;
  double       e = fma(-h, g, 0.5);
  g              = fma(g, e, g);
  h              = fma(h, e, h);
  e              = fma(-h, g, 0.5);
  g              = fma(g, e, g);
  h              = fma(h, e, h);
  return {g, 2.0 * h};
}

__device__ __forceinline__ void mmff_vdw(const double r, const double Rs, const double eps, double& e, double& dE_dr) {
  const double Rs2 = Rs * Rs, Rs7 = Rs2 * Rs2 * Rs2 * Rs;
  const double r2 = r * r, r6 = r2 * r2 * r2, r7 = r6 * r;
  const double t = r + 0.07 * Rs, den = r7 + 0.12 * Rs7;
  const double ip = rcp_lean(t * den), it = ip * den, iden = ip * t;  // 1 / t, 1 / den
  const double a   = 1.07 * Rs * it;
  const double a2  = a * a, a7 = a2 * a2 * a2 * a;
  const double b   = 1.12 * Rs7 * iden;
  e                = eps * a7 * (b - 2.0);
  const double da7 = -7.0 * a7 * it;
  const double db  = -b * 7.0 * r6 * iden;
  dE_dr            = eps * (da7 * (b - 2.0) + a7 * db);
}

// buffered Coulomb: chargeTerm = qi qj / D; dielModel 1 = constant (1/(r+b)), 2 = distance dependent (1/(r+b)^2)
__device__ __forceinline__ void mmff_ele(const double r, const double chargeTerm, const int dielModel, const bool is14, double& e,
                                         double& dE_dr) {
  const double inv = rcp_lean(r + 0.05);
  const double s   = is14 ? 0.75 : 1.0;
  const bool   sq  = dielModel == 2;
  e                = s * 332.0716 * chargeTerm * (sq ? inv * inv : inv);
  dE_dr            = (sq ? -2.0 : -1.0) * e * inv;
}

// ---- UFF terms (uff_kernels_device.cuh:37-580) --------------------------------------------------

// chain() on a plain double (energy-only instantiation of the templated terms)
__device__ __forceinline__ double chain(const double, const double f, const double) { return f; }

// harmonic stretch: E(r) and dE/dr
__device__ __forceinline__ void uff_bond(const double r, const double r0, const double k, double& e, double& dE_dr) {
  const double dr = r - r0;
  e               = 0.5 * k * dr * dr;
  dE_dr           = k * dr;
}

// Angle bend as a polynomial of c = cos(theta) (sin^2 = 1 - c^2), uff_kernels_device.cuh:78-108:
//   order 0: k (C0 + C1 cos(theta) + C2 cos(2 theta));  order n = 1..4: k (1 - cos(n theta)) / n^2 with the linear
//   case written 1 + cos(theta); plus exp(-20 (theta - theta0 + 0.25)) when order is 1..4 and cos(theta) > 0.866
//   (:167-170).  Orders outside 0..4 give k / order^2 (reference `default:` branch).
template <typename T>
__device__ __forceinline__ T uff_angle(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const double theta0, const double k,
                                       const int order, const double C0, const double C1, const double C2) {
  bool    ok;
  const T c = cos_angle(p1, p2, p3, ok);
  if (!ok) return c * 0.0;  // zero-length arm: no energy, no gradient (:157-159, :190-192)
  const T c2 = c * c;
  const T s2 = 1.0 - c2;
  T       e;
  if (order == 0) {
    e = k * (C0 + C1 * c + C2 * (c2 - s2));
  } else {
    T f;
    switch (order) {
      case 1: f = -c; break;
      case 2: f = c2 - s2; break;
      case 3: f = c * (c2 - 3.0 * s2); break;
      case 4: f = c2 * c2 - 6.0 * c2 * s2 + s2 * s2; break;
      default: f = c * 0.0; break;
    }
    e = (k / static_cast<double>(order * order)) * (1.0 - f);
    if (order < 5 && value(c) > 0.8660) {
      const T      theta = acos_(c);
      const double ex    = exp(-20.0 * (value(theta) - theta0 + 0.25));
      e                  = e + chain(theta - theta0, ex, -20.0 * ex);
    }
  }
  return e;
}
// Fourier torsion k/2 (1 - cosTerm cos(n phi)), n in {2, 3, 6}, as a polynomial of c = cos(phi) (:302-326).
// Other orders: energy 0, no gradient.
template <typename T> __device__ __forceinline__ T uff_torsion(const T c, const double k, const int order, const double cosTerm) {
  const T s2 = 1.0 - c * c;
  T       cn;
  switch (order) {
    case 2: cn = 1.0 - 2.0 * s2; break;
    case 3: cn = c * (c * c - 3.0 * s2); break;
    case 6: cn = 1.0 + s2 * (-32.0 * s2 * s2 + 48.0 * s2 - 18.0); break;
    default: return c * 0.0;
  }
  return 0.5 * k * (1.0 - cosTerm * cn);
}

// Inversion, value as `inversion` above.  The GRADIENT keeps a convention of the reference (inherited from RDKit's
// UFF Inversion contrib): its dE/dW is -k (C1 cosY - 4 C2 cosY sinY) (uff_kernels_device.cuh:497), i.e. the C2 part
// has the opposite sign of the true derivative of the energy it reports.  Coordinates after minimisation must match
// that minimiser, so the derivative carried here is k (C1 - 4 C2 sinY) d(sinY) while the value stays
// k (C0 + C1 sinY + C2 (2 sinY^2 - 1)).  (C2 != 0 only for group-15 centres; C, N, O use C2 = 0.)
__device__ __forceinline__ double uff_inversion(const Vec3<double>& p1, const Vec3<double>& p2, const Vec3<double>& p3,
                                                const Vec3<double>& p4, const double k, const double C0, const double C1,
                                                const double C2) {
  return inversion(p1, p2, p3, p4, C0, C1, C2, k);
}
template <int NP>
__device__ __forceinline__ Dual<NP> uff_inversion(const Vec3<Dual<NP>>& p1, const Vec3<Dual<NP>>& p2, const Vec3<Dual<NP>>& p3,
                                                  const Vec3<Dual<NP>>& p4, const double k, const double C0, const double C1,
                                                  const double C2) {
  // sinY with its derivative: inversion() with C0 = 0, C1 = 1, C2 = 0, k = 1 IS sinY (or the constant 1 when degenerate)
  const Dual<NP> sinY = inversion(p1, p2, p3, p4, 0.0, 1.0, 0.0, 1.0);
  return chain(sinY, k * (C0 + C1 * sinY.v + C2 * (2.0 * sinY.v * sinY.v - 1.0)), k * (C1 - 4.0 * C2 * sinY.v));
}

// 12-6 with a distance cutoff: E(r), dE/dr; zero beyond `threshold` (:527-580)
__device__ __forceinline__ void uff_vdw(const double r, const double xij, const double wellDepth, const double threshold, double& e,
                                        double& dE_dr) {
  e     = 0.0;
  dE_dr = 0.0;
  if (r > threshold || r <= 0.0) return;
  const double q  = xij / r;
  const double q2 = q * q, q6 = q2 * q2 * q2, q12 = q6 * q6;
  e               = wellDepth * (q12 - 2.0 * q6);
  dE_dr           = 12.0 * wellDepth / xij * (q6 * q - q12 * q);
}

// ---- constraint terms shared by MMFF and UFF (mmff_kernels_device.cuh:663-1036) --------------------

__device__ __forceinline__ double atan2_(const double y, const double x) { return atan2(y, x); }
template <int NP> __device__ __forceinline__ Dual<NP> atan2_(const Dual<NP>& y, const Dual<NP>& x) {
  const double den = x.v * x.v + y.v * y.v;
  Dual<NP>     r;
  r.v = atan2(y.v, x.v);
#pragma unroll
  for (int k = 0; k < NP; ++k) r.d[k] = den > 0.0 ? (x.v * y.d[k] - y.v * x.d[k]) / den : 0.0;
  return r;
}

__device__ __forceinline__ double normalize_angle_deg(double a) {  // into (-180, 180]
  a = fmod(a, 360.0);
  if (a < -180.0) {
    a += 360.0;
  } else if (a > 180.0) {
    a -= 360.0;
  }
  return a;
}

// signed dihedral 1-2-3-4 in radians with the reference's sign and floors (computeSignedDihedral, :899-960)
template <typename T>
__device__ __forceinline__ T signed_dihedral(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const Vec3<T>& p4) {
  const Vec3<T> r0 = p1 - p2, r1 = p3 - p2, r3 = p4 - p3;
  const Vec3<T> r2 = {-r1.x, -r1.y, -r1.z};
  Vec3<T>       t0 = cross(r0, r1), t1 = cross(r2, r3);
  T             d0 = sqrt_(dot(t0, t0)), d1 = sqrt_(dot(t1, t1));
  if (value(d0) < 1.0e-5) d0 = d0 * 0.0 + 1.0e-5;
  if (value(d1) < 1.0e-5) d1 = d1 * 0.0 + 1.0e-5;
  t0 = {t0.x / d0, t0.y / d0, t0.z / d0};
  t1 = {t1.x / d1, t1.y / d1, t1.z / d1};
  const T       cosPhi = clamp_unit(dot(t0, t1));
  const Vec3<T> m      = cross(t0, r1);
  T             ml     = sqrt_(dot(m, m));
  if (value(ml) < 1.0e-5) ml = ml * 0.0 + 1.0e-5;
  return -1.0 * atan2_(dot(m, t1) / ml, cosPhi);
}

// offset of a dihedral (degrees) from the window [minDeg, maxDeg] on the circle (computeDihedralConstraintTerm, :879-897)
__device__ __forceinline__ double dihedral_window_offset(const double dihedral, const double minDeg, const double maxDeg) {
  double target = dihedral;
  if (!(dihedral > minDeg && dihedral < maxDeg) && !(dihedral > minDeg && minDeg > maxDeg) &&
      !(dihedral < maxDeg && minDeg > maxDeg)) {
    const double toMin = normalize_angle_deg(dihedral - minDeg);
    const double toMax = normalize_angle_deg(dihedral - maxDeg);
    target             = fabs(toMin) < fabs(toMax) ? minDeg : maxDeg;
  }
  return normalize_angle_deg(dihedral - target);
}

template <typename T>
__device__ __forceinline__ T torsion_constraint(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const Vec3<T>& p4,
                                                const double minDeg, const double maxDeg, const double k) {
  const T      phi = kRad2Deg * signed_dihedral(p1, p2, p3, p4);
  const double off = dihedral_window_offset(value(phi), minDeg, maxDeg);
  const T      d   = phi - (value(phi) - off);  // same derivative as phi, value = offset from the window
  return k * d * d;
}

// position restraint: E(dist) and dE/ddist for dist = |p - ref|
__device__ __forceinline__ void position_constraint(const double dist, const double maxDispl, const double k, double& e, double& dE) {
  const double t = dist > maxDispl ? dist - maxDispl : 0.0;
  e              = 0.5 * k * t * t;
  dE             = k * t;
}

// angle restraint with the reference's arm-length floor (1e-5 on the squared lengths, :806-812)
template <typename T>
__device__ __forceinline__ T angle_constraint_ff(const Vec3<T>& p1, const Vec3<T>& p2, const Vec3<T>& p3, const double minDeg,
                                                 const double maxDeg, const double k) {
  const Vec3<T> r1 = p1 - p2, r2 = p3 - p2;
  T             l1 = dot(r1, r1), l2 = dot(r2, r2);
  if (value(l1) < 1.0e-5) l1 = l1 * 0.0 + 1.0e-5;
  if (value(l2) < 1.0e-5) l2 = l2 * 0.0 + 1.0e-5;
  const T theta = kRad2Deg * acos_(clamp_unit(dot(r1, r2) / sqrt_(l1 * l2)));
  if (value(theta) < minDeg) return k * (theta - minDeg) * (theta - minDeg);
  if (value(theta) > maxDeg) return k * (theta - maxDeg) * (theta - maxDeg);
  return theta * 0.0;
}

}  // namespace ff
}  // namespace nvmk
