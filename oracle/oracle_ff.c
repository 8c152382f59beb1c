/* CPU oracle for the force-field / BFGS / ETKDG path — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * A C + OpenMP fp64 restatement of
 *   - the DG / ETK / MMFF94 / UFF term energies with HAND-DERIVED analytic gradients
 *     (reference: src/forcefields/dist_geom_kernels_device.cuh:37-830, mmff_kernels_device.cuh:28-660,
 *      uff_kernels_device.cuh:37-580 — themselves ports of RDKit's ForceField contribs),
 *   - RDKit's BFGS (reference: src/minimizer/bfgs_minimize_permol_kernels.cu:35-745, constants :29-33),
 *   - the ETKDG stage pipeline with its stereochemistry checks (reference: src/etkdg.cpp:331-419,
 *     src/etkdg_impl.cpp:111-159,272-326, src/etkdg_stage_*.cu),
 * over the same flattened term tables the product consumes (include/nvmolkit_amd.h: nvmk_ff_batch layouts).
 *
 * Why it exists next to oracle/ff.py (numpy, finite-difference gradients): (1) it is the CPU baseline bench.py times
 * beside the GPU for the conformer half of the metric; (2) it minimises 100-200-atom systems in
 * milliseconds, so BFGS parity can be tested at the sizes the benchmark runs; (3) its gradients are derived by hand
 * from geometric primitives (cos angle, cos dihedral, Wilson sine) — an independent derivation from the product's
 * forward-mode dual numbers.  tests/test_oracle_ff_c.py pins it against oracle/ff.py (energies to 1e-12 relative,
 * gradients against central differences) before anything is compared with the GPU.
 *
 * Pinning status: "parity unpinned" against RDKit contribs for the same reason as oracle/ff.py (no RDKit in any image);
 * BFGS is pinned by the reference's RDKit-free quartic case (tests/test_bfgs_minimizer.cu:823-1029).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_DG 0
#define ORC_ETK 1
#define ORC_MMFF 2
#define ORC_QUARTIC 3
#define ORC_UFF 4

typedef struct {
  const int32_t* starts;
  const int32_t* idx;
  const double*  par;
} orc_group;

/* Same field order as nvmk_ff_batch (include/nvmolkit_amd.h:213-230); every pointer is a HOST pointer here. */
typedef struct {
  int32_t        kind;
  int32_t        n_systems;
  const int32_t* atom_starts;
  orc_group      groups[12];
  const int32_t* system_mol;
  uint32_t       group_mask;
  const int32_t* etk_ref12_starts;
  const double*  etk_ref12;
  const int32_t* etk_ref13_starts;
  const double*  etk_ref13;
} orc_ff_batch;

static const double PI_      = 3.14159265358979323846;
static const double RAD2DEG_ = 180.0 / 3.14159265358979323846;
static const double DEG2RAD_ = 3.14159265358979323846 / 180.0;
static const double MDYNE    = 143.9325;

static int kind_dim(int kind) { return (kind == ORC_DG || kind == ORC_QUARTIC) ? 4 : 3; }

/* ---- small vector helpers ---------------------------------------------------------------------- */
typedef struct { double x, y, z; } v3;
static inline v3     vsub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline v3     vadd(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline v3     vscale(v3 a, double s) { return (v3){a.x * s, a.y * s, a.z * s}; }
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3     vcross(v3 a, v3 b) { return (v3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline v3     vneg(v3 a) { return (v3){-a.x, -a.y, -a.z}; }
static inline v3     ld(const double* pos, int dim, int a) { return (v3){pos[a * dim], pos[a * dim + 1], pos[a * dim + 2]}; }
static inline void   push(double* g, int dim, int a, v3 f) {
  g[a * dim] += f.x;
  g[a * dim + 1] += f.y;
  g[a * dim + 2] += f.z;
}

/* cos of the angle 1-2-3 and its gradient wrt the three points.  ok = 0: a zero-length arm (no energy, no gradient).
 * A cosine clamped to +-1 has zero gradient (the product's clamp_unit on dual numbers does the same). */
static int cos_angle_g(v3 p1, v3 p2, v3 p3, double* c, v3 g[3]) {
  const v3     r1 = vsub(p1, p2), r2 = vsub(p3, p2);
  const double l1 = vdot(r1, r1), l2 = vdot(r2, r2);
  g[0] = g[1] = g[2] = (v3){0, 0, 0};
  *c                 = 0.0;
  if (!(l1 > 1.0e-16 && l2 > 1.0e-16)) return 0;
  const double inv = 1.0 / sqrt(l1 * l2);
  double       cc  = vdot(r1, r2) * inv;
  if (cc > 1.0) {
    *c = 1.0;
    return 1;
  }
  if (cc < -1.0) {
    *c = -1.0;
    return 1;
  }
  *c   = cc;
  g[0] = vsub(vscale(r2, inv), vscale(r1, cc / l1));
  g[2] = vsub(vscale(r1, inv), vscale(r2, cc / l2));
  g[1] = vneg(vadd(g[0], g[2]));
  return 1;
}

/* cos of the dihedral 1-2-3-4 and its gradient.  ok = 0: collinear (the callers use cos = 0, no gradient). */
static int cos_dihedral_g(v3 p1, v3 p2, v3 p3, v3 p4, double* c, v3 g[4]) {
  const v3     r1 = vsub(p1, p2), r2 = vsub(p3, p2), r4 = vsub(p4, p3);
  const v3     t1 = vcross(r1, r2), t2 = vcross(vneg(r2), r4);
  const double a = vdot(t1, t1), b = vdot(t2, t2), d = a * b;
  g[0] = g[1] = g[2] = g[3] = (v3){0, 0, 0};
  *c                        = 0.0;
  if (!(d > 1.0e-16)) return 0;
  const double inv = 1.0 / sqrt(d);
  double       cc  = vdot(t1, t2) * inv;
  if (cc > 1.0) {
    *c = 1.0;
    return 1;
  }
  if (cc < -1.0) {
    *c = -1.0;
    return 1;
  }
  *c = cc;
  const v3 v = vsub(vscale(t2, inv), vscale(t1, cc / a)); /* dc/dt1 */
  const v3 w = vsub(vscale(t1, inv), vscale(t2, cc / b)); /* dc/dt2 */
  /* t1 = r1 x r2, t2 = r4 x r2:  d(a x b).v = da.(b x v) + db.(v x a) */
  const v3 gr1 = vcross(r2, v);
  const v3 gr2 = vadd(vcross(v, r1), vcross(w, r4));
  const v3 gr4 = vcross(r2, w);
  g[0]         = gr1;
  g[1]         = vneg(vadd(gr1, gr2));
  g[2]         = vsub(gr2, gr4);
  g[3]         = gr4;
  return 1;
}

/* s = n . rJL / (|n| |rJL|), n = (p1 - p2) x (p3 - p2), rJL = p4 - p2 (Wilson sine / cosY of the inversion), gradient
 * wrt the four points.  Returns 0 when n or rJL vanish (thresholds: squared lengths, as the product). */
static int wilson_g(v3 p1, v3 p2, v3 p3, v3 p4, double nTol, double* s, v3 g[4]) {
  const v3     rI = vsub(p1, p2), rK = vsub(p3, p2), rL = vsub(p4, p2);
  const v3     n  = vcross(rI, rK);
  const double ln = vdot(n, n), lL = vdot(rL, rL);
  g[0] = g[1] = g[2] = g[3] = (v3){0, 0, 0};
  *s                        = 0.0;
  if (ln < nTol || lL < 1.0e-16) return 0;
  const double inv = 1.0 / sqrt(ln * lL);
  double       ss  = vdot(n, rL) * inv;
  if (ss > 1.0) {
    *s = 1.0;
    return 1;
  }
  if (ss < -1.0) {
    *s = -1.0;
    return 1;
  }
  *s           = ss;
  const v3 dn  = vsub(vscale(rL, inv), vscale(n, ss / ln));
  const v3 dL  = vsub(vscale(n, inv), vscale(rL, ss / lL));
  const v3 gI  = vcross(rK, dn);
  const v3 gK  = vcross(dn, rI);
  g[0]         = gI;
  g[2]         = gK;
  g[3]         = dL;
  g[1]         = vneg(vadd(vadd(gI, gK), dL));
  return 1;
}

/* ---- per-system evaluation -------------------------------------------------------------------- */
typedef struct {
  const orc_ff_batch* b;
  int                 sys; /* system index (atom_starts, etk refs) */
  int                 ms;  /* row of the term tables */
} sysref;

static inline int on(const orc_ff_batch* b, int g) {
  const uint32_t m = b->group_mask ? b->group_mask : 0xfffu;
  return ((m >> g) & 1u) && b->groups[g].starts != NULL;
}

static inline double pair_d2(const double* pos, int dim, int nd, int i, int j, double d[4]) {
  double s = 0.0;
  for (int c = 0; c < 4; ++c) {
    d[c] = c < nd ? pos[i * dim + c] - pos[j * dim + c] : 0.0;
    s += d[c] * d[c];
  }
  return s;
}
static inline void pair_push(double* g, int dim, int nd, int i, int j, const double d[4], double f) {
  for (int c = 0; c < nd; ++c) {
    g[i * dim + c] += f * d[c];
    g[j * dim + c] -= f * d[c];
  }
}

/* flat-bottom distance restraint (dist_geom_kernels_device.cuh:368-392, :696-729) */
static double flat_bottom(const double* pos, double* grad, int dim, int i, int j, double lo, double hi, double k) {
  double       d[4];
  const double dist = sqrt(pair_d2(pos, dim, 3, i, j, d));
  double       diff = 0.0;
  if (dist < lo) diff = dist - lo;
  else if (dist > hi) diff = dist - hi;
  if (grad && diff != 0.0) pair_push(grad, dim, 3, i, j, d, k * diff / (dist > 1.0e-8 ? dist : 1.0e-8));
  return 0.5 * k * diff * diff;
}

static double eval_dg(sysref r, const double* pos, double* grad, double w0, double w1) {
  const orc_ff_batch* b = r.b;
  double              e = 0.0;
  if (on(b, 0)) { /* distance violations in 4-D (:37-95) */
    const orc_group* g = &b->groups[0];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[2 * t], j = g->idx[2 * t + 1];
      const double lb2 = g->par[3 * t], ub2 = g->par[3 * t + 1], w = g->par[3 * t + 2];
      double       d[4];
      const double d2 = pair_d2(pos, 4, 4, i, j, d);
      if (d2 > ub2) {
        const double val = d2 / ub2 - 1.0;
        e += w * val * val;
        if (grad) pair_push(grad, 4, 4, i, j, d, 2.0 * (2.0 * w * val / ub2));
      } else if (d2 < lb2) {
        const double s = lb2 + d2, val = 2.0 * lb2 / s - 1.0;
        e += w * val * val;
        if (grad) pair_push(grad, 4, 4, i, j, d, 2.0 * (-4.0 * w * val * lb2 / (s * s)));
      }
    }
  }
  if (on(b, 1)) { /* chiral volumes, weight w0; RDKit's gradient is HALF the derivative (:97-207, :172-176) */
    const orc_group* g = &b->groups[1];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a  = g->idx + 4 * t;
      const v3       p4 = ld(pos, 4, a[3]);
      const v3       va = vsub(ld(pos, 4, a[0]), p4), vb = vsub(ld(pos, 4, a[1]), p4), vc = vsub(ld(pos, 4, a[2]), p4);
      const double   vol = vdot(va, vcross(vb, vc));
      const double   lo = g->par[2 * t], hi = g->par[2 * t + 1];
      double         dv = 0.0;
      if (vol < lo) dv = vol - lo;
      else if (vol > hi) dv = vol - hi;
      e += w0 * dv * dv;
      if (grad && dv != 0.0) {
        const double f  = w0 * dv; /* 0.5 * 2 w dv */
        const v3     g1 = vcross(vb, vc), g2 = vcross(vc, va), g3 = vcross(va, vb);
        push(grad, 4, a[0], vscale(g1, f));
        push(grad, 4, a[1], vscale(g2, f));
        push(grad, 4, a[2], vscale(g3, f));
        push(grad, 4, a[3], vscale(vadd(vadd(g1, g2), g3), -f));
      }
    }
  }
  if (on(b, 2)) { /* fourth dimension: E = w x4^2, RDKit gradient w x4 (:209-231) */
    const orc_group* g = &b->groups[2];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[t];
      const double x = pos[i * 4 + 3];
      e += w1 * x * x;
      if (grad) grad[i * 4 + 3] += w1 * x;
    }
  }
  return e;
}

static double eval_etk(sysref r, const double* pos, double* grad) {
  const orc_ff_batch* b = r.b;
  double              e = 0.0;
  if (on(b, 0)) { /* experimental torsions: sum_k fc[k] (1 + sign[k] cos((k+1) phi)) (:237-313, :447-575) */
    const orc_group* g = &b->groups[0];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a  = g->idx + 4 * t;
      const double*  fc = g->par + 12 * t;
      const double*  sg = fc + 6;
      double         c;
      v3             gc[4];
      const int      ok = cos_dihedral_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), ld(pos, 3, a[3]), &c, gc);
      if (!ok) c = 0.0; /* degenerate: cosPhi = 0 (:286-288) */
      const double c2 = c * c, c3 = c2 * c, c4 = c3 * c, c5 = c4 * c, c6 = c5 * c;
      const double k[6]  = {c, 2 * c2 - 1, 4 * c3 - 3 * c, 8 * c4 - 8 * c2 + 1, 16 * c5 - 20 * c3 + 5 * c,
                            32 * c6 - 48 * c4 + 18 * c2 - 1};
      const double dk[6] = {1.0, 4 * c, 12 * c2 - 3, 32 * c3 - 16 * c, 80 * c4 - 60 * c2 + 5, 192 * c5 - 192 * c3 + 36 * c};
      double       dE    = 0.0;
      for (int q = 0; q < 6; ++q) {
        e += fc[q] * (1.0 + sg[q] * k[q]);
        dE += fc[q] * sg[q] * dk[q];
      }
      if (grad && ok) {
        for (int m = 0; m < 4; ++m) push(grad, 3, a[m], vscale(gc[m], dE));
      }
    }
  }
  if (on(b, 1)) { /* inversions: k (C0 + C1 sinY + C2 cos2W) (:315-366, :577-694) */
    const orc_group* g = &b->groups[1];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a  = g->idx + 4 * t;
      const double   C0 = g->par[4 * t], C1 = g->par[4 * t + 1], C2 = g->par[4 * t + 2], k = g->par[4 * t + 3];
      const v3       p1 = ld(pos, 3, a[0]), p2 = ld(pos, 3, a[1]), p3 = ld(pos, 3, a[2]), p4 = ld(pos, 3, a[3]);
      const v3       rI = vsub(p1, p2), rK = vsub(p3, p2), rL = vsub(p4, p2);
      const double   lI = vdot(rI, rI), lK = vdot(rK, rK), lL = vdot(rL, rL);
      double         cosY;
      v3             gs[4];
      if (lI < 1.0e-16 || lK < 1.0e-16 || lL < 1.0e-16 || !wilson_g(p1, p2, p3, p4, 1.0e-16 * lI * lK, &cosY, gs)) {
        e += k * (C0 + C1 + C2); /* degenerate: cosY = 0 */
        continue;
      }
      double sinYSq = 1.0 - cosY * cosY;
      int    flat   = 0;
      if (sinYSq < 1.0e-16) {
        sinYSq = 1.0e-16;
        flat   = 1;
      }
      const double sinY = sqrt(sinYSq);
      e += k * (C0 + C1 * sinY + C2 * (2.0 * sinYSq - 1.0));
      if (grad && !flat) {
        const double dE = k * (C1 * (-cosY / sinY) + C2 * (-4.0 * cosY));
        for (int m = 0; m < 4; ++m) push(grad, 3, a[m], vscale(gs[m], dE));
      }
    }
  }
  for (int gi = 2; gi <= 5; ++gi) { /* flat-bottom distances: 1-2, 1-3, long range */
    if (gi == 4 || !on(b, gi)) continue;
    const orc_group* g      = &b->groups[gi];
    const double*    ref    = NULL;
    if (gi == 2 && b->etk_ref12) ref = b->etk_ref12 + b->etk_ref12_starts[r.sys];
    if (gi == 3 && b->etk_ref13) ref = b->etk_ref13 + b->etk_ref13_starts[r.sys];
    const int t0 = g->starts[r.ms];
    for (int t = t0; t < g->starts[r.ms + 1]; ++t) {
      double lo = g->par[4 * t], hi = g->par[4 * t + 1];
      if (ref && g->par[4 * t + 3] == 0.0) { /* re-centred on the current geometry (etkdg_stage_etk_minimization.cu:32-64) */
        const double half = 0.5 * (hi - lo);
        lo                = ref[t - t0] - half;
        hi                = ref[t - t0] + half;
      }
      e += flat_bottom(pos, grad, 3, g->idx[2 * t], g->idx[2 * t + 1], lo, hi, g->par[4 * t + 2]);
    }
  }
  if (on(b, 4)) { /* 1-3 angle restraints, force constant 1 (:394-445, :731-830) */
    const orc_group* g = &b->groups[4];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 3 * t;
      double         c;
      v3             gc[3];
      if (!cos_angle_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), &c, gc)) continue;
      const double theta = RAD2DEG_ * acos(c);
      double       diff  = 0.0;
      if (theta < g->par[2 * t]) diff = theta - g->par[2 * t];
      else if (theta > g->par[2 * t + 1]) diff = theta - g->par[2 * t + 1];
      e += diff * diff;
      if (grad && diff != 0.0) {
        const double s2 = 1.0 - c * c;
        const double dt = s2 > 1.0e-16 ? -RAD2DEG_ / sqrt(s2) : 0.0;
        for (int m = 0; m < 3; ++m) push(grad, 3, a[m], vscale(gc[m], 2.0 * diff * dt));
      }
    }
  }
  return e;
}

static double eval_mmff(sysref r, const double* pos, double* grad) {
  const orc_ff_batch* b = r.b;
  double              e = 0.0;
  if (on(b, 0)) { /* bond stretch (:241-295) */
    const orc_group* g = &b->groups[0];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[2 * t], j = g->idx[2 * t + 1];
      double       d[4];
      const double rr = sqrt(pair_d2(pos, 3, 3, i, j, d));
      const double dr = rr - g->par[2 * t], kb = g->par[2 * t + 1];
      const double cs = -2.0, cs2 = 7.0 / 12.0 * cs * cs;
      e += 0.5 * MDYNE * kb * dr * dr * (1.0 + cs * dr + cs2 * dr * dr);
      if (grad && rr > 0.0) pair_push(grad, 3, 3, i, j, d, MDYNE * kb * dr * (1.0 + 1.5 * cs * dr + 2.0 * cs2 * dr * dr) / rr);
    }
  }
  if (on(b, 1)) { /* angle bend (:297-390) */
    const orc_group* g = &b->groups[1];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 3 * t;
      const double   theta0 = g->par[3 * t], ka = g->par[3 * t + 1];
      const int      lin    = g->par[3 * t + 2] != 0.0;
      double         c;
      v3             gc[3];
      if (!cos_angle_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), &c, gc)) continue;
      double dE;
      if (lin) {
        e += MDYNE * ka * (1.0 + c);
        dE = MDYNE * ka;
      } else {
        const double cb = -0.4 * DEG2RAD_;
        const double dt = RAD2DEG_ * acos(c) - theta0;
        const double pre = 0.5 * MDYNE * DEG2RAD_ * DEG2RAD_ * ka;
        e += pre * dt * dt * (1.0 + cb * dt);
        const double s2 = 1.0 - c * c;
        dE              = pre * (2.0 * dt + 3.0 * cb * dt * dt) * (s2 > 1.0e-16 ? -RAD2DEG_ / sqrt(s2) : 0.0);
      }
      if (grad) {
        for (int m = 0; m < 3; ++m) push(grad, 3, a[m], vscale(gc[m], dE));
      }
    }
  }
  if (on(b, 2)) { /* stretch-bend (:392-495) */
    const orc_group* g = &b->groups[2];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 3 * t;
      const double*  p = g->par + 5 * t;
      const v3       p1 = ld(pos, 3, a[0]), p2 = ld(pos, 3, a[1]), p3 = ld(pos, 3, a[2]);
      const v3       r1 = vsub(p1, p2), r2 = vsub(p3, p2);
      const double   d1 = sqrt(vdot(r1, r1)), d2 = sqrt(vdot(r2, r2));
      if (d1 < 1.0e-8 || d2 < 1.0e-8) continue;
      double c;
      v3     gc[3];
      cos_angle_g(p1, p2, p3, &c, gc);
      const double dt   = RAD2DEG_ * acos(c) - p[0];
      const double sb   = (d1 - p[1]) * p[3] + (d2 - p[2]) * p[4];
      e += 2.51210 * dt * sb;
      if (grad) {
        const double s2  = 1.0 - c * c;
        const double dth = 2.51210 * sb * (s2 > 1.0e-16 ? -RAD2DEG_ / sqrt(s2) : 0.0);
        const double f1 = 2.51210 * dt * p[3] / d1, f2 = 2.51210 * dt * p[4] / d2;
        const v3     g1 = vadd(vscale(gc[0], dth), vscale(r1, f1));
        const v3     g3 = vadd(vscale(gc[2], dth), vscale(r2, f2));
        push(grad, 3, a[0], g1);
        push(grad, 3, a[2], g3);
        push(grad, 3, a[1], vsub(vscale(gc[1], dth), vadd(vscale(r1, f1), vscale(r2, f2))));
      }
    }
  }
  if (on(b, 3)) { /* Wilson out-of-plane (:28-108, :497-538) */
    const orc_group* g = &b->groups[3];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 4 * t;
      double         s;
      v3             gs[4];
      if (!wilson_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), ld(pos, 3, a[3]), 1.0e-16, &s, gs)) continue;
      const double chi = RAD2DEG_ * asin(s);
      const double pre = 0.5 * MDYNE * DEG2RAD_ * DEG2RAD_ * g->par[t];
      e += pre * chi * chi;
      if (grad) {
        const double c2 = 1.0 - s * s;
        const double dE = pre * 2.0 * chi * RAD2DEG_ * (c2 > 1.0e-16 ? 1.0 / sqrt(c2) : 1.0e8);
        for (int m = 0; m < 4; ++m) push(grad, 3, a[m], vscale(gs[m], dE));
      }
    }
  }
  if (on(b, 4)) { /* torsion (:110-188, :540-576) */
    const orc_group* g = &b->groups[4];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 4 * t;
      const double   V1 = g->par[3 * t], V2 = g->par[3 * t + 1], V3 = g->par[3 * t + 2];
      double         c;
      v3             gc[4];
      const int      ok = cos_dihedral_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), ld(pos, 3, a[3]), &c, gc);
      if (!ok) c = 0.0;
      const double c2 = c * c;
      e += 0.5 * (V1 * (1.0 + c) + V2 * (1.0 - (2.0 * c2 - 1.0)) + V3 * (1.0 + (4.0 * c2 * c - 3.0 * c)));
      if (grad && ok) {
        const double dE = 0.5 * (V1 - 4.0 * V2 * c + V3 * (12.0 * c2 - 3.0));
        for (int m = 0; m < 4; ++m) push(grad, 3, a[m], vscale(gc[m], dE));
      }
    }
  }
  if (on(b, 5)) { /* buffered 14-7 van der Waals (:190-239, :578-600) */
    const orc_group* g = &b->groups[5];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[2 * t], j = g->idx[2 * t + 1];
      const double Rs = g->par[2 * t], eps = g->par[2 * t + 1];
      double       d[4];
      const double rr  = sqrt(pair_d2(pos, 3, 3, i, j, d));
      const double Rs2 = Rs * Rs, Rs7 = Rs2 * Rs2 * Rs2 * Rs;
      const double r2 = rr * rr, r6 = r2 * r2 * r2, r7 = r6 * rr;
      const double aa  = 1.07 * Rs / (rr + 0.07 * Rs);
      const double a2 = aa * aa, a7 = a2 * a2 * a2 * aa;
      const double den = r7 + 0.12 * Rs7;
      const double bb  = 1.12 * Rs7 / den;
      e += eps * a7 * (bb - 2.0);
      if (grad && rr > 0.0) {
        const double da7 = -7.0 * a7 / (rr + 0.07 * Rs);
        const double db  = -bb * 7.0 * r6 / den;
        pair_push(grad, 3, 3, i, j, d, eps * (da7 * (bb - 2.0) + a7 * db) / rr);
      }
    }
  }
  if (on(b, 6)) { /* buffered Coulomb (:602-660) */
    const orc_group* g = &b->groups[6];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[2 * t], j = g->idx[2 * t + 1];
      const double q = g->par[3 * t];
      const int    model = (int)g->par[3 * t + 1];
      const double s = g->par[3 * t + 2] != 0.0 ? 0.75 : 1.0;
      double       d[4];
      const double rr = sqrt(pair_d2(pos, 3, 3, i, j, d));
      const double rb = rr + 0.05;
      double       et, dE;
      if (model == 2) {
        et = s * 332.0716 * q / (rb * rb);
        dE = -2.0 * et / rb;
      } else {
        et = s * 332.0716 * q / rb;
        dE = -et / rb;
      }
      e += et;
      if (grad && rr > 0.0) pair_push(grad, 3, 3, i, j, d, dE / rr);
    }
  }
  return e;
}

static double eval_uff(sysref r, const double* pos, double* grad) {
  const orc_ff_batch* b = r.b;
  double              e = 0.0;
  if (on(b, 0)) { /* harmonic stretch (:37-76) */
    const orc_group* g = &b->groups[0];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[2 * t], j = g->idx[2 * t + 1];
      double       d[4];
      const double rr = sqrt(pair_d2(pos, 3, 3, i, j, d));
      const double dr = rr - g->par[2 * t], k = g->par[2 * t + 1];
      e += 0.5 * k * dr * dr;
      if (grad) {
        if (rr > 0.0) {
          pair_push(grad, 3, 3, i, j, d, k * dr / rr);
        } else { /* coincident atoms: pushed apart along (1, 1, 1) with k / 100 (:56-58) */
          const double one[4] = {1.0, 1.0, 1.0, 0.0};
          pair_push(grad, 3, 3, i, j, one, k * 0.01);
        }
      }
    }
  }
  if (on(b, 1)) { /* angle bend (:78-240) */
    const orc_group* g = &b->groups[1];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 3 * t;
      const double*  p = g->par + 6 * t;
      const int      order = (int)p[2];
      double         c;
      v3             gc[3];
      if (!cos_angle_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), &c, gc)) continue;
      const double c2 = c * c, s2 = 1.0 - c2, k = p[1];
      double       et, dE;
      if (order == 0) {
        et = k * (p[3] + p[4] * c + p[5] * (c2 - s2));
        dE = k * (p[4] + 4.0 * p[5] * c);
      } else {
        double f, df;
        switch (order) {
          case 1: f = -c; df = -1.0; break;
          case 2: f = c2 - s2; df = 4.0 * c; break;
          case 3: f = c * (c2 - 3.0 * s2); df = 12.0 * c2 - 3.0; break; /* 4c^3 - 3c */
          case 4: f = c2 * c2 - 6.0 * c2 * s2 + s2 * s2; df = 32.0 * c2 * c - 16.0 * c; break; /* 8c^4 - 8c^2 + 1 */
          default: f = 0.0; df = 0.0; break;
        }
        const double pre = k / (double)(order * order);
        et               = pre * (1.0 - f);
        dE               = -pre * df;
        if (order < 5 && c > 0.8660) {
          const double theta = acos(c);
          const double ex    = exp(-20.0 * (theta - p[0] + 0.25));
          et += ex;
          dE += -20.0 * ex * (s2 > 1.0e-16 ? -1.0 / sqrt(s2) : 0.0);
        }
      }
      e += et;
      if (grad) {
        for (int m = 0; m < 3; ++m) push(grad, 3, a[m], vscale(gc[m], dE));
      }
    }
  }
  if (on(b, 2)) { /* Fourier torsion (:242-389) */
    const orc_group* g = &b->groups[2];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a = g->idx + 4 * t;
      const double   k = g->par[3 * t], cosTerm = g->par[3 * t + 2];
      const int      order = (int)g->par[3 * t + 1];
      double         c;
      v3             gc[4];
      const int      ok = cos_dihedral_g(ld(pos, 3, a[0]), ld(pos, 3, a[1]), ld(pos, 3, a[2]), ld(pos, 3, a[3]), &c, gc);
      if (!ok) c = 0.0;
      const double c2 = c * c, s2 = 1.0 - c2;
      double       cn, dcn;
      switch (order) {
        case 2: cn = 1.0 - 2.0 * s2; dcn = 4.0 * c; break;
        case 3: cn = c * (c2 - 3.0 * s2); dcn = 12.0 * c2 - 3.0; break;
        case 6: cn = 1.0 + s2 * (-32.0 * s2 * s2 + 48.0 * s2 - 18.0); dcn = (-96.0 * s2 * s2 + 96.0 * s2 - 18.0) * (-2.0 * c); break;
        default: continue;
      }
      e += 0.5 * k * (1.0 - cosTerm * cn);
      if (grad && ok) {
        const double dE = -0.5 * k * cosTerm * dcn;
        for (int m = 0; m < 4; ++m) push(grad, 3, a[m], vscale(gc[m], dE));
      }
    }
  }
  if (on(b, 3)) { /* inversion; the gradient keeps the reference's sign for the C2 part (:442-525, :497) */
    const orc_group* g = &b->groups[3];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int32_t* a  = g->idx + 4 * t;
      const double   k = g->par[4 * t], C0 = g->par[4 * t + 1], C1 = g->par[4 * t + 2], C2 = g->par[4 * t + 3];
      const v3       p1 = ld(pos, 3, a[0]), p2 = ld(pos, 3, a[1]), p3 = ld(pos, 3, a[2]), p4 = ld(pos, 3, a[3]);
      const v3       rI = vsub(p1, p2), rK = vsub(p3, p2), rL = vsub(p4, p2);
      const double   lI = vdot(rI, rI), lK = vdot(rK, rK), lL = vdot(rL, rL);
      double         cosY;
      v3             gs[4];
      if (lI < 1.0e-16 || lK < 1.0e-16 || lL < 1.0e-16 || !wilson_g(p1, p2, p3, p4, 1.0e-16 * lI * lK, &cosY, gs)) {
        e += k * (C0 + C1 + C2);
        continue;
      }
      double sinYSq = 1.0 - cosY * cosY;
      int    flat   = 0;
      if (sinYSq < 1.0e-16) {
        sinYSq = 1.0e-16;
        flat   = 1;
      }
      const double sinY = sqrt(sinYSq);
      e += k * (C0 + C1 * sinY + C2 * (2.0 * sinYSq - 1.0));
      if (grad && !flat) {
        const double dE = k * (C1 - 4.0 * C2 * sinY) * (-cosY / sinY); /* reference convention: -4 C2, not +4 C2 */
        for (int m = 0; m < 4; ++m) push(grad, 3, a[m], vscale(gs[m], dE));
      }
    }
  }
  if (on(b, 4)) { /* 12-6 with cutoff (:527-580) */
    const orc_group* g = &b->groups[4];
    for (int t = g->starts[r.ms]; t < g->starts[r.ms + 1]; ++t) {
      const int    i = g->idx[2 * t], j = g->idx[2 * t + 1];
      const double xij = g->par[3 * t], wd = g->par[3 * t + 1], thr = g->par[3 * t + 2];
      double       d[4];
      const double rr = sqrt(pair_d2(pos, 3, 3, i, j, d));
      if (rr > thr) continue;
      if (rr <= 0.0) {
        if (grad) {
          const double one[4] = {1.0, 1.0, 1.0, 0.0};
          pair_push(grad, 3, 3, i, j, one, 100.0);
        }
        continue;
      }
      const double q = xij / rr, q2 = q * q, q6 = q2 * q2 * q2, q12 = q6 * q6;
      e += wd * (q12 - 2.0 * q6);
      if (grad) pair_push(grad, 3, 3, i, j, d, 12.0 * wd / xij * (q6 * q - q12 * q) / rr);
    }
  }
  return e;
}

/* Energy of one system; when grad != NULL it is ZEROED first and receives the (unscaled) gradient. */
static double system_eval(sysref r, const double* pos, double* grad, double w0, double w1, int n_atoms, int coord_start) {
  const int kind = r.b->kind, dim = kind_dim(kind);
  if (grad) memset(grad, 0, sizeof(double) * (size_t)n_atoms * dim);
  switch (kind) {
    case ORC_DG: return eval_dg(r, pos, grad, w0, w1);
    case ORC_ETK: return eval_etk(r, pos, grad);
    case ORC_MMFF: return eval_mmff(r, pos, grad);
    case ORC_UFF: return eval_uff(r, pos, grad);
    default: { /* quartic test field (tests/test_bfgs_minimizer.cu:823-860) */
      double e = 0.0;
      for (int p = 0; p < n_atoms * 4; ++p) {
        if ((p & 3) == 3 && w0 == 0.0) continue;
        const double diff = pos[p] - (double)(coord_start + p);
        e += diff * diff * diff * diff;
        if (grad) grad[p] += 4.0 * diff * diff * diff;
      }
      return e;
    }
  }
}

static sysref make_ref(const orc_ff_batch* b, int s) { return (sysref){b, s, b->system_mol ? b->system_mol[s] : s}; }

void orc_ff_energy(const orc_ff_batch* b, double w0, double w1, const double* pos, const uint8_t* active, double* energies) {
  const int dim = kind_dim(b->kind);
#pragma omp parallel for schedule(dynamic, 8)
  for (int s = 0; s < b->n_systems; ++s) {
    if (active && !active[s]) continue;
    const int a0 = b->atom_starts[s];
    energies[s]  = system_eval(make_ref(b, s), pos + (size_t)a0 * dim, NULL, w0, w1, b->atom_starts[s + 1] - a0, a0 * dim);
  }
}

void orc_ff_gradient(const orc_ff_batch* b, double w0, double w1, const double* pos, const uint8_t* active, double* grad) {
  const int dim = kind_dim(b->kind);
#pragma omp parallel for schedule(dynamic, 8)
  for (int s = 0; s < b->n_systems; ++s) {
    if (active && !active[s]) continue;
    const int a0 = b->atom_starts[s];
    system_eval(make_ref(b, s), pos + (size_t)a0 * dim, grad + (size_t)a0 * dim, w0, w1, b->atom_starts[s + 1] - a0, a0 * dim);
  }
}

/* ---- BFGS (RDKit BFGSOpt.h as restated by bfgs_minimize_permol_kernels.cu:35-745) ------------------------------ */
#define FUNCTOL 1.0e-4
#define MOVETOL 1.0e-7
#define TOLX (4.0 * 3.0e-8)
#define EPS_HESS 3.0e-8
#define MAX_LS 1000

typedef struct {
  double* x;
  double* g;
  double* d;
  double* trial;
  double* old;
  double* dg;
  double* hdg;
  double* H;
  size_t  cap;
} bfgs_ws;

static void ws_reserve(bfgs_ws* w, size_t n) {
  if (n <= w->cap) return;
  free(w->x);
  free(w->H);
  w->x     = (double*)malloc(sizeof(double) * 7 * n);
  w->g     = w->x + n;
  w->d     = w->g + n;
  w->trial = w->d + n;
  w->old   = w->trial + n;
  w->dg    = w->old + n;
  w->hdg   = w->dg + n;
  w->H     = (double*)malloc(sizeof(double) * n * n);
  w->cap   = n;
}
static void ws_free(bfgs_ws* w) {
  free(w->x);
  free(w->H);
  memset(w, 0, sizeof(*w));
}

static double scaled_grad(sysref r, const double* x, double* g, double w0, double w1, int n_atoms, int coord_start, int scale,
                          int n, double* gscale) {
  (void)system_eval(r, x, g, w0, w1, n_atoms, coord_start);
  double sc = scale ? 0.1 : 1.0, mx = 0.0;
  for (int i = 0; i < n; ++i) {
    if (scale) g[i] *= sc;
    mx = fmax(mx, fabs(g[i]));
  }
  if (scale && mx > 10.0) {
    while (mx * sc > 10.0) sc *= 0.5;
    for (int i = 0; i < n; ++i) g[i] *= sc;
  }
  *gscale = sc;
  return mx;
}

/* Minimises one system in place.  Returns iterations; *converged, *energy set. */
static int bfgs_one(sysref r, double* xio, int n_atoms, int coord_start, double w0, double w1, int max_iters, double grad_tol,
                    int scale, bfgs_ws* w, double* energy, int* converged, int64_t* n_evals) {
  const int dim = kind_dim(r.b->kind), n = n_atoms * dim;
  *converged    = 0;
  if (n == 0) {
    *energy = 0.0;
    return 0;
  }
  ws_reserve(w, (size_t)n);
  double *x = w->x, *g = w->g, *d = w->d, *trial = w->trial, *old = w->old, *dg = w->dg, *hdg = w->hdg, *H = w->H;
  memcpy(x, xio, sizeof(double) * n);
  double ePrev = system_eval(r, x, NULL, w0, w1, n_atoms, coord_start);
  double gscale;
  scaled_grad(r, x, g, w0, w1, n_atoms, coord_start, scale, n, &gscale);
  int64_t evals = 1;
  memset(H, 0, sizeof(double) * (size_t)n * n);
  double sumsq = 0.0;
  for (int i = 0; i < n; ++i) {
    H[(size_t)i * n + i] = 1.0;
    d[i]                 = -g[i];
    sumsq += x[i] * x[i];
  }
  const double maxStep2 = 1.0e4 * fmax(sumsq, (double)n * (double)n);
  int          it       = 0;
  while (it < max_iters) {
    memcpy(old, x, sizeof(double) * n);
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += d[i] * d[i];
    if (s > maxStep2) {
      const double sc = sqrt(maxStep2 / s);
      for (int i = 0; i < n; ++i) d[i] *= sc;
    }
    double slope = 0.0, test = 0.0;
    for (int i = 0; i < n; ++i) {
      slope += d[i] * g[i];
      test = fmax(test, fabs(d[i]) / fmax(fabs(x[i]), 1.0));
    }
    const double lamMin = MOVETOL / (test > 0.0 ? test : 1.0e-20);
    double       lam = 1.0, lam2 = 0.0, e2 = 0.0, eNew = ePrev;
    for (int ls = 0; ls < MAX_LS; ++ls) {
      for (int i = 0; i < n; ++i) trial[i] = old[i] + lam * d[i];
      eNew = system_eval(r, trial, NULL, w0, w1, n_atoms, coord_start);
      ++evals;
      const double eDiff = eNew - ePrev;
      if (lam < lamMin || eDiff <= FUNCTOL * lam * slope) break;
      double tmp;
      if (ls == 0) {
        tmp = -slope / (2.0 * (eDiff - slope));
      } else {
        const double rhs1 = eDiff - lam * slope, rhs2 = e2 - ePrev - lam2 * slope;
        const double a  = (rhs1 / (lam * lam) - rhs2 / (lam2 * lam2)) / (lam - lam2);
        const double bq = (-lam2 * rhs1 / (lam * lam) + lam * rhs2 / (lam2 * lam2)) / (lam - lam2);
        if (a == 0.0) {
          tmp = -slope / (2.0 * bq);
        } else {
          const double disc = bq * bq - 3.0 * a * slope;
          if (disc < 0.0) tmp = 0.5 * lam;
          else if (bq <= 0.0) tmp = (-bq + sqrt(disc)) / (3.0 * a);
          else tmp = -slope / (bq + sqrt(disc));
        }
        tmp = fmin(tmp, 0.5 * lam);
      }
      lam2 = lam;
      e2   = eNew;
      lam  = fmax(tmp, 0.1 * lam);
    }
    double stepTest = 0.0;
    for (int i = 0; i < n; ++i) {
      x[i]     = trial[i];
      d[i]     = trial[i] - old[i]; /* xi */
      dg[i]    = g[i];
      stepTest = fmax(stepTest, fabs(d[i]) / fmax(fabs(x[i]), 1.0));
    }
    ePrev = eNew;
    if (stepTest < TOLX) {
      *converged = 1;
      break;
    }
    scaled_grad(r, x, g, w0, w1, n_atoms, coord_start, scale, n, &gscale);
    double gTest = 0.0;
    for (int i = 0; i < n; ++i) {
      dg[i] = g[i] - dg[i];
      gTest = fmax(gTest, fabs(g[i]) * fmax(fabs(x[i]), 1.0));
    }
    gTest /= fmax(ePrev * gscale, 1.0);
    if (gTest < grad_tol) {
      *converged = 1;
      break;
    }
    double fac = 0.0, fae = 0.0, sumDG = 0.0, sumXi = 0.0;
    for (int i = 0; i < n; ++i) {
      double        acc = 0.0;
      const double* Hi  = H + (size_t)i * n;
      for (int j = 0; j < n; ++j) acc += Hi[j] * dg[j];
      hdg[i] = acc;
    }
    for (int i = 0; i < n; ++i) {
      fac += dg[i] * d[i];
      fae += dg[i] * hdg[i];
      sumDG += dg[i] * dg[i];
      sumXi += d[i] * d[i];
    }
    if (fac > 0.0 && fac * fac > EPS_HESS * sumDG * sumXi) {
      const double rfac = 1.0 / fac, fad = 1.0 / fae;
      for (int i = 0; i < n; ++i) dg[i] = rfac * d[i] - fad * hdg[i]; /* u */
      for (int i = 0; i < n; ++i) {
        double*      Hi = H + (size_t)i * n;
        const double a = rfac * d[i], bq = fad * hdg[i], c = fae * dg[i];
        for (int j = 0; j < n; ++j) Hi[j] += a * d[j] - bq * hdg[j] + c * dg[j];
      }
    }
    for (int i = 0; i < n; ++i) {
      double        acc = 0.0;
      const double* Hi  = H + (size_t)i * n;
      for (int j = 0; j < n; ++j) acc += Hi[j] * g[j];
      d[i] = -acc;
    }
    ++it;
  }
  memcpy(xio, x, sizeof(double) * n);
  *energy = ePrev;
  if (n_evals) *n_evals += evals;
  return it;
}

void orc_bfgs_minimize(const orc_ff_batch* b, double w0, double w1, int max_iters, double grad_tol, int scale_grads, double* pos,
                       const uint8_t* active, double* energies, int16_t* statuses, int32_t* iters) {
  const int dim = kind_dim(b->kind);
#pragma omp parallel
  {
    bfgs_ws w;
    memset(&w, 0, sizeof(w));
#pragma omp for schedule(dynamic, 1)
    for (int s = 0; s < b->n_systems; ++s) {
      if (active && !active[s]) continue;
      const int a0 = b->atom_starts[s];
      int       conv;
      double    e;
      const int it = bfgs_one(make_ref(b, s), pos + (size_t)a0 * dim, b->atom_starts[s + 1] - a0, a0 * dim, w0, w1, max_iters,
                              grad_tol, scale_grads, &w, &e, &conv, NULL);
      energies[s]  = e;
      if (statuses) statuses[s] = conv ? 0 : 1;
      if (iters) iters[s] = it;
    }
    ws_free(&w);
  }
}

/* ---- ETKDG pipeline ------------------------------------------------------------------------------------------ */
#define ORC_N_STAGES 11

typedef struct {
  int32_t        n_mols;
  const int32_t* n_atoms;
  orc_group      dg[3];
  orc_group      etk[6];
  const int32_t* check_starts;
  const int32_t* check_kind;
  const int32_t* check_idx;
  const double*  check_par;
  const int32_t* num_impropers;
} orc_molset;

typedef struct {
  int32_t  confs_per_mol;
  int32_t  max_iterations;
  int32_t  batch_size;
  int32_t  use_exp_torsions;
  int32_t  use_basic_knowledge;
  int32_t  enforce_chirality;
  double   box_size;
  double   force_tol;
  uint64_t seed;
} orc_etkdg_params;

static uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

/* start coordinates of attempt `attempt` (the product's counter-based generator, nvmolkit_amd/csrc/etkdg.hip
 * random_coords_kernel; the reference draws from RDKit's RNG, src/etkdg_stage_coordgen.cu:100-121) */
void orc_etkdg_random_coords(uint64_t seed, uint64_t attempt, int n_atoms, double box, double* pos4) {
  const uint64_t base = splitmix64(seed ^ (attempt * 0x9e3779b97f4a7c15ull));
  for (int c = 0; c < 4 * n_atoms; ++c) {
    const uint64_t h = splitmix64(base + (uint64_t)c);
    const double   u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    pos4[c]          = (u - 0.5) * box;
  }
}

/* stereochemistry checks (src/etkdg_stage_stereochem_checks.cu:25-442); pos4 = 4 doubles per atom */
static int same_side(double tol, v3 v1, v3 v2, v3 v3_, v3 v4, v3 p0) {
  const v3     n  = vcross(vsub(v2, v1), vsub(v3_, v1));
  const double d1 = vdot(n, vsub(v4, v1)), d2 = vdot(n, vsub(p0, v1));
  if (fabs(d1) < tol || fabs(d2) < tol) return 0;
  return !((d1 < 0.0) ^ (d2 < 0.0));
}
static v3 unit_noguard(v3 a) { /* no zero guard: NaN propagates like RDKit's Point3D::normalize (kernel_utils.cuh:140-145) */
  const double l = sqrt(vdot(a, a));
  return (v3){a.x / l, a.y / l, a.z / l};
}
static int tetrahedral_ok(const double* p, const int32_t* ix, int volumeTest, int fused, double tol) {
  const v3 p0 = ld(p, 4, ix[0]), p1 = ld(p, 4, ix[1]), p2 = ld(p, 4, ix[2]), p3 = ld(p, 4, ix[3]), p4 = ld(p, 4, ix[4]);
  if (volumeTest) {
    const v3     d1 = unit_noguard(vsub(p0, p1)), d2 = unit_noguard(vsub(p0, p2)), d3 = unit_noguard(vsub(p0, p3)),
             d4  = unit_noguard(vsub(p0, p4));
    const double lim = (fused ? 0.25 : 1.0) * 0.50;
    if (fabs(vdot(vcross(d1, d2), d3)) < lim) return 0;
    if (fabs(vdot(vcross(d1, d2), d4)) < lim) return 0;
    if (fabs(vdot(vcross(d1, d3), d4)) < lim) return 0;
    if (fabs(vdot(vcross(d2, d3), d4)) < lim) return 0;
  }
  if (ix[0] == ix[4]) return 1;
  return same_side(tol, p1, p2, p3, p4, p0) && same_side(tol, p2, p3, p4, p1, p0) && same_side(tol, p3, p4, p1, p2, p0) &&
         same_side(tol, p4, p1, p2, p3, p0);
}
static int check_fails(const orc_molset* ms, int m, int kind, const double* p) {
  if (!ms->check_starts) return 0;
  for (int t = ms->check_starts[m]; t < ms->check_starts[m + 1]; ++t) {
    if (ms->check_kind[t] != kind) continue;
    const int32_t* ix = ms->check_idx + 5 * t;
    const double   a = ms->check_par[2 * t], b = ms->check_par[2 * t + 1];
    int            fail = 0;
    switch (kind) {
      case 0: fail = !tetrahedral_ok(p, ix, 1, a != 0.0, 0.3); break;
      case 3: fail = !tetrahedral_ok(p, ix, 0, 0, 0.1); break;
      case 1: {
        const v3     p4  = ld(p, 4, ix[4]);
        const double vol = vdot(vsub(ld(p, 4, ix[1]), p4), vcross(vsub(ld(p, 4, ix[2]), p4), vsub(ld(p, 4, ix[3]), p4)));
        const int    oppA = (signbit(vol) != 0) != (signbit(a) != 0), oppB = (signbit(vol) != 0) != (signbit(b) != 0);
        fail = (a > 0 && vol < a && (vol / a < 0.8 || oppA)) || (b < 0 && vol > b && (vol / b < 0.8 || oppB));
        break;
      }
      case 2: {
        const v3     d    = vsub(ld(p, 4, ix[0]), ld(p, 4, ix[1]));
        const double dist = sqrt(vdot(d, d));
        fail = (dist < a && fabs(dist - a) > 0.1 * b) || (dist > b && fabs(dist - b) > 0.1 * b);
        break;
      }
      case 4: {
        const v3     p0 = ld(p, 4, ix[0]), p1 = ld(p, 4, ix[1]), p2 = ld(p, 4, ix[2]), p3 = ld(p, 4, ix[3]);
        const v3     r1 = vsub(p2, p1), c1 = vcross(vsub(p0, p1), r1), c2 = vcross(vsub(p3, p2), r1);
        const double dot   = vdot(c1, c2) / sqrt(vdot(c1, c1) * vdot(c2, c2));
        const double angle = dot <= -1.0 ? PI_ : (dot >= 1.0 ? 0.0 : acos(dot));
        fail               = (angle - 0.5 * PI_) * a < 0.0;
        break;
      }
      case 5: {
        const v3 u = unit_noguard(vsub(ld(p, 4, ix[1]), ld(p, 4, ix[0]))), v = unit_noguard(vsub(ld(p, 4, ix[1]), ld(p, 4, ix[2])));
        fail       = (vdot(u, v) + 1.0) < 1.0e-3;
        break;
      }
      default: break;
    }
    if (fail) return 1;
  }
  return 0;
}

/* One attempt: the eleven stages of src/etkdg.cpp:331-419 on one molecule, in two halves — `half` 0: stages 0-4 (start
 * coordinates, first minimisation, its checks, fourth-dimension minimisation), `half` 1: stages 5-10 on the coordinates the first
 * half left in pos4 — because the product drops a molecule's surplus attempts between them (prune_surplus_kernel of
 * nvmolkit_amd/csrc/etkdg.hip).  Returns the index of the failing stage or -1; pos4 holds the coordinates on exit. */
static int etkdg_attempt(const orc_molset* ms, const orc_etkdg_params* prm, int m, uint64_t attempt, double* pos4, double* pos3,
                         double* ref, bfgs_ws* w, int64_t* bfgs_iters, int half) {
  const int    na    = ms->n_atoms[m];
  const int    useEtk = prm->use_exp_torsions || prm->use_basic_knowledge;
  const int32_t starts01[2] = {0, na};
  const int32_t molrow[1]   = {m};
  orc_ff_batch  dg;
  memset(&dg, 0, sizeof(dg));
  dg.kind        = ORC_DG;
  dg.n_systems   = 1;
  dg.atom_starts = starts01;
  dg.system_mol  = molrow;
  for (int g = 0; g < 3; ++g) dg.groups[g] = ms->dg[g];
  const sysref rdg = {&dg, 0, m};
  double       e;
  int          conv;
  if (half == 0) {
    orc_etkdg_random_coords(prm->seed, attempt, na, prm->box_size, pos4);
    /* stage 1: first minimisation, repeated until converged (etkdg_stage_distgeom_minimize.cu:53-58), E / atom < 0.05 */
    for (int rep = 0; rep < 50; ++rep) {
      *bfgs_iters += bfgs_one(rdg, pos4, na, 0, 1.0, 0.1, 400, prm->force_tol, 1, w, &e, &conv, NULL);
      if (conv) break;
    }
    e = system_eval(rdg, pos4, NULL, 1.0, 0.1, na, 0);
    if (na > 0 && e / na >= 0.05) return 1;
    if (check_fails(ms, m, 0, pos4)) return 2;
    if (prm->enforce_chirality && check_fails(ms, m, 1, pos4)) return 3;
    for (int rep = 0; rep < 50; ++rep) { /* stage 4: fourth-dimension minimisation */
      *bfgs_iters += bfgs_one(rdg, pos4, na, 0, 0.2, 1.0, 200, prm->force_tol, 1, w, &e, &conv, NULL);
      if (conv) break;
    }
    return -1;
  }
  if (useEtk) { /* stage 5 */
    orc_ff_batch etk;
    memset(&etk, 0, sizeof(etk));
    etk.kind        = ORC_ETK;
    etk.n_systems   = 1;
    etk.atom_starts = starts01;
    etk.system_mol  = molrow;
    for (int g = 0; g < 6; ++g) etk.groups[g] = ms->etk[g];
    etk.group_mask = prm->use_basic_knowledge ? 0x3fu : 0x3du;
    const int32_t zero2[2] = {0, 0};
    const int     n12 = ms->etk[2].starts[m + 1] - ms->etk[2].starts[m], n13 = ms->etk[3].starts[m + 1] - ms->etk[3].starts[m];
    for (int k = 0; k < 2; ++k) {
      const orc_group* g  = &ms->etk[2 + k];
      double*          rr = ref + (k ? n12 : 0);
      for (int t = g->starts[m]; t < g->starts[m + 1]; ++t) {
        const v3 d = vsub(ld(pos4, 4, g->idx[2 * t]), ld(pos4, 4, g->idx[2 * t + 1]));
        rr[t - g->starts[m]] = sqrt(vdot(d, d));
      }
    }
    (void)n13;
    etk.etk_ref12_starts = zero2;
    etk.etk_ref12        = ref;
    etk.etk_ref13_starts = zero2;
    etk.etk_ref13        = ref + n12;
    for (int a = 0; a < na; ++a) {
      for (int c = 0; c < 3; ++c) pos3[3 * a + c] = pos4[4 * a + c];
    }
    const sysref retk = {&etk, 0, m};
    *bfgs_iters += bfgs_one(retk, pos3, na, 0, 1.0, 1.0, 300, prm->force_tol, 1, w, &e, &conv, NULL);
    int planarFail = 0;
    if (prm->use_basic_knowledge) {
      orc_ff_batch planar = etk;
      planar.group_mask   = 0x2u;
      const sysref rp     = {&planar, 0, m};
      planarFail          = system_eval(rp, pos3, NULL, 1.0, 1.0, na, 0) > 0.7 * ms->num_impropers[m];
    }
    for (int a = 0; a < na; ++a) {
      for (int c = 0; c < 3; ++c) pos4[4 * a + c] = pos3[3 * a + c];
    }
    if (planarFail) return 5;
  }
  if (check_fails(ms, m, 5, pos4)) return 6;
  if (prm->enforce_chirality) {
    if (check_fails(ms, m, 1, pos4)) return 7;
    if (check_fails(ms, m, 2, pos4)) return 8;
    if (check_fails(ms, m, 3, pos4)) return 9;
    if (check_fails(ms, m, 4, pos4)) return 10;
  }
  return -1;
}

/* Whole embedding.  Attempts are handed out exactly as the product's scheduler does (round-robin over molecules,
 * oversubscription by rounds, batches of batch_size sorted largest-first; src/etkdg_impl.cpp:272-326 + the stable sort
 * of nvmolkit_amd/csrc/etkdg.hip), so attempt k of a run has the same start coordinates on both sides.
 * coords: conformer c of molecule m at 3 * (confs_per_mol * sum_{k<m} n_atoms[k] + c * n_atoms[m]). */
typedef struct { int mol; int pos; } sort_item;
static const int32_t* g_sort_natoms;
static int cmp_desc_atoms(const void* a, const void* b) {
  const sort_item* x = (const sort_item*)a;
  const sort_item* y = (const sort_item*)b;
  const int        d = g_sort_natoms[y->mol] - g_sort_natoms[x->mol];
  return d ? d : x->pos - y->pos; /* stable */
}

int64_t orc_etkdg_embed(const orc_molset* ms, const orc_etkdg_params* prm, double* coords, int32_t* conf_counts,
                        int32_t* stage_failures) {
  const int nMols = ms->n_mols, confs = prm->confs_per_mol, maxTries = prm->max_iterations * confs;
  int64_t*  slot = (int64_t*)calloc((size_t)nMols + 1, sizeof(int64_t));
  int*      completed = (int*)calloc((size_t)nMols, sizeof(int));
  int*      attempts  = (int*)calloc((size_t)nMols, sizeof(int));
  int       maxAtoms = 0, maxRef = 1;
  for (int m = 0; m < nMols; ++m) {
    slot[m + 1]    = slot[m] + (int64_t)ms->n_atoms[m] * confs * 3;
    conf_counts[m] = 0;
    if (ms->n_atoms[m] > maxAtoms) maxAtoms = ms->n_atoms[m];
    if (ms->etk[2].starts && ms->etk[3].starts) {
      const int nr = ms->etk[2].starts[m + 1] - ms->etk[2].starts[m] + ms->etk[3].starts[m + 1] - ms->etk[3].starts[m];
      if (nr > maxRef) maxRef = nr;
    }
  }
  if (stage_failures) memset(stage_failures, 0, sizeof(int32_t) * ORC_N_STAGES);
  sort_item* ids      = (sort_item*)malloc(sizeof(sort_item) * (size_t)prm->batch_size);
  int*       result   = (int*)malloc(sizeof(int) * (size_t)prm->batch_size);
  double*    batchPos = (double*)malloc(sizeof(double) * 4 * (size_t)maxAtoms * (size_t)prm->batch_size);
  uint64_t   dispatched = 0;
  int64_t    totalIters = 0;
  int        round = 1;
  for (;;) {
    int n = 0, prev = -1;
    while (n < prm->batch_size && prev != n) { /* Scheduler::dispatch */
      prev            = n;
      const int limit = maxTries < confs * round ? maxTries : confs * round;
      for (int m = 0; m < nMols; ++m) {
        while (completed[m] < confs && attempts[m] < limit) {
          if (n >= prm->batch_size) break;
          ids[n] = (sort_item){m, n};
          ++n;
          ++attempts[m];
        }
      }
      if (attempts[nMols - 1] == limit) ++round;
    }
    if (n == 0) break;
    const uint64_t base = dispatched;
    dispatched += (uint64_t)n;
    g_sort_natoms = ms->n_atoms;
    qsort(ids, (size_t)n, sizeof(sort_item), cmp_desc_atoms);
#pragma omp parallel
    {
      bfgs_ws w;
      memset(&w, 0, sizeof(w));
      double* pos3 = (double*)malloc(sizeof(double) * 3 * (size_t)(maxAtoms > 0 ? maxAtoms : 1));
      double* ref  = (double*)malloc(sizeof(double) * (size_t)maxRef);
      int64_t its  = 0;
#pragma omp for schedule(dynamic, 1)
      for (int s = 0; s < n; ++s) {
        result[s] = etkdg_attempt(ms, prm, ids[s].mol, base + (uint64_t)s, batchPos + 4 * (size_t)maxAtoms * s, pos3, ref, &w, &its, 0);
      }
#pragma omp single
      {
        /* between the halves: a molecule's attempts beyond what it still misses plus one spare leave, in batch order, without a
         * failure being counted (-2); what it misses is what it missed when the batch was handed out */
        for (int s = 0; s < n; ++s) {
          if (result[s] != -1) continue;
          const int m = ids[s].mol, keep = confs - conf_counts[m] + 1;
          int       rank = 0;
          for (int j = s - 1; j >= 0 && ids[j].mol == m; --j) rank += (result[j] == -1 || result[j] == -3);
          if (rank >= keep) result[s] = -3; /* marked, still counted by the attempts after it */
        }
        for (int s = 0; s < n; ++s) {
          if (result[s] == -3) result[s] = -2;
        }
      }
#pragma omp for schedule(dynamic, 1)
      for (int s = 0; s < n; ++s) {
        if (result[s] == -1)
          result[s] = etkdg_attempt(ms, prm, ids[s].mol, base + (uint64_t)s, batchPos + 4 * (size_t)maxAtoms * s, pos3, ref, &w, &its, 1);
      }
#pragma omp atomic
      totalIters += its;
      free(pos3);
      free(ref);
      ws_free(&w);
    }
    for (int s = 0; s < n; ++s) { /* record + pack, in batch order like the product */
      const int m = ids[s].mol;
      if (result[s] == -2) continue; /* a surplus attempt that left between the halves */
      if (result[s] >= 0) {
        if (stage_failures) ++stage_failures[result[s]];
        continue;
      }
      ++completed[m];
      if (conf_counts[m] < confs) {
        const int     na = ms->n_atoms[m];
        double*       o  = coords + slot[m] + (int64_t)conf_counts[m] * na * 3;
        const double* p  = batchPos + 4 * (size_t)maxAtoms * s;
        for (int a = 0; a < na; ++a) {
          for (int c = 0; c < 3; ++c) o[3 * a + c] = p[4 * a + c];
        }
        ++conf_counts[m];
      }
    }
  }
  free(ids);
  free(result);
  free(batchPos);
  free(slot);
  free(completed);
  free(attempts);
  return totalIters;
}
