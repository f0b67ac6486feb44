// kernels_cell.h -- the Q2-Q1 sum-factorised cell operation shared by the slot / block kernels
// (kernels_ops.hip) and the brick kernels (kernels_brick.hip): 1-D kernels on three values, the
// argument block, and cell_compute (gather, evaluate, q-point operation, integrate) templated on
// how the lanes of one cell exchange their velocity-gradient rows.
//   matrix_free_operators.cc:171-278 (Stokes), 399-432 (B^T), 705-783 (A block)
#pragma once
#include "agmg_internal.h"

namespace agmg {
namespace {

constexpr double SQ06 = 0.774596669241483377035853079956;  // sqrt(3/5)
constexpr double XQ0 = 0.5 - 0.5 * SQ06;
// Q2 shape values at the first Gauss point (the third is the mirror image, the middle one is
// the unit vector e_1 because the node 1/2 is a Gauss point)
constexpr double N00 = (1 - XQ0) * (1 - 2 * XQ0), N01 = 4 * XQ0 * (1 - XQ0), N02 = XQ0 * (2 * XQ0 - 1);
constexpr double M0 = 1 - XQ0, M1 = XQ0;  // Q1 shape values at the first Gauss point
constexpr double INV_S = 1.0 / (0.5 * SQ06);  // 1 / spacing of the Gauss points
constexpr double WQ0 = 5.0 / 18.0, WQ1 = 8.0 / 18.0;

__host__ __device__ constexpr double wq1(int q) { return q == 1 ? WQ1 : WQ0; }
__host__ __device__ constexpr double n1(int q, int i) {  // Q1 shape i at Gauss point q
  return q == 1 ? 0.5 : ((q == 0) == (i == 0) ? M0 : M1);
}
__host__ __device__ constexpr double n2(int q, int i) {  // Q2 shape i at Gauss point q
  return q == 1 ? (i == 1 ? 1.0 : 0.0) : (q == 0 ? (i == 0 ? N00 : (i == 1 ? N01 : N02))
                                                 : (i == 0 ? N02 : (i == 1 ? N01 : N00)));
}
__host__ __device__ constexpr double d2(int q, int i) {  // Q2 shape derivative i at Gauss point q
  const double x = q == 0 ? XQ0 : (q == 1 ? 0.5 : 1.0 - XQ0);
  return i == 0 ? 4 * x - 3 : (i == 1 ? 4 - 8 * x : 4 * x - 1);
}

// ---- 1-D kernels on three values -------------------------------------------------------
// nodal values -> values at the Gauss points (in place)
__device__ __forceinline__ void interp3(double &a, double &b, double &c) {
  const double t0 = fma(N00, a, fma(N01, b, N02 * c));
  const double t2 = fma(N02, a, fma(N01, b, N00 * c));
  a = t0;
  c = t2;
}
// transpose of interp3 (in place)
__device__ __forceinline__ void interp3T(double &a, double &b, double &c) {
  const double o0 = fma(N00, a, N02 * c);
  const double o1 = fma(N01, a + c, b);
  const double o2 = fma(N02, a, N00 * c);
  a = o0;
  b = o1;
  c = o2;
}
// collocation derivative at the (equispaced) Gauss points, times the spacing
__device__ __forceinline__ void deriv3(double a, double b, double c, double &d0, double &d1, double &d2_) {
  // centred difference h and second difference s: d0 = h - s, d2 = h + s (6 operations instead of 8)
  const double h = 0.5 * (c - a);
  const double s = fma(-2.0, b, a + c);
  d0 = h - s;
  d1 = h;
  d2_ = h + s;
}
// transpose of deriv3, accumulating
__device__ __forceinline__ void deriv3T(double f0, double f1, double f2, double &ga, double &gb, double &gc) {
  // with t = f0 - f2 and m = (f0 + f1 + f2) / 2: rows (-t - m, 2 t, -t + m) (8 operations instead of 10)
  const double t = f0 - f2;
  const double m = 0.5 * ((f0 + f1) + f2);
  ga -= t + m;
  gb = fma(2.0, t, gb);
  gc += m - t;
}
__host__ __device__ constexpr double dcol(int q, int i) {  // collocation derivative matrix * spacing
  return q == 0 ? (i == 0 ? -1.5 : (i == 1 ? 2.0 : -0.5))
                : (q == 1 ? (i == 0 ? -0.5 : (i == 1 ? 0.0 : 0.5)) : (i == 0 ? 0.5 : (i == 1 ? -2.0 : 1.5)));
}

// Stokes pressure rows: the DIM lanes of a cell each hold the divergence contribution of their
// velocity component; they are summed across the lanes (fixed order c = 0, 1, 2) so that one value
// per (cell, pressure node) goes to the slots instead of DIM
template <int DIM, int NP>
__device__ __forceinline__ void sum_over_components(double *pout) {
#pragma unroll
  for (int b = 0; b < NP; ++b) {
    const double a1 = __shfl_down(pout[b], 1, 64);
    if (DIM == 3) {
      const double a2 = __shfl_down(pout[b], 2, 64);
      pout[b] = (pout[b] + a1) + a2;
    } else {
      pout[b] = pout[b] + a1;
    }
  }
}

struct CellArgs {
  int n_cells, cstride;
  const uint32_t *cnvT, *cnpT;
  const double *hT;                  // Cartesian: [DIM][cstride]
  const double *invJT, *jxwT;        // general: [q][DIM*DIM][cstride], [q][cstride]
  const double *etaT;
  double lambda;
  const double *src_u, *src_p;
  double *slots_u, *slots_p;
  // Newton-derivative tables (Stokes operator, active level): symmetric tensors unrolled
  // (diagonal, then 01, 02, 12): [q][NS][cstride]; pressure factor [q][cstride]
  const double *nw_strain, *nw_dEde, *nw_dEdp;
  int nw_on, nw_symmetrize;
  // prescribed dilation (Stokes operator, active level): LHS term [q][cstride]; with Newton
  // derivatives d/d(strain rate) [q][NS][cstride] and d/dp [q][cstride] (or nullptr)
  const double *dil_lhs, *dil_de, *dil_dp;
  // 1: read the source WITHOUT the constraints (FEEvaluation::read_dof_values_plain): the right-hand
  // side correction for inhomogeneous boundary values (local_smoothing.cc:813, 817)
  int plain;
};

enum { KOP_STOKES = 0, KOP_A = 1, KOP_BT = 2 };

// compile-time ablation switches for tools/ablate.sh (never defined in the product build)
#ifndef AGMG_ABLATE
#define AGMG_ABLATE 0
#endif

#if AGMG_ABLATE & 32
#define AGMG_CELLS_WAVES 3
#else
#define AGMG_CELLS_WAVES 2
#endif

template <int DIM> struct KDims {
  static constexpr int NV = (DIM == 3) ? 27 : 9;
  static constexpr int NP = (DIM == 3) ? 8 : 4;
  static constexpr int CPW = 64 / DIM;        // cells per wavefront
  static constexpr int LROW = 64 * DIM + 2;   // padded row of the gradient-exchange tile
  static constexpr int XB = 3 * LROW;         // doubles of exchange tile per wavefront
};

// How the lanes that share a cell hand each other their velocity-gradient rows (sym grad u needs
// du_d/dx_c from the lane of component d).
//   put(x, c, dd, v): this lane (component c) publishes d u_c / d x_dd at q-point x of the line
//   get(x, from, dir): d u_from / d x_dir of this lane's cell
//   publish(): everything put is visible to the cell's lanes; release(): the gets are done
// WaveXch: the DIM lanes of a cell sit side by side in ONE wavefront (lane = DIM * j + c); a
// wave-private LDS tile [3][LROW], ordered by wavefront-scope fences (LDS is in-order per wave).
template <int DIM> struct WaveXch {
  double (*xb)[KDims<DIM>::LROW];
  int lane, j;
  __device__ __forceinline__ void put(int x, int c, int dd, double v) const { (void)c; xb[x][lane * DIM + dd] = v; }
  __device__ __forceinline__ double get(int x, int from, int dir) const { return xb[x][(j * DIM + from) * DIM + dir]; }
  __device__ __forceinline__ void publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ void release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// One cell per DIM lanes: gather, sum-factorised evaluate, q-point operation, integrate.
// Lane (j, c) handles component c of cell `cell`; G receives the 3^DIM velocity
// contributions of that component, pout this lane's share of the 2^DIM pressure rows (Stokes).
// ex: the exchange of the velocity gradient between the lanes of a cell (WaveXch).
// eta_tile (ETA_LDS): wave-private LDS copy [q][CPW] of the per-q viscosities of the wave's cells.
// GATHER: read the cell's velocity values from global memory here; otherwise the caller has
// filled u[] (the block kernel gathers from an LDS copy of the block's nodes).
// GEN: general (MappingQ) geometry: full inverse Jacobian and JxW per q-point instead of cell extents.
// NEWTON (Stokes only, WaveXch only): the optional terms of StokesOperator::local_apply, selected at
// run time by A.nw_on (Newton derivatives, matrix_free_operators.cc:196-213, 249-258: three cell-wide
// sums over the q-points first, then an extra stress at every q-point) and A.dil_lhs (prescribed
// dilation, :230-234, 259-268: extra terms in the pressure rows).
template <int DIM, int OP, bool COMP, bool PERQ, bool ETA_LDS, bool GATHER, bool GEN, bool NEWTON = false, class EX = WaveXch<DIM>,
          bool PIPE = false>
__device__ __forceinline__ void cell_compute(const CellArgs &A, const int cell, const int c, const int j,
                                             const int lane, EX &ex, double *G,
                                             double *pout, const double *eta_tile, double *u,
                                             const int eta_stride = KDims<DIM>::CPW) {
  constexpr int NV = KDims<DIM>::NV;
  constexpr int NP = KDims<DIM>::NP;
  constexpr int NPLP = NP / 2;  // pressure nodes per plane
  constexpr bool HAS_U = (OP != KOP_BT);
  constexpr bool HAS_P = (OP != KOP_A);
  const size_t cs = (size_t)A.cstride;

  // geometry (MappingCartesian): ih[d] = 1 / (spacing * h_d)
  double ih[DIM], det = 1.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const double h = GEN ? 1.0 : A.hT[d * cs + cell];
    det *= h;
    ih[d] = INV_S / h;
  }
  double sel[DIM];   // 1 on this lane's own component
#pragma unroll
  for (int d = 0; d < DIM; ++d) sel[d] = (d == c) ? 1.0 : 0.0;
  (void)sel;
  const double eta_c = PERQ ? 1.0 : A.etaT[cell];
  double cw[DIM], lamc[DIM], ptc[DIM], compc[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    cw[d] = ih[d] * det * eta_c;
    lamc[d] = (d == c) ? A.lambda * det * ih[d] : 0.0;
    ptc[d] = (d == c) ? -0.5 * A.lambda * det : 0.0;
    compc[d] = (d == c) ? (2.0 / 3.0) * cw[d] : 0.0;
  }
  (void)lamc; (void)ptc; (void)compc;
  // Cartesian, incompressible, no Newton terms: the gradient rows stay in reference space (no 1/h scaling of
  // the 9 derivatives of a line); the stress coefficients carry the geometry instead:
  //   flux_dd = f (cw_dd / h_dd) dref u_c/d xi_dd + f (cw_dd / h_c) dref u_dd/d xi_c
  constexpr bool FOLD = !GEN && !COMP && !NEWTON && HAS_U;
  double Pc[DIM], Qc[DIM], ptcF[DIM];
  {
    double ihc = ih[0];
#pragma unroll
    for (int d = 1; d < DIM; ++d) ihc = (d == c) ? ih[d] : ihc;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      Pc[d] = cw[d] * ih[d];
      Qc[d] = cw[d] * ihc;
      ptcF[d] = (d == c) ? -A.lambda * det * ih[d] : 0.0;
    }
  }
  (void)Pc; (void)Qc; (void)ptcF;

  // gather (constrained DoFs read as zero)
  if (HAS_U && GATHER) {
#pragma unroll
    for (int a = 0; a < NV; ++a) {
#if AGMG_ABLATE & 4
      const uint32_t idx = (uint32_t)(cell + a);
#else
      const uint32_t idx = A.cnvT[a * cs + cell];
#endif
#if AGMG_ABLATE & 2
      const double v = (double)(idx & 1023u);
#else
      const double v = A.src_u[(size_t)(idx & NODE_MASK) * DIM + c];
#endif
      u[a] = (((idx >> (29 + c)) & 1u) && !A.plain) ? 0.0 : v;
    }
  }
  if (HAS_U) {
    // nodal values -> values at the Gauss points, axis by axis
#pragma unroll
    for (int l = 0; l < NV / 3; ++l) interp3(u[3 * l], u[3 * l + 1], u[3 * l + 2]);
#pragma unroll
    for (int l = 0; l < NV / 3; ++l) {
      const int b = (l % 3) + 9 * (l / 3);
      interp3(u[b], u[b + 3], u[b + 6]);
    }
    if (DIM == 3) {
#pragma unroll
      for (int l = 0; l < 9; ++l) interp3(u[l], u[l + 9], u[l + 18]);
    }
  }
  // pressure: nodal values -> per plane values along the last axis
  double pz[3][NPLP];
  if (HAS_P) {
    double pin[NP];
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      const uint32_t idx = A.cnpT[b * cs + cell];
      const double v = A.src_p[idx & NODE_MASK];
      pin[b] = (((idx >> 29) & 1u) && !A.plain) ? 0.0 : v;
    }
#pragma unroll
    for (int z = 0; z < 3; ++z)
#pragma unroll
      for (int b = 0; b < NPLP; ++b) pz[z][b] = fma(n1(z, 0), pin[b], n1(z, 1) * pin[NPLP + b]);
  }

  // ---- Newton: S1 = sum_q dEde_q : sym grad u_q, S3 = sum_q strain_q : sym grad u_q (a symmetric
  // table tensor contracted with sym grad u only needs this lane's own gradient row; the rows of the
  // DIM lanes of the cell are summed through the LDS tile), S2 = sum_q dEdp_q p_q
  constexpr int NS = DIM * (DIM + 1) / 2;
  int symidx[DIM];   // unrolled index of tensor entry (c, dd)
#pragma unroll
  for (int dd = 0; dd < DIM; ++dd) symidx[dd] = (dd == c) ? c : c + dd + (DIM == 3 ? 2 : 1);
  (void)symidx;
  double S1 = 0.0, S2 = 0.0, S3 = 0.0;
  if constexpr (NEWTON) { if (A.nw_on) {
    auto xb = ex.xb;
#pragma unroll
    for (int L = 0; L < NV / 3; ++L) {
      __builtin_amdgcn_sched_barrier(0);
      const int qy = (DIM == 3) ? L % 3 : L, qz = (DIM == 3) ? L / 3 : 0;
      const int base = 3 * L;
      double d[DIM][3], pq[3];
      {
        double t0, t1;
        if (DIM == 3) {
          t0 = fma(n1(qy, 0), pz[qz][0], n1(qy, 1) * pz[qz][2]);
          t1 = fma(n1(qy, 0), pz[qz][1], n1(qy, 1) * pz[qz][3]);
        } else {
          t0 = pz[qy][0];
          t1 = pz[qy][1];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) pq[x] = fma(n1(x, 0), t0, n1(x, 1) * t1);
      }
      deriv3(u[base], u[base + 1], u[base + 2], d[0][0], d[0][1], d[0][2]);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        double s1 = 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (dcol(qy, m) != 0.0) s1 = fma(dcol(qy, m), u[9 * qz + 3 * m + x], s1);
        d[1][x] = s1;
        if (DIM == 3) {
          double s2 = 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m)
            if (dcol(qz, m) != 0.0) s2 = fma(dcol(qz, m), u[9 * m + 3 * qy + x], s2);
          d[DIM - 1][x] = s2;
        }
      }
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const size_t q = (size_t)(base + x);
        double g[DIM];
#pragma unroll
        for (int dd = 0; dd < DIM; ++dd) {
          if (GEN) {
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < DIM; ++e) acc = fma(d[e][x], A.invJT[(q * (DIM * DIM) + e * DIM + dd) * cs + cell], acc);
            g[dd] = INV_S * acc;
          } else {
            g[dd] = d[dd][x] * ih[dd];
          }
        }
#pragma unroll
        for (int dd = 0; dd < DIM; ++dd) {
          S1 = fma(A.nw_dEde[(q * NS + symidx[dd]) * cs + cell], g[dd], S1);
          if (A.nw_symmetrize) S3 = fma(A.nw_strain[(q * NS + symidx[dd]) * cs + cell], g[dd], S3);
        }
        S2 = fma(A.nw_dEdp[q * cs + cell], pq[x], S2);
      }
      // keep the lines apart (as in the main loop): without the pins the table loads of all
      // lines are hoisted to the top and spill
      asm volatile("" : "+v"(S1), "+v"(S2), "+v"(S3) : : "memory");
    }
    xb[0][lane] = S1;
    xb[1][lane] = S3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    S1 = 0.0;
    S3 = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) { S1 += xb[0][j * DIM + k]; S3 += xb[1][j * DIM + k]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } }
#pragma unroll
  for (int a = 0; a < NV; ++a) G[a] = 0.0;
#pragma unroll
  for (int b = 0; b < NP; ++b) pout[b] = 0.0;

  // one batch = one line of three Gauss points along axis 0.  pre: reference derivatives of the line,
  // physical gradient row of this lane's component, handed to the exchange; post: stress from the
  // exchanged gradient, transposed derivatives into G.
  constexpr int NL = NV / 3;
  using Jac = double[GEN ? 3 : 1][DIM][DIM];   // inverse Jacobians of the line's q-points (general geometry)
  auto pre = [&](const int L, double (&d)[DIM][3], Jac &Ji) {
    const int qy = (DIM == 3) ? L % 3 : L, qz = (DIM == 3) ? L / 3 : 0;
    const int base = 3 * L;
    if (GEN && !HAS_U && !PIPE) {
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int e = 0; e < DIM; ++e)
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd)
            Ji[x][e][dd] = A.invJT[((size_t)(base + x) * (DIM * DIM) + e * DIM + dd) * cs + cell];
    }
    if (HAS_U) {
      // reference-space derivatives (times the Gauss point spacing)
      deriv3(u[base], u[base + 1], u[base + 2], d[0][0], d[0][1], d[0][2]);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        double s1 = 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (dcol(qy, m) != 0.0) s1 = fma(dcol(qy, m), u[9 * qz + 3 * m + x], s1);
        d[1][x] = s1;
        if (DIM == 3) {
          double s2 = 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m)
            if (dcol(qz, m) != 0.0) s2 = fma(dcol(qz, m), u[9 * m + 3 * qy + x], s2);
          d[DIM - 1][x] = s2;
        }
      }
      // physical gradient row of this lane's component -> LDS
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        if (GEN) {
          double g[DIM];
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd) {
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < DIM; ++e) {
              Ji[x][e][dd] = A.invJT[((size_t)(base + x) * (DIM * DIM) + e * DIM + dd) * cs + cell];
              acc = fma(d[e][x], Ji[x][e][dd], acc);
            }
            g[dd] = INV_S * acc;
          }
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd) d[dd][x] = g[dd];
        } else if (!FOLD) {
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd) d[dd][x] *= ih[dd];
        }
#pragma unroll
        for (int dd = 0; dd < DIM; ++dd) ex.put(x, c, dd, d[dd][x]);
      }
    }
  };
  auto post = [&](const int L, double (&d)[DIM][3], Jac &Ji) {
    const int qy = (DIM == 3) ? L % 3 : L, qz = (DIM == 3) ? L / 3 : 0;
    const int base = 3 * L;
    double pq[3], pt[3];
    if (HAS_P) {
      // Q1 values on this line
      double t0, t1;
      if (DIM == 3) {
        t0 = fma(n1(qy, 0), pz[qz][0], n1(qy, 1) * pz[qz][2]);
        t1 = fma(n1(qy, 0), pz[qz][1], n1(qy, 1) * pz[qz][3]);
      } else {
        t0 = pz[qy][0];
        t1 = pz[qy][1];
      }
#pragma unroll
      for (int x = 0; x < 3; ++x) pq[x] = fma(n1(x, 0), t0, n1(x, 1) * t1);
    }
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const double wq = (DIM == 3) ? wq1(x) * wq1(qy) * wq1(qz) : wq1(x) * wq1(qy);
      double etav = 1.0;
#if !(AGMG_ABLATE & 8)
      if (PERQ) etav = ETA_LDS ? eta_tile[(base + x) * eta_stride + j] : A.etaT[(size_t)(base + x) * cs + cell];
#endif
      const double f = wq * etav;
      if (HAS_U) {
        // column c of the velocity gradient: du_dd/dx_c lives in the lane of component dd
        double t[DIM], grow[DIM];   // grow: this lane's own physical gradient row (dilation derivative)
#pragma unroll
        for (int dd = 0; dd < DIM; ++dd) {
          grow[dd] = d[dd][x];
          const double got = ex.get(x, dd, c);
          t[dd] = FOLD ? fma(Pc[dd], d[dd][x], Qc[dd] * got) : d[dd][x] + got;
        }
        (void)grow;
        double fdiv = 0.0, div = 0.0;
        if (COMP) {
#pragma unroll
          for (int e = 0; e < DIM; ++e) div += ex.get(x, e, e);
          fdiv = f * div;
        }
        (void)fdiv;
        if (GEN) {
          // JxW carries the quadrature weight
          const double jxw = A.jxwT[(size_t)(base + x) * cs + cell];
          const double etaq = eta_c * etav;
          double tau[DIM];
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd) {
            tau[dd] = etaq * t[dd];
            if (COMP) tau[dd] = fma(-sel[dd] * (2.0 / 3.0) * etaq, div, tau[dd]);
            if (HAS_P) tau[dd] = fma(-sel[dd] * A.lambda, pq[x], tau[dd]);
            if (NEWTON && A.nw_on) {
              const size_t qn = ((size_t)(base + x) * NS + symidx[dd]) * cs + cell;
              const double e_cd = A.nw_strain[qn], d_cd = A.nw_dEde[qn];
              tau[dd] += (A.nw_symmetrize ? e_cd * S1 + d_cd * S3 : 2.0 * e_cd * S1) + 2.0 * e_cd * S2;
            }
          }
#pragma unroll
          for (int e = 0; e < DIM; ++e) {
            double acc = 0.0;
#pragma unroll
            for (int dd = 0; dd < DIM; ++dd) {
              // Stokes runs out of registers holding the line's Jacobians across the exchange: reload (L1 hit)
              const double jv = (PIPE || (OP == KOP_STOKES && DIM == 3))
                                    ? A.invJT[((size_t)(base + x) * (DIM * DIM) + e * DIM + dd) * cs + cell]
                                    : Ji[x][e][dd];
              acc = fma(jv, tau[dd], acc);
            }
            d[e][x] = (jxw * INV_S) * acc;
          }
          if (OP == KOP_STOKES) {
            double s = 0.0;
#pragma unroll
            for (int dd = 0; dd < DIM; ++dd) s = fma(sel[dd], t[dd], s);
            pt[x] = (-0.5 * A.lambda * jxw) * s;
          }
          if (NEWTON && A.dil_lhs) {   // pressure rows of the prescribed dilation (lane 0 carries the p terms)
            const size_t q = (size_t)(base + x);
            double extra = (c == 0) ? -A.lambda * A.lambda * A.dil_lhs[q * cs + cell] * pq[x] : 0.0;
            if (A.nw_on && A.dil_de) {
              double s = 0.0;
#pragma unroll
              for (int dd = 0; dd < DIM; ++dd) s = fma(A.dil_de[(q * NS + symidx[dd]) * cs + cell], grow[dd], s);
              extra += A.lambda * s;
              if (c == 0) extra += A.lambda * A.lambda * A.dil_dp[q * cs + cell] * pq[x];
            }
            pt[x] = fma(jxw, extra, pt[x]);
          }
        } else {
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd) {
            double F = FOLD ? f * t[dd] : (cw[dd] * f) * t[dd];
            if (COMP) F = fma(-compc[dd], fdiv, F);
            if (HAS_P) F = fma(-lamc[dd], wq * pq[x], F);
            if (NEWTON && A.nw_on) {
              const size_t qn = ((size_t)(base + x) * NS + symidx[dd]) * cs + cell;
              const double e_cd = A.nw_strain[qn], d_cd = A.nw_dEde[qn];
              const double nw = (A.nw_symmetrize ? e_cd * S1 + d_cd * S3 : 2.0 * e_cd * S1) + 2.0 * e_cd * S2;
              F = fma(ih[dd] * det * wq, nw, F);
            }
            d[dd][x] = F;
          }
          if (OP == KOP_STOKES) {
            double s = 0.0;
#pragma unroll
            for (int dd = 0; dd < DIM; ++dd) s = FOLD ? fma(ptcF[dd], grow[dd], s) : fma(ptc[dd], t[dd], s);
            pt[x] = wq * s;
            if (NEWTON && A.dil_lhs) {
              const size_t q = (size_t)(base + x);
              double extra = (c == 0) ? -A.lambda * A.lambda * A.dil_lhs[q * cs + cell] * pq[x] : 0.0;
              if (A.nw_on && A.dil_de) {
                double sd = 0.0;
#pragma unroll
                for (int dd = 0; dd < DIM; ++dd) sd = fma(A.dil_de[(q * NS + symidx[dd]) * cs + cell], grow[dd], sd);
                extra += A.lambda * sd;
                if (c == 0) extra += A.lambda * A.lambda * A.dil_dp[q * cs + cell] * pq[x];
              }
              pt[x] = fma(wq * det, extra, pt[x]);
            }
          }
        }
      } else if (GEN) {
        const double jxw = A.jxwT[(size_t)(base + x) * cs + cell];
#pragma unroll
        for (int e = 0; e < DIM; ++e) {
          double acc = 0.0;
#pragma unroll
          for (int dd = 0; dd < DIM; ++dd)
            acc = fma(PIPE ? A.invJT[((size_t)(base + x) * (DIM * DIM) + e * DIM + dd) * cs + cell] : Ji[x][e][dd], sel[dd], acc);
          d[e][x] = (-A.lambda * jxw * INV_S) * pq[x] * acc;
        }
      } else {
#pragma unroll
        for (int dd = 0; dd < DIM; ++dd) d[dd][x] = -lamc[dd] * (wq * pq[x]);
      }
    }
    if (HAS_U) ex.release();
    // transposed derivatives into G
    deriv3T(d[0][0], d[0][1], d[0][2], G[base], G[base + 1], G[base + 2]);
#pragma unroll
    for (int x = 0; x < 3; ++x) {
#pragma unroll
      for (int m = 0; m < 3; ++m)
        if (dcol(qy, m) != 0.0) G[9 * qz + 3 * m + x] = fma(dcol(qy, m), d[1][x], G[9 * qz + 3 * m + x]);
      if (DIM == 3) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (dcol(qz, m) != 0.0) G[9 * m + 3 * qy + x] = fma(dcol(qz, m), d[DIM - 1][x], G[9 * m + 3 * qy + x]);
      }
    }
    // pin the updated accumulators here: without this the compiler sinks the flux / G arithmetic
    // of all lines below the LDS exchanges of all lines and spills the exchanged values
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      asm volatile("" : "+v"(G[base + x]) : : "memory");
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (m != qy && dcol(qy, m) != 0.0) asm volatile("" : "+v"(G[9 * qz + 3 * m + x]) : : "memory");
        if (DIM == 3 && m != qz && dcol(qz, m) != 0.0) asm volatile("" : "+v"(G[9 * m + 3 * qy + x]) : : "memory");
      }
    }
    if (OP == KOP_STOKES) {
      // pressure test function: Gauss points of the line -> Q1 nodes
      double r[2];
#pragma unroll
      for (int bx = 0; bx < 2; ++bx) r[bx] = fma(n1(0, bx), pt[0], fma(n1(1, bx), pt[1], n1(2, bx) * pt[2]));
      if (DIM == 3) {
#pragma unroll
        for (int bz = 0; bz < 2; ++bz)
#pragma unroll
          for (int by = 0; by < 2; ++by)
#pragma unroll
            for (int bx = 0; bx < 2; ++bx)
              pout[4 * bz + 2 * by + bx] = fma(n1(qz, bz) * n1(qy, by), r[bx], pout[4 * bz + 2 * by + bx]);
      } else {
#pragma unroll
        for (int by = 0; by < 2; ++by)
#pragma unroll
          for (int bx = 0; bx < 2; ++bx) pout[2 * by + bx] = fma(n1(qy, by), r[bx], pout[2 * by + bx]);
      }
#pragma unroll
      for (int b = 0; b < NP; ++b) asm volatile("" : "+v"(pout[b]) : : "memory");
    }
  };
  if (!PIPE) {
#pragma unroll
    for (int L = 0; L < NL; ++L) {
#if !(AGMG_ABLATE & 16)
      __builtin_amdgcn_sched_barrier(0);
#endif
      double d[DIM][3];
      Jac Ji;
      pre(L, d, Ji);
      if (HAS_U) ex.publish();
      post(L, d, Ji);
    }
  } else {
    // software pipeline (two exchange tiles): line L + 1's gradient rows are computed and handed over
    // between the barrier that publishes line L and the use of line L's exchanged values, so the
    // barrier skew and the LDS read latency hide under independent arithmetic
    double dbuf[2][DIM][3];
    Jac Jd;
    pre(0, dbuf[0], Jd);
#pragma unroll
    for (int L = 0; L < NL; ++L) {
      __builtin_amdgcn_sched_barrier(0);
      if (HAS_U) ex.publish();
      if (L + 1 < NL) pre(L + 1, dbuf[(L + 1) & 1], Jd);
      post(L, dbuf[L & 1], Jd);
    }
  }
  // Gauss points -> nodes (transposed interpolation), axis by axis
#pragma unroll
  for (int l = 0; l < NV / 3; ++l) interp3T(G[3 * l], G[3 * l + 1], G[3 * l + 2]);
#pragma unroll
  for (int l = 0; l < NV / 3; ++l) {
    const int b = (l % 3) + 9 * (l / 3);
    interp3T(G[b], G[b + 3], G[b + 6]);
  }
  if (DIM == 3) {
#pragma unroll
    for (int l = 0; l < 9; ++l) interp3T(G[l], G[l + 9], G[l + 18]);
  }
}


// ---- node-owner pull + vector epilogue -----------------------------------------------------
struct PullArgs {
  const uint32_t *ptr, *list;
  const double *slots;
  const uint8_t *cmask;
  int node0, n_range;       // nodes [node0, node0 + n_range)
  uint32_t recv_base;       // list entries >= recv_base address recv[(entry - recv_base) * NC + c]
  const double *recv;       // remote partial sums (ghost exchange), or nullptr
  double *ghost_out;        // != nullptr: raw sums to ghost_out[(node - node0) * NC + c], no epilogue
  EpiArgs e;
};

// vector epilogue shared by the pull, block and fix-up kernels; false = leave dst untouched
__device__ __forceinline__ bool epi_value(const EpiArgs &e, const size_t t, const double sum, const bool con,
                                          double &out) {
  switch (e.epi) {
    case EPI_STORE:
      out = con ? (e.x ? e.x[t] : 0.0) : sum;
      return true;
    case EPI_RESID:
      out = e.b[t] - (con ? (e.x ? e.x[t] : 0.0) : sum);
      return true;
    case EPI_CHEB: {
      const double x = e.x[t];
      const double Ax = con ? x : sum;
      double v = fma(e.f2 * e.dinv[t], e.b[t] - Ax, (1.0 + e.f1) * x);
      if (e.xold) v = fma(-e.f1, e.xold[t], v);
      out = v;
      return true;
    }
    case EPI_ADD:
      if (con) return false;
      out = e.dst[t] + sum;
      return true;
    case EPI_SUB:
      if (con) return false;
      out = e.dst[t] - sum;
      return true;
    case EPI_DIAG:
      out = 1.0 / (con ? 1.0 : sum);
      return true;
    default:
      out = con ? 1.0 : sum;
      return true;
  }
}

// NC: components per node (one thread each); NPARTS: partial values per slot summed by the thread
template <int NC, int NPARTS>
__device__ __forceinline__ void pull_body(const PullArgs &P, const unsigned block) {
  const long tl = (long)block * 256 + threadIdx.x;
  if (tl >= (long)P.n_range * NC) return;
  const int nl = (int)(tl / NC), c = (int)(tl - (long)nl * NC);
  const int node = P.node0 + nl;
  const uint32_t s = P.ptr[node], en = P.ptr[node + 1];
  double sum = 0.0;
  for (uint32_t k = s; k < en; ++k) {
    const uint32_t ent = P.list[k];
    if (ent >= P.recv_base) {
      sum += P.recv[(size_t)(ent - P.recv_base) * NC + c];
    } else {
      const size_t base = (size_t)ent * (NC * NPARTS) + c * NPARTS;
#pragma unroll
      for (int part = 0; part < NPARTS; ++part) sum += P.slots[base + part];
    }
  }
  if (P.ghost_out) { P.ghost_out[tl] = sum; return; }
  const size_t t = (size_t)node * NC + c;
  const bool con = P.cmask ? ((P.cmask[node] >> c) & 1u) : false;
  double out;
  if (epi_value(P.e, t, sum, con, out)) P.e.dst[t] = out;
}

// block index -> work item such that the 8 groups of workgroups {blockIdx % 8 = x} (one XCD, one L2
// each) take contiguous runs of items; a bijection on [0, nb)
__device__ __forceinline__ unsigned xcd_contiguous(const unsigned b, const unsigned nb) {
  const unsigned q = nb >> 3, rem = nb & 7u, x = b & 7u, i = b >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i;
}

inline CellArgs make_cell_args(const DeviceLevel &L, const CoeffData &cd, const double *su, const double *sp, double *slu,
                   double *slp) {
  CellArgs a;
  a.n_cells = L.n_cells;
  a.cstride = L.cstride;
  a.cnvT = L.cnvT;
  a.cnpT = L.cnpT;
  a.hT = L.cell_hT;
  a.invJT = L.invJT;
  a.jxwT = L.jxwT;
  a.etaT = cd.etaT;
  a.lambda = cd.lambda;
  a.src_u = su;
  a.src_p = sp;
  a.slots_u = slu;
  a.slots_p = slp;
  a.nw_strain = cd.nw_strain;
  a.nw_dEde = cd.nw_dEde;
  a.nw_dEdp = cd.nw_dEdp;
  a.nw_symmetrize = cd.nw_symmetrize;
  a.nw_on = cd.newton ? 1 : 0;
  a.dil_lhs = cd.dilation ? cd.dil_lhs : nullptr;
  a.dil_de = cd.dil_de;
  a.dil_dp = cd.dil_dp;
  a.plain = 0;
  return a;
}


}  // namespace
}  // namespace agmg
