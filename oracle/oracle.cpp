// oracle.cpp -- CPU restatement of ASPECT's matrix-free block-GMG Stokes solve.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (aspect_amd/, the HIP
// library) may import, call, link or execute this file; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
// checker / CPU baseline ("port").
//
// Parity pin: the reference (tjhei/aspect 3.1.0-pre) cannot be built here
// (deal.II >= 9.5 + p4est + Trilinos are absent) and holds no
// operator-level golden vectors; this oracle is pinned by the reference's
// integration goldens (iteration counts, DoF counts, viscosity ranges;
// tests/golden/*.json, tests/test_oracle_goldens.py).  Operator-level parity
// is pinned internally: the sum-factorised operators below must agree with an
// independent dense cell-matrix assembly of the bilinear form
// source/simulator/assemblers/stokes.cc:463-479.
//
// What follows which reference lines:
//   cell integrals      source/simulator/solver/matrix_free_operators.cc
//                         :171-278 (Stokes) :399-432 (B^T) :539-576 (mass) :705-740 (A)
//   diagonals           matrix_free_operators.cc:592-639, 799-829
//   wrapper semantics   deal.II MatrixFreeOperators::Base::vmult
//   Chebyshev           deal.II PreconditionChebyshev, constants
//                         stokes_matrix_free_local_smoothing.cc:926-937, 954-965
//   multigrid           deal.II Multigrid::level_v_step / mg::SmootherRelaxation /
//                         MGTransferMF, wiring local_smoothing.cc:1040-1058
//   block preconditioner include/aspect/simulator/solver/stokes_matrix_free.h:237-354
//   solve               local_smoothing.cc:1119-1227 (tolerance), :1379-1495 (phases)
//   Krylov              deal.II SolverGMRES (right prec.), SolverFGMRES, SolverIDR,
//                         SolverCG, SolverBicgstab
#include "../include/agmg.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <map>
#endif

namespace {

using Vec = std::vector<double>;
std::string g_error;

// ------------------------------------------------------------------ element
// Q2 nodes {0,1/2,1}, Q1 {0,1}, Gauss-3 on the unit interval.
struct Tables {
  double xq[3], wq[3];
  double N2[3][3], D2[3][3], N1[3][2];  // [q][i]
  double P2[5][3];                      // Q2 embedding rows at x = 0,1/4,1/2,3/4,1
  double P1[3][2];
  Tables() {
    const double s = std::sqrt(0.6);
    xq[0] = 0.5 - 0.5 * s; xq[1] = 0.5; xq[2] = 0.5 + 0.5 * s;
    wq[0] = 5. / 18.; wq[1] = 8. / 18.; wq[2] = 5. / 18.;
    for (int q = 0; q < 3; ++q) {
      const double x = xq[q];
      N2[q][0] = (1 - x) * (1 - 2 * x); N2[q][1] = 4 * x * (1 - x); N2[q][2] = x * (2 * x - 1);
      D2[q][0] = 4 * x - 3; D2[q][1] = 4 - 8 * x; D2[q][2] = 4 * x - 1;
      N1[q][0] = 1 - x; N1[q][1] = x;
    }
    for (int r = 0; r < 5; ++r) {
      const double x = 0.25 * r;
      P2[r][0] = (1 - x) * (1 - 2 * x); P2[r][1] = 4 * x * (1 - x); P2[r][2] = x * (2 * x - 1);
    }
    for (int r = 0; r < 3; ++r) { P1[r][0] = 1 - 0.5 * r; P1[r][1] = 0.5 * r; }
  }
};
const Tables T;

inline int ipow(int b, int e) { int r = 1; while (e--) r *= b; return r; }

// apply a 1-D matrix M[nout][nin] along direction dir of a tensor (x fastest)
// with input extents n[3]; output extents equal except n[dir] = nout.
void apply1d(const double *M, int nout, int nin, int dir, const int n[3],
             const double *in, double *out) {
  int no[3] = {n[0], n[1], n[2]};
  no[dir] = nout;
  const int si[3] = {1, n[0], n[0] * n[1]};
  const int so[3] = {1, no[0], no[0] * no[1]};
  for (int k = 0; k < no[2]; ++k)
    for (int j = 0; j < no[1]; ++j)
      for (int i = 0; i < no[0]; ++i) {
        const int o[3] = {i, j, k};
        int base = 0;
        for (int d = 0; d < 3; ++d) if (d != dir) base += o[d] * si[d];
        double s = 0;
        for (int a = 0; a < nin; ++a) s += M[o[dir] * nin + a] * in[base + a * si[dir]];
        out[i * so[0] + j * so[1] + k * so[2]] = s;
      }
}
// transpose application: out[nin-extent] = M^T in[nout-extent]
void apply1d_T(const double *M, int nout, int nin, int dir, const int n_out_ext[3],
               const double *in, double *out) {
  // in has extents n_out_ext with n[dir]=nout ; out has n[dir]=nin
  double Mt[25];
  for (int a = 0; a < nin; ++a) for (int b = 0; b < nout; ++b) Mt[a * nout + b] = M[b * nin + a];
  apply1d(Mt, nin, nout, dir, n_out_ext, in, out);
}

// ------------------------------------------------------------------ level
struct CellData {
  bool set = false;
  int is_compressible = 0, newton = 0, dilation = 0, symmetrize = 0;
  double lambda = 1.0;
  int nqv = 1;
  Vec eta;  // [n_cells][nqv]
  // Newton tables (active level, local_smoothing.cc:447-663): symmetric tensors unrolled as deal.II
  // does, diagonal first then (0,1), (0,2), (1,2); [n_cells][nq][dim(dim+1)/2] and [n_cells][nq]
  Vec strain, dEde, dEdp;
  // prescribed dilation (active level): LHS term [n_cells][nq]; with Newton its derivatives
  Vec dil_lhs, dil_de, dil_dp;
  // free-surface stabilisation faces (active level; apply_stabilization_free_surface_faces,
  // free_surface_stabilization_term_table: local_smoothing.cc:665-756)
  int n_faces = 0;
  std::vector<uint32_t> face_cell;
  std::vector<uint8_t> face_no;        // deal.II face number 2*d + side
  Vec face_stab, face_normal, face_jxw;  // [face][q][dim], [face][q][dim] or empty, [face][q] or empty
};
inline int symc(int dim, int a, int b) {
  if (a == b) return a;
  if (a > b) std::swap(a, b);
  return dim == 2 ? 2 : (a == 0 ? (b == 1 ? 3 : 4) : 5);
}

struct Level {
  int dim = 3, n_cells = 0, n_nodes_v = 0, n_nodes_p = 0;
  int nv = 27, np = 8, nq = 27;
  std::vector<uint32_t> cnv, cnp;
  std::vector<uint8_t> cmv, cmp;
  int geom = AGMG_GEOM_CARTESIAN;
  Vec cell_h, invJ, jxw;
  std::vector<int32_t> parent;
  std::vector<uint8_t> child;
  CellData cd;
  Vec inv_diag_A, inv_diag_M, diag_M;
  // hanging-node constraints (agmg_level_desc): CSR over the hanging nodes of the two spaces
  std::vector<uint32_t> hang_n[2], hang_m[2];     // [0] velocity nodes, [1] pressure nodes
  std::vector<int32_t> hang_ptr[2];
  Vec hang_w[2];
  bool has_hanging() const { return !hang_n[0].empty() || !hang_n[1].empty(); }
  // local smoothing on adaptive meshes: cmv / cmp above hold boundary + refinement-edge rows (the level operator of
  // MatrixFreeOperators::Base: identity on both), bnd_v / bnd_p the boundary rows alone (transfers, interface operators)
  std::vector<uint8_t> bnd_v, bnd_p, edge_v, edge_p;
  bool has_edge = false;
  std::vector<uint32_t> copy_act[2], copy_lev[2];   // MGLevelGlobalTransfer::copy_indices, [0] velocity, [1] pressure nodes
  bool set = false;
  int ndv() const { return dim * n_nodes_v; }
  int ndp() const { return n_nodes_p; }
  bool constrained_v(int node, int c) const { return (cmv[node] >> c) & 1; }
  bool constrained_p(int node) const { return cmp[node] & 1; }
  bool boundary_v(int node, int c) const { return ((has_edge ? bnd_v[node] : cmv[node]) >> c) & 1; }
  bool boundary_p(int node) const { return (has_edge ? bnd_p[node] : cmp[node]) & 1; }
};

struct Geo {  // geometry at one q-point
  double invJ[3][3];
  double jxw;
};

template <int DIM>
inline void cell_geometry(const Level &L, int cell, int q, Geo &g) {
  if (L.geom == AGMG_GEOM_CARTESIAN) {
    double det = 1;
    for (int e = 0; e < DIM; ++e) for (int d = 0; d < DIM; ++d) g.invJ[e][d] = 0;
    for (int d = 0; d < DIM; ++d) { g.invJ[d][d] = 1.0 / L.cell_h[cell * DIM + d]; det *= L.cell_h[cell * DIM + d]; }
    double w = 1; int qq = q;
    for (int d = 0; d < DIM; ++d) { w *= T.wq[qq % 3]; qq /= 3; }
    g.jxw = w * det;
  } else {
    const double *p = &L.invJ[(size_t(cell) * L.nq + q) * DIM * DIM];
    for (int e = 0; e < DIM; ++e) for (int d = 0; d < DIM; ++d) g.invJ[e][d] = p[e * DIM + d];
    g.jxw = L.jxw[size_t(cell) * L.nq + q];
  }
}

// reference-space gradients of a Q2 scalar at the q-points: g[e][q]
template <int DIM>
void q2_ref_grad(const double *u, double g[][27]) {
  const int n[3] = {3, 3, DIM == 3 ? 3 : 1};
  double a[27], b[27];
  for (int e = 0; e < DIM; ++e) {
    const double *src = u;
    double *bufs[2] = {a, b};
    int w = 0;
    for (int d = 0; d < DIM; ++d) {
      double *dst = (d == DIM - 1) ? g[e] : bufs[w];
      apply1d(d == e ? &T.D2[0][0] : &T.N2[0][0], 3, 3, d, n, src, dst);
      src = dst; w ^= 1;
    }
  }
}
// integrate: out[i] = sum_e sum_q dN_e[q][i] t[e][q]
template <int DIM>
void q2_ref_grad_T(const double t[][27], double *out) {
  const int n[3] = {3, 3, DIM == 3 ? 3 : 1};
  const int nn = DIM == 3 ? 27 : 9;
  for (int i = 0; i < nn; ++i) out[i] = 0;
  double a[27], b[27];
  for (int e = 0; e < DIM; ++e) {
    const double *src = t[e];
    double *bufs[2] = {a, b};
    int w = 0;
    for (int d = 0; d < DIM; ++d) {
      double *dst = bufs[w];
      apply1d_T(d == e ? &T.D2[0][0] : &T.N2[0][0], 3, 3, d, n, src, dst);
      src = dst; w ^= 1;
    }
    for (int i = 0; i < nn; ++i) out[i] += src[i];
  }
}
template <int DIM>
void q1_values(const double *p, double *v) {
  int n[3] = {2, 2, DIM == 3 ? 2 : 1};
  double a[27], b[27];
  const double *src = p;
  double *bufs[2] = {a, b};
  int w = 0;
  for (int d = 0; d < DIM; ++d) {
    double *dst = (d == DIM - 1) ? v : bufs[w];
    apply1d(&T.N1[0][0], 3, 2, d, n, src, dst);
    n[d] = 3; src = dst; w ^= 1;
  }
}
template <int DIM>
void q1_values_T(const double *v, double *out) {
  int n[3] = {3, 3, DIM == 3 ? 3 : 1};
  double a[27], b[27];
  const double *src = v;
  double *bufs[2] = {a, b};
  int w = 0;
  for (int d = 0; d < DIM; ++d) {
    double *dst = (d == DIM - 1) ? out : bufs[w];
    apply1d_T(&T.N1[0][0], 3, 2, d, n, src, dst);
    n[d] = 2; src = dst; w ^= 1;
  }
}

enum Mode { M_STOKES, M_A, M_BT };

// One cell of StokesOperator / ABlockOperator / BTBlockOperator::local_apply.
// uin[c][i], pin[i] are the gathered (constraint-resolved) local values;
// uout/pout receive the cell contributions.
template <int DIM>
void cell_apply(const Level &L, const CellData &cd, int cell, Mode mode,
                const double uin[][27], const double *pin, double uout[][27], double *pout) {
  const int NQ = DIM == 3 ? 27 : 9;
  double gref[3][3][27];   // [c][e][q]
  double pq[27];
  if (mode != M_BT)
    for (int c = 0; c < DIM; ++c) q2_ref_grad<DIM>(uin[c], gref[c]);
  if (mode != M_A) q1_values<DIM>(pin, pq);
  double tref[3][3][27];
  double ptest[27];
  const bool comp = cd.is_compressible || cd.dilation;
  constexpr int NS = DIM * (DIM + 1) / 2;
  // Newton derivative terms, matrix_free_operators.cc:196-213: three cell-wide sums over the
  // q-points (no weights), used at every q-point below (:249-258)
  double S1 = 0, S2 = 0, S3 = 0;
  const bool newton = cd.newton && mode == M_STOKES;
  if (newton)
    for (int q = 0; q < NQ; ++q) {
      Geo g; cell_geometry<DIM>(L, cell, q, g);
      const double *de = &cd.dEde[(size_t(cell) * NQ + q) * NS], *st = &cd.strain[(size_t(cell) * NQ + q) * NS];
      for (int c = 0; c < DIM; ++c)
        for (int d = 0; d < DIM; ++d) {
          double gcd = 0, gdc = 0;
          for (int e = 0; e < DIM; ++e) { gcd += gref[c][e][q] * g.invJ[e][d]; gdc += gref[d][e][q] * g.invJ[e][c]; }
          const double sym = 0.5 * (gcd + gdc);
          S1 += de[symc(DIM, c, d)] * sym;
          if (cd.symmetrize) S3 += st[symc(DIM, c, d)] * sym;
        }
      S2 += cd.dEdp[size_t(cell) * NQ + q] * pq[q];
    }
  for (int q = 0; q < NQ; ++q) {
    Geo g; cell_geometry<DIM>(L, cell, q, g);
    const double eta2 = 2.0 * cd.eta[size_t(cell) * cd.nqv + (cd.nqv == 1 ? 0 : q)];
    double tau[3][3] = {{0}};
    double div = 0;
    if (mode != M_BT) {
      double grad[3][3];
      for (int c = 0; c < DIM; ++c)
        for (int d = 0; d < DIM; ++d) {
          double s = 0;
          for (int e = 0; e < DIM; ++e) s += gref[c][e][q] * g.invJ[e][d];
          grad[c][d] = s;
        }
      for (int c = 0; c < DIM; ++c)
        for (int d = 0; d < DIM; ++d) tau[c][d] = eta2 * 0.5 * (grad[c][d] + grad[d][c]);
      for (int d = 0; d < DIM; ++d) div += grad[d][d];
      if (mode == M_A) {
        if (comp) {  // matrix_free_operators.cc:730-736
          double tr = 0; for (int d = 0; d < DIM; ++d) tr += tau[d][d];
          for (int d = 0; d < DIM; ++d) tau[d][d] -= (1.0 / 3.0) * tr;
        }
      } else if (comp) {  // :243-246
        for (int d = 0; d < DIM; ++d) tau[d][d] -= eta2 / 3.0 * div;
      }
    }
    if (mode != M_A) {
      for (int d = 0; d < DIM; ++d) tau[d][d] -= cd.lambda * pq[q];
      if (mode == M_STOKES) {
        double pterm = -cd.lambda * div;                                  // matrix_free_operators.cc:227-228
        if (cd.dilation) pterm -= cd.lambda * cd.lambda * cd.dil_lhs[size_t(cell) * NQ + q] * pq[q];   // :230-234
        if (cd.dilation && newton && !cd.dil_de.empty()) {                // :259-268
          const double *dd = &cd.dil_de[(size_t(cell) * NQ + q) * NS];
          double s = 0;
          for (int c = 0; c < DIM; ++c)
            for (int d = 0; d < DIM; ++d) {
              double gcd = 0, gdc = 0;
              for (int e = 0; e < DIM; ++e) { gcd += gref[c][e][q] * g.invJ[e][d]; gdc += gref[d][e][q] * g.invJ[e][c]; }
              s += dd[symc(DIM, c, d)] * 0.5 * (gcd + gdc);
            }
          pterm += (s + cd.dil_dp[size_t(cell) * NQ + q] * cd.lambda * pq[q]) * cd.lambda;
        }
        ptest[q] = pterm * g.jxw;
      }
    }
    if (newton) {
      const double *de = &cd.dEde[(size_t(cell) * NQ + q) * NS], *st = &cd.strain[(size_t(cell) * NQ + q) * NS];
      for (int c = 0; c < DIM; ++c)
        for (int d = 0; d < DIM; ++d) {
          const double e_cd = st[symc(DIM, c, d)], d_cd = de[symc(DIM, c, d)];
          tau[c][d] += (cd.symmetrize ? e_cd * S1 + d_cd * S3 : 2.0 * e_cd * S1) + 2.0 * e_cd * S2;
        }
    }
    for (int c = 0; c < DIM; ++c)
      for (int e = 0; e < DIM; ++e) {
        double s = 0;
        for (int d = 0; d < DIM; ++d) s += tau[c][d] * g.invJ[e][d];
        tref[c][e][q] = s * g.jxw;
      }
  }
  for (int c = 0; c < DIM; ++c) q2_ref_grad_T<DIM>(tref[c], uout[c]);
  if (mode == M_STOKES) q1_values_T<DIM>(ptest, pout);
}

template <int DIM>
void cell_apply_mass(const Level &L, const CellData &cd, int cell, const double *pin, double *pout) {
  const int NQ = DIM == 3 ? 27 : 9;
  double pq[27];
  q1_values<DIM>(pin, pq);
  for (int q = 0; q < NQ; ++q) {
    Geo g; cell_geometry<DIM>(L, cell, q, g);
    const double eta = cd.eta[size_t(cell) * cd.nqv + (cd.nqv == 1 ? 0 : q)];
    pq[q] *= cd.lambda * cd.lambda / eta * g.jxw;   // :559-575
  }
  q1_values_T<DIM>(pq, pout);
}

// Threads of the CPU baseline leg of bench.py (ORACLE_THREADS, default 1 = the serial loop the
// tests use).  Threaded cell loops run colour by colour: cells of one colour share no node, so
// there are no races and the sums do not depend on the thread count.
static int oracle_threads() {
  static const int n = [] { const char *e = getenv("ORACLE_THREADS"); return e ? std::max(1, atoi(e)) : 1; }();
  return n;
}
static const std::vector<std::vector<int>> &cell_colours(const Level &L) {
  static std::map<const Level *, std::vector<std::vector<int>>> cache;
  auto it = cache.find(&L);
  if (it != cache.end() && !it->second.empty()) return it->second;
  std::vector<std::vector<int>> &col = cache[&L];
  std::vector<uint64_t> used(L.n_nodes_v, 0);
  for (int cell = 0; cell < L.n_cells; ++cell) {
    uint64_t forbidden = 0;
    for (int i = 0; i < L.nv; ++i) forbidden |= used[L.cnv[size_t(cell) * L.nv + i]];
    int k = 0;
    while (k < 63 && ((forbidden >> k) & 1)) ++k;
    for (int i = 0; i < L.nv; ++i) used[L.cnv[size_t(cell) * L.nv + i]] |= uint64_t(1) << k;
    if ((int)col.size() <= k) col.resize(k + 1);
    col[k].push_back(cell);
  }
  return col;
}

// geometry of face q-point q of boundary face f: outward normal and JxW (caller's tables, or
// +-e_d and the face area weights of a Cartesian cell)
template <int DIM>
inline void face_geometry(const Level &L, const CellData &cd, int f, int q, double n[3], double &jxw) {
  constexpr int NQF = DIM == 3 ? 9 : 3;
  const int cell = cd.face_cell[f], d = cd.face_no[f] / 2, side = cd.face_no[f] % 2;
  if (!cd.face_normal.empty()) {
    for (int c = 0; c < DIM; ++c) n[c] = cd.face_normal[(size_t(f) * NQF + q) * DIM + c];
    jxw = cd.face_jxw[size_t(f) * NQF + q];
    return;
  }
  for (int c = 0; c < DIM; ++c) n[c] = 0.0;
  n[d] = side ? 1.0 : -1.0;
  double w = 1.0;
  int qq = q;
  for (int k = 0; k < DIM; ++k) {
    if (k == d) continue;
    w *= T.wq[qq % 3] * L.cell_h[size_t(cell) * DIM + k];
    qq /= 3;
  }
  jxw = w;
}
// cell-local index of face node a (lexicographic over the tangential directions, lower first)
template <int DIM>
inline int face_node_local(int fno, int a) {
  const int d = fno / 2, side = fno % 2;
  int idx = 0, stride = 1;
  for (int k = 0; k < DIM; ++k) {
    int coord;
    if (k == d) coord = side ? 2 : 0;
    else { coord = a % 3; a /= 3; }
    idx += coord * stride;
    stride *= 3;
  }
  return idx;
}

// StokesOperator::local_apply_boundary_face (matrix_free_operators.cc:294-333): on every free-surface
// face evaluate u at the face q-points, submit -(g . u) n, integrate against the velocity values.
template <int DIM>
void face_terms(const Level &L, const CellData &cd, const double *su, double *du) {
  constexpr int NF = DIM == 3 ? 9 : 3, NQF = NF;
  for (int f = 0; f < cd.n_faces; ++f) {
    const uint32_t *nv = &L.cnv[size_t(cd.face_cell[f]) * L.nv];
    int node[NF];
    double u[NF][DIM], out[NF][DIM];
    for (int a = 0; a < NF; ++a) {
      node[a] = nv[face_node_local<DIM>(cd.face_no[f], a)];
      for (int c = 0; c < DIM; ++c) {
        u[a][c] = L.constrained_v(node[a], c) ? 0.0 : su[size_t(node[a]) * DIM + c];
        out[a][c] = 0.0;
      }
    }
    for (int q = 0; q < NQF; ++q) {
      double shape[NF];
      for (int a = 0; a < NF; ++a) shape[a] = DIM == 3 ? T.N2[q % 3][a % 3] * T.N2[q / 3][a / 3] : T.N2[q][a];
      double uq[DIM];
      for (int c = 0; c < DIM; ++c) {
        uq[c] = 0.0;
        for (int a = 0; a < NF; ++a) uq[c] += shape[a] * u[a][c];
      }
      double n[3], jxw, gu = 0.0;
      face_geometry<DIM>(L, cd, f, q, n, jxw);
      for (int c = 0; c < DIM; ++c) gu += cd.face_stab[(size_t(f) * NQF + q) * DIM + c] * uq[c];
      for (int a = 0; a < NF; ++a)
        for (int c = 0; c < DIM; ++c) out[a][c] += shape[a] * (-gu * n[c]) * jxw;
    }
    for (int a = 0; a < NF; ++a)
      for (int c = 0; c < DIM; ++c)
        if (!L.constrained_v(node[a], c)) du[size_t(node[a]) * DIM + c] += out[a][c];
  }
}
// the same term as a face matrix K[(a,c)][(b,c')] = -sum_q phi_a phi_b n_c g_c' JxW (the bilinear form
// of assemblers/stokes.cc free-surface stabilisation; second formulation for the oracle's self-check)
template <int DIM>
void dense_face_terms(const Level &L, const CellData &cd, const double *su, double *du) {
  constexpr int NF = DIM == 3 ? 9 : 3, NQF = NF, ND = NF * DIM;
  for (int f = 0; f < cd.n_faces; ++f) {
    const uint32_t *nv = &L.cnv[size_t(cd.face_cell[f]) * L.nv];
    std::vector<double> K(size_t(ND) * ND, 0.0);
    for (int q = 0; q < NQF; ++q) {
      double n[3], jxw;
      face_geometry<DIM>(L, cd, f, q, n, jxw);
      const double *g = &cd.face_stab[(size_t(f) * NQF + q) * DIM];
      for (int a = 0; a < NF; ++a)
        for (int b = 0; b < NF; ++b) {
          const double pa = DIM == 3 ? T.N2[q % 3][a % 3] * T.N2[q / 3][a / 3] : T.N2[q][a];
          const double pb = DIM == 3 ? T.N2[q % 3][b % 3] * T.N2[q / 3][b / 3] : T.N2[q][b];
          for (int c = 0; c < DIM; ++c)
            for (int c2 = 0; c2 < DIM; ++c2) K[size_t(a * DIM + c) * ND + b * DIM + c2] -= pa * pb * n[c] * g[c2] * jxw;
        }
    }
    for (int a = 0; a < NF; ++a) {
      const int na = nv[face_node_local<DIM>(cd.face_no[f], a)];
      for (int c = 0; c < DIM; ++c) {
        if (L.constrained_v(na, c)) continue;
        double sum = 0.0;
        for (int b = 0; b < NF; ++b) {
          const int nb = nv[face_node_local<DIM>(cd.face_no[f], b)];
          for (int c2 = 0; c2 < DIM; ++c2)
            if (!L.constrained_v(nb, c2)) sum += K[size_t(a * DIM + c) * ND + b * DIM + c2] * su[size_t(nb) * DIM + c2];
        }
        du[size_t(na) * DIM + c] += sum;
      }
    }
  }
}

// dst = Op src with MatrixFreeOperators::Base::vmult semantics.
// su/sp, du/dp: velocity and pressure parts (may be null when unused).
template <int DIM>
void op_vmult(const Level &L, const CellData &cd, Mode mode,
              const double *su, const double *sp, double *du, double *dp) {
  const int NV = L.nv, NP = L.np;
  std::fill(du, du + L.ndv(), 0.0);
  if (dp) std::fill(dp, dp + L.ndp(), 0.0);
  const int nthreads = oracle_threads();
  const bool threaded = nthreads > 1 && L.n_cells >= 512;
  std::vector<std::vector<int>> serial_one;
  if (!threaded) { serial_one.resize(1); }
  const std::vector<std::vector<int>> &colours = threaded ? cell_colours(L) : serial_one;
  for (size_t colour = 0; colour < colours.size(); ++colour) {
  const int n_in_colour = threaded ? (int)colours[colour].size() : L.n_cells;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (threaded)
  for (int k = 0; k < n_in_colour; ++k) {
    const int cell = threaded ? colours[colour][k] : k;
    double uin[3][27], pin[8], uout[3][27], pout[8];
    const uint32_t *nv = &L.cnv[size_t(cell) * NV];
    const uint32_t *np = &L.cnp[size_t(cell) * NP];
    if (mode != M_BT)
      for (int i = 0; i < NV; ++i)
        for (int c = 0; c < DIM; ++c)
          uin[c][i] = L.constrained_v(nv[i], c) ? 0.0 : su[size_t(nv[i]) * DIM + c];
    if (mode != M_A)
      for (int i = 0; i < NP; ++i) pin[i] = L.constrained_p(np[i]) ? 0.0 : sp[np[i]];
    cell_apply<DIM>(L, cd, cell, mode, uin, pin, uout, pout);
    for (int i = 0; i < NV; ++i)
      for (int c = 0; c < DIM; ++c)
        if (!L.constrained_v(nv[i], c)) du[size_t(nv[i]) * DIM + c] += uout[c][i];
    if (mode == M_STOKES)
      for (int i = 0; i < NP; ++i)
        if (!L.constrained_p(np[i])) dp[np[i]] += pout[i];
  }
  }
  if (mode == M_STOKES && cd.n_faces) face_terms<DIM>(L, cd, su, du);
  // constrained rows are identity
  for (int n = 0; n < L.n_nodes_v; ++n)
    if (L.cmv[n])
      for (int c = 0; c < DIM; ++c)
        if (L.constrained_v(n, c)) du[size_t(n) * DIM + c] += su ? su[size_t(n) * DIM + c] : 0.0;
  if (dp && sp)
    for (int n = 0; n < L.n_nodes_p; ++n)
      if (L.constrained_p(n)) dp[n] += sp[n];
}

template <int DIM>
void mass_vmult(const Level &L, const CellData &cd, const double *sp, double *dp) {
  const int NP = L.np;
  std::fill(dp, dp + L.ndp(), 0.0);
  for (int cell = 0; cell < L.n_cells; ++cell) {
    double pin[8], pout[8];
    const uint32_t *np = &L.cnp[size_t(cell) * NP];
    for (int i = 0; i < NP; ++i) pin[i] = L.constrained_p(np[i]) ? 0.0 : sp[np[i]];
    cell_apply_mass<DIM>(L, cd, cell, pin, pout);
    for (int i = 0; i < NP; ++i) if (!L.constrained_p(np[i])) dp[np[i]] += pout[i];
  }
  for (int n = 0; n < L.n_nodes_p; ++n) if (L.constrained_p(n)) dp[n] += sp[n];
}

// ---- independent dense restatement of the bilinear form (assemblers/stokes.cc:463-479)
template <int DIM>
void dense_cell_matrix(const Level &L, const CellData &cd, int cell, std::vector<double> &K) {
  const int NV = L.nv, NP = L.np, NQ = L.nq, ND = DIM * NV + NP;
  K.assign(size_t(ND) * ND, 0.0);
  const bool comp = cd.is_compressible || cd.dilation;
  for (int q = 0; q < NQ; ++q) {
    Geo g; cell_geometry<DIM>(L, cell, q, g);
    const double eta = cd.eta[size_t(cell) * cd.nqv + (cd.nqv == 1 ? 0 : q)];
    int qi[3] = {q % 3, (q / 3) % 3, q / 9};
    // shape gradients in physical space
    std::vector<double> gphi(size_t(NV) * DIM), psi(NP);
    for (int i = 0; i < NV; ++i) {
      int ii[3] = {i % 3, (i / 3) % 3, i / 9};
      double gr[3];
      for (int e = 0; e < DIM; ++e) {
        double v = 1;
        for (int d = 0; d < DIM; ++d) v *= (d == e ? T.D2[qi[d]][ii[d]] : T.N2[qi[d]][ii[d]]);
        gr[e] = v;
      }
      for (int d = 0; d < DIM; ++d) {
        double s = 0;
        for (int e = 0; e < DIM; ++e) s += gr[e] * g.invJ[e][d];
        gphi[i * DIM + d] = s;
      }
    }
    for (int i = 0; i < NP; ++i) {
      int ii[3] = {i % 2, (i / 2) % 2, i / 4};
      double v = 1;
      for (int d = 0; d < DIM; ++d) v *= T.N1[qi[d]][ii[d]];
      psi[i] = v;
    }
    // symmetric gradient of vector shape function (i,c): eps_ab = 0.5 (delta_ca g_b + delta_cb g_a)
    auto eps = [&](int i, int c, double e[3][3]) {
      for (int a = 0; a < DIM; ++a) for (int b = 0; b < DIM; ++b)
        e[a][b] = 0.5 * ((a == c ? gphi[i * DIM + b] : 0.0) + (b == c ? gphi[i * DIM + a] : 0.0));
    };
    for (int i = 0; i < NV; ++i) for (int c = 0; c < DIM; ++c) {
      double ei[3][3]; eps(i, c, ei);
      const double divi = gphi[i * DIM + c];
      const int row = c * NV + i;
      for (int j = 0; j < NV; ++j) for (int c2 = 0; c2 < DIM; ++c2) {
        double ej[3][3]; eps(j, c2, ej);
        double s = 0;
        for (int a = 0; a < DIM; ++a) for (int b = 0; b < DIM; ++b) s += ei[a][b] * ej[a][b];
        double v = 2.0 * eta * s;
        if (comp) v -= 2.0 * eta / 3.0 * divi * gphi[j * DIM + c2];
        K[size_t(row) * ND + c2 * NV + j] += v * g.jxw;
      }
      for (int j = 0; j < NP; ++j) {
        K[size_t(row) * ND + DIM * NV + j] += -cd.lambda * divi * psi[j] * g.jxw;
        K[size_t(DIM * NV + j) * ND + row] += -cd.lambda * psi[j] * divi * g.jxw;
      }
    }
    if (cd.dilation)   // - lambda^2 dilation_lhs p q
      for (int i = 0; i < NP; ++i) for (int j = 0; j < NP; ++j)
        K[size_t(DIM * NV + i) * ND + DIM * NV + j] += -cd.lambda * cd.lambda * cd.dil_lhs[size_t(cell) * NQ + q] * psi[i] * psi[j] * g.jxw;
  }
}
// Newton terms of the dense cell matrix: column j of the extra block is
//   sum_q [ (2 E_q S1_j | E_q S1_j + D_q S3_j) + 2 E_q S2_j ] : eps_q(phi_i) JxW_q
// with the cell-wide sums S1_j = sum_q' D_q' : eps_q'(phi_j), S3_j = sum_q' E_q' : eps_q'(phi_j)
// (velocity columns) and S2_j = sum_q' dEdp_q' psi_j(q') (pressure columns)
template <int DIM>
void dense_newton_terms(const Level &L, const CellData &cd, int cell, std::vector<double> &K) {
  const int NV = L.nv, NP = L.np, NQ = L.nq, ND = DIM * NV + NP;
  constexpr int NS = DIM * (DIM + 1) / 2;
  std::vector<double> gphi(size_t(NQ) * NV * DIM), psi(size_t(NQ) * NP), jxw(NQ);
  for (int q = 0; q < NQ; ++q) {
    Geo g; cell_geometry<DIM>(L, cell, q, g);
    jxw[q] = g.jxw;
    int qi[3] = {q % 3, (q / 3) % 3, q / 9};
    for (int i = 0; i < NV; ++i) {
      int ii[3] = {i % 3, (i / 3) % 3, i / 9};
      double gr[3];
      for (int e = 0; e < DIM; ++e) {
        double v = 1;
        for (int d = 0; d < DIM; ++d) v *= (d == e ? T.D2[qi[d]][ii[d]] : T.N2[qi[d]][ii[d]]);
        gr[e] = v;
      }
      for (int d = 0; d < DIM; ++d) {
        double s = 0;
        for (int e = 0; e < DIM; ++e) s += gr[e] * g.invJ[e][d];
        gphi[(size_t(q) * NV + i) * DIM + d] = s;
      }
    }
    for (int i = 0; i < NP; ++i) {
      int ii[3] = {i % 2, (i / 2) % 2, i / 4};
      double v = 1;
      for (int d = 0; d < DIM; ++d) v *= T.N1[qi[d]][ii[d]];
      psi[size_t(q) * NP + i] = v;
    }
  }
  // table tensor : eps_q(phi_(i,c))
  auto contract = [&](const double *t, int q, int i, int c) {
    double s = 0;
    for (int b = 0; b < DIM; ++b) s += t[symc(DIM, c, b)] * gphi[(size_t(q) * NV + i) * DIM + b];
    return s;   // T symmetric: T : sym(grad) = sum_b T_cb d_b phi
  };
  std::vector<double> colS1(ND, 0.0), colS3(ND, 0.0), colS2(ND, 0.0);
  for (int q = 0; q < NQ; ++q) {
    const double *de = &cd.dEde[(size_t(cell) * NQ + q) * NS], *st = &cd.strain[(size_t(cell) * NQ + q) * NS];
    for (int j = 0; j < NV; ++j) for (int c2 = 0; c2 < DIM; ++c2) {
      colS1[c2 * NV + j] += contract(de, q, j, c2);
      colS3[c2 * NV + j] += contract(st, q, j, c2);
    }
    for (int j = 0; j < NP; ++j) colS2[DIM * NV + j] += cd.dEdp[size_t(cell) * NQ + q] * psi[size_t(q) * NP + j];
  }
  if (cd.dilation && !cd.dil_de.empty())   // pressure rows: lambda (D_q : eps_q(phi_j) + dDdp_q lambda psi_j) psi_i JxW
    for (int q = 0; q < NQ; ++q) {
      const double *dd = &cd.dil_de[(size_t(cell) * NQ + q) * NS];
      for (int i = 0; i < NP; ++i) {
        const double pi = psi[size_t(q) * NP + i] * jxw[q] * cd.lambda;
        for (int j = 0; j < NV; ++j) for (int c2 = 0; c2 < DIM; ++c2)
          K[size_t(DIM * NV + i) * ND + c2 * NV + j] += contract(dd, q, j, c2) * pi;
        for (int j = 0; j < NP; ++j)
          K[size_t(DIM * NV + i) * ND + DIM * NV + j] += cd.dil_dp[size_t(cell) * NQ + q] * cd.lambda * psi[size_t(q) * NP + j] * pi;
      }
    }
  for (int q = 0; q < NQ; ++q) {
    const double *de = &cd.dEde[(size_t(cell) * NQ + q) * NS], *st = &cd.strain[(size_t(cell) * NQ + q) * NS];
    for (int i = 0; i < NV; ++i) for (int c = 0; c < DIM; ++c) {
      const double Ei = contract(st, q, i, c), Di = contract(de, q, i, c);
      const int row = c * NV + i;
      for (int col = 0; col < ND; ++col) {
        const double v = (cd.symmetrize ? Ei * colS1[col] + Di * colS3[col] : 2.0 * Ei * colS1[col]) + 2.0 * Ei * colS2[col];
        K[size_t(row) * ND + col] += v * jxw[q];
      }
    }
  }
}

template <int DIM>
void dense_mass_matrix(const Level &L, const CellData &cd, int cell, std::vector<double> &K) {
  const int NP = L.np, NQ = L.nq;
  K.assign(size_t(NP) * NP, 0.0);
  for (int q = 0; q < NQ; ++q) {
    Geo g; cell_geometry<DIM>(L, cell, q, g);
    const double eta = cd.eta[size_t(cell) * cd.nqv + (cd.nqv == 1 ? 0 : q)];
    int qi[3] = {q % 3, (q / 3) % 3, q / 9};
    double psi[8];
    for (int i = 0; i < NP; ++i) {
      int ii[3] = {i % 2, (i / 2) % 2, i / 4};
      double v = 1;
      for (int d = 0; d < DIM; ++d) v *= T.N1[qi[d]][ii[d]];
      psi[i] = v;
    }
    for (int i = 0; i < NP; ++i) for (int j = 0; j < NP; ++j)
      K[i * NP + j] += cd.lambda * cd.lambda / eta * psi[i] * psi[j] * g.jxw;
  }
}
template <int DIM>
void dense_vmult(const Level &L, const CellData &cd, int which, const double *src, double *dst) {
  const int NV = L.nv, NP = L.np, ND = DIM * NV + NP;
  const int nu = L.ndv(), npd = L.ndp();
  const double *su = src, *sp = (which == AGMG_OP_MASS) ? src : src + nu;
  double *du = dst, *dp = (which == AGMG_OP_MASS) ? dst : dst + nu;
  if (which != AGMG_OP_MASS) std::fill(du, du + nu, 0.0);
  if (which != AGMG_OP_A) std::fill(dp, dp + npd, 0.0);
  std::vector<double> K;
  for (int cell = 0; cell < L.n_cells; ++cell) {
    const uint32_t *nv = &L.cnv[size_t(cell) * NV];
    const uint32_t *np = &L.cnp[size_t(cell) * NP];
    if (which == AGMG_OP_MASS) {
      dense_mass_matrix<DIM>(L, cd, cell, K);
      for (int i = 0; i < NP; ++i) {
        double s = 0;
        for (int j = 0; j < NP; ++j) s += K[i * NP + j] * sp[np[j]];
        dp[np[i]] += s;
      }
      continue;
    }
    dense_cell_matrix<DIM>(L, cd, cell, K);
    if (cd.newton && which == AGMG_OP_STOKES) dense_newton_terms<DIM>(L, cd, cell, K);
    std::vector<double> x(ND, 0.0), y(ND, 0.0);
    if (which != AGMG_OP_BT)
      for (int i = 0; i < NV; ++i) for (int c = 0; c < DIM; ++c)
        x[c * NV + i] = L.constrained_v(nv[i], c) ? 0.0 : su[size_t(nv[i]) * DIM + c];
    if (which != AGMG_OP_A)
      for (int i = 0; i < NP; ++i) x[DIM * NV + i] = sp[np[i]];
    for (int r = 0; r < ND; ++r) {
      double s = 0;
      for (int k = 0; k < ND; ++k) s += K[size_t(r) * ND + k] * x[k];
      y[r] = s;
    }
    for (int i = 0; i < NV; ++i) for (int c = 0; c < DIM; ++c)
      if (!L.constrained_v(nv[i], c)) du[size_t(nv[i]) * DIM + c] += y[c * NV + i];
    if (which == AGMG_OP_STOKES)
      for (int i = 0; i < NP; ++i) dp[np[i]] += y[DIM * NV + i];
  }
  if (which == AGMG_OP_STOKES && cd.n_faces) dense_face_terms<DIM>(L, cd, su, du);
  if (which != AGMG_OP_MASS)
    for (int n = 0; n < L.n_nodes_v; ++n)
      for (int c = 0; c < DIM; ++c)
        if (L.constrained_v(n, c)) du[size_t(n) * DIM + c] += su[size_t(n) * DIM + c];
}

// ------------------------------------------------------------------ diagonals
// MatrixFreeTools::compute_diagonal: (A e_i)_i cell-wise, constrained -> 1, invert.
template <int DIM>
void compute_diagonals(Level &L) {
  const int NV = L.nv, NP = L.np;
  Vec dA(L.ndv(), 0.0), dM(L.ndp(), 0.0);
  for (int cell = 0; cell < L.n_cells; ++cell) {
    const uint32_t *nv = &L.cnv[size_t(cell) * NV];
    const uint32_t *np = &L.cnp[size_t(cell) * NP];
    for (int c = 0; c < DIM; ++c)
      for (int i = 0; i < NV; ++i) {
        double uin[3][27] = {{0}}, uout[3][27];
        uin[c][i] = 1.0;
        cell_apply<DIM>(L, L.cd, cell, M_A, uin, nullptr, uout, nullptr);
        if (!L.constrained_v(nv[i], c)) dA[size_t(nv[i]) * DIM + c] += uout[c][i];
      }
    for (int i = 0; i < NP; ++i) {
      double pin[8] = {0}, pout[8];
      pin[i] = 1.0;
      cell_apply_mass<DIM>(L, L.cd, cell, pin, pout);
      if (!L.constrained_p(np[i])) dM[np[i]] += pout[i];
    }
  }
  for (int n = 0; n < L.n_nodes_v; ++n)
    for (int c = 0; c < DIM; ++c)
      if (L.constrained_v(n, c)) dA[size_t(n) * DIM + c] = 1.0;
  for (int n = 0; n < L.n_nodes_p; ++n) if (L.constrained_p(n)) dM[n] = 1.0;
  L.diag_M = dM;
  L.inv_diag_A.resize(dA.size());
  L.inv_diag_M.resize(dM.size());
  for (size_t i = 0; i < dA.size(); ++i) L.inv_diag_A[i] = 1.0 / dA[i];
  for (size_t i = 0; i < dM.size(); ++i) L.inv_diag_M[i] = 1.0 / dM[i];
}

// ------------------------------------------------------------------ vector ops
double dot(const Vec &a, const Vec &b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}
double norm(const Vec &a) { return std::sqrt(dot(a, a)); }
void axpy(Vec &y, double a, const Vec &x) { for (size_t i = 0; i < y.size(); ++i) y[i] += a * x[i]; }

// eigenvalues of a symmetric tridiagonal matrix (implicit QL), ascending
std::vector<double> tridiag_eigenvalues(std::vector<double> d, std::vector<double> e) {
  const int n = int(d.size());
  e.resize(n, 0.0);
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 1e-300 + 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (++iter > 200) break;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[m] = 0.0;
      }
    } while (m != l);
  }
  std::sort(d.begin(), d.end());
  return d;
}

using LinOp = std::function<void(Vec &, const Vec &)>;

// deal.II SolverCG (preconditioned), stopping on ||r||_2 <= tol or it == max_it.
// Returns iterations; optionally collects the Lanczos tridiagonal for eigenvalues.
struct CGResult { int its; double res; bool converged; std::vector<double> eigs; };
CGResult solver_cg(const LinOp &A, const LinOp &P, Vec &x, const Vec &b, int max_it, double tol,
                   bool iteration_number_control, bool want_eigs) {
  const size_t n = b.size();
  Vec g(n), d(n), h(n), Ad(n);
  A(g, x);
  for (size_t i = 0; i < n; ++i) g[i] = b[i] - g[i];
  double res = norm(g);
  CGResult R{0, res, false, {}};
  std::vector<double> diag, off;
  if (res <= tol || (iteration_number_control && max_it == 0)) { R.converged = true; return R; }
  double gh = 0, alpha = 0, beta = 0;
  int it = 0;
  while (true) {
    ++it;
    P(h, g);
    const double gh_old = gh;
    gh = dot(g, h);
    if (it > 1) {
      beta = gh / gh_old;
      for (size_t i = 0; i < n; ++i) d[i] = h[i] + beta * d[i];
    } else d = h;
    A(Ad, d);
    const double alpha_old = alpha;
    alpha = gh / dot(d, Ad);
    axpy(x, alpha, d);
    axpy(g, -alpha, Ad);
    res = norm(g);
    if (want_eigs) {
      if (it == 1) diag.push_back(1.0 / alpha);
      else { diag.push_back(1.0 / alpha + beta / alpha_old); off.push_back(std::sqrt(beta) / alpha_old); }
    }
    R.its = it; R.res = res;
    if (res <= tol) { R.converged = true; break; }
    if (it >= max_it) { R.converged = iteration_number_control; break; }
    if (!std::isfinite(res)) break;
  }
  if (want_eigs && !diag.empty()) R.eigs = tridiag_eigenvalues(diag, off);
  return R;
}

enum { ST_SUCCESS = 0, ST_ITERATE = 1, ST_FAILURE = 2 };
// deal.II SolverBicgstab with AdditionalData(exact_residual = false) as BlockSchurGMGPreconditioner
// uses it for a non-symmetric A block (stokes_matrix_free.h:321-333): right-preconditioned
// BiCGStab, the residual norm taken from the recursively updated residual, a check after the
// first half step; a breakdown (|rho| tiny) restarts from the current iterate.
CGResult solver_bicgstab(const LinOp &A, const LinOp &P, Vec &x, const Vec &b, int max_it, double tol) {
  const size_t n = b.size();
  const double breakdown = std::numeric_limits<double>::min();
  Vec r(n), rbar(n), p(n), y(n), z(n), t(n), v(n);
  CGResult R{0, 0.0, false, {}};
  int step = 0;
  auto status = [&](double res) {   // SolverControl::check
    R.its = step; R.res = res;
    if (res <= tol) return ST_SUCCESS;
    if (step >= max_it || !std::isfinite(res)) return ST_FAILURE;
    return ST_ITERATE;
  };
  for (int restart = 0; restart < 1000; ++restart) {
    A(r, x);
    for (size_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
    double res = norm(r);
    int state = status(res);
    if (state == ST_SUCCESS) { R.converged = true; return R; }
    if (state == ST_FAILURE) return R;
    double alpha = 1, omega = 1, rho = 1;
    rbar = r;
    bool startup = true, broke = false;
    do {
      ++step;
      double rhobar = dot(r, rbar);
      if (std::fabs(rhobar) < breakdown) { broke = true; break; }
      const double beta = rhobar * alpha / (rho * omega);
      rho = rhobar;
      if (startup) { p = r; startup = false; }
      else for (size_t i = 0; i < n; ++i) p[i] = beta * p[i] + r[i] - beta * omega * v[i];
      P(y, p);
      A(v, y);
      rhobar = dot(rbar, v);
      if (std::fabs(rhobar) < breakdown) { broke = true; break; }
      alpha = rho / rhobar;
      axpy(r, -alpha, v);
      res = norm(r);
      if (status(res) == ST_SUCCESS) { axpy(x, alpha, y); R.converged = true; return R; }
      P(z, r);
      A(t, z);
      rhobar = dot(t, r);
      const double tt = dot(t, t);
      if (tt < breakdown) { broke = true; break; }
      omega = rhobar / tt;
      for (size_t i = 0; i < n; ++i) x[i] += alpha * y[i] + omega * z[i];
      axpy(r, -omega, t);
      res = norm(r);
      state = status(res);
    } while (state == ST_ITERATE);
    if (!broke) { R.converged = state == ST_SUCCESS; return R; }
  }
  return R;
}

// ------------------------------------------------------------------ Chebyshev
struct Chebyshev {
  LinOp A;
  const Vec *dinv = nullptr;
  int degree = 4;
  double range = 15.;
  int eig_its = 10;
  double lmin = 0, lmax = 0, theta = 1, delta = 0;
  int n = 0;
  void estimate() {
    // set_initial_guess: (global index mod 11) minus the mean
    Vec b(n), x(n, 0.0);
    double mean = 0;
    for (int i = 0; i < n; ++i) { b[i] = double(i % 11); mean += b[i]; }
    mean /= n;
    for (int i = 0; i < n; ++i) b[i] -= mean;
    LinOp P = [&](Vec &d, const Vec &s) { for (int i = 0; i < n; ++i) d[i] = (*dinv)[i] * s[i]; };
    CGResult r = solver_cg(A, P, x, b, eig_its, 1e-10, true, true);
    double emin = 1, emax = 1;
    if (!r.eigs.empty()) { emin = r.eigs.front(); emax = 1.2 * r.eigs.back(); }
    lmax = emax;
    const double alpha = range > 1. ? emax / range : std::min(0.9 * emax, emin);
    lmin = alpha;
    delta = (emax - alpha) * 0.5;
    theta = (emax + alpha) * 0.5;
  }
  // PreconditionChebyshev::vmult (zero start) -> degree-1 matvecs
  void vmult(Vec &x, const Vec &b) const {
    Vec xo(n, 0.0), t(n);
    for (int i = 0; i < n; ++i) x[i] = (1. / theta) * (*dinv)[i] * b[i];
    if (degree < 2 || std::fabs(delta) < 1e-40) return;
    double rhok = delta / theta, sigma = theta / delta;
    for (int k = 0; k < degree - 1; ++k) {
      A(t, x);
      const double rhokp = 1. / (2. * sigma - rhok);
      const double f1 = rhokp * rhok, f2 = 2. * rhokp / delta;
      rhok = rhokp;
      update(k + 1, f1, f2, x, xo, t, b);
    }
  }
  // PreconditionChebyshev::step (non-zero start) -> degree matvecs
  void step(Vec &x, const Vec &b) const {
    Vec xo(n, 0.0), t(n);
    A(t, x);
    update(1, 0., 1. / theta, x, xo, t, b);
    if (degree < 2 || std::fabs(delta) < 1e-40) return;
    double rhok = delta / theta, sigma = theta / delta;
    for (int k = 0; k < degree - 1; ++k) {
      A(t, x);
      const double rhokp = 1. / (2. * sigma - rhok);
      const double f1 = rhokp * rhok, f2 = 2. * rhokp / delta;
      rhok = rhokp;
      update(k + 2, f1, f2, x, xo, t, b);
    }
  }
  void update(int idx, double f1, double f2, Vec &x, Vec &xo, const Vec &t, const Vec &b) const {
    if (idx == 1)
      for (int i = 0; i < n; ++i) xo[i] = (1. + f1) * x[i] + f2 * (*dinv)[i] * (b[i] - t[i]);
    else
      for (int i = 0; i < n; ++i) xo[i] = (1. + f1) * x[i] - f1 * xo[i] + f2 * (*dinv)[i] * (b[i] - t[i]);
    x.swap(xo);
  }
};

// ------------------------------------------------------------------ context
struct Ctx {
  agmg_options opt;
  std::vector<Level> levels;
  CellData active_cd;          // active mesh = finest level tables + its own coefficients
  std::vector<Chebyshev> chebA, chebS;
  bool smoothers_ready = false;
  int nA_cheap = 0, nA_exp = 0, nS_cheap = 0, nS_exp = 0;
  Level active;                // the active mesh when it is not the finest level (adaptive refinement)
  bool has_active = false;
  Level &fin() { return levels.back(); }
  Level &act() { return has_active ? active : levels.back(); }
  const Level &act() const { return has_active ? active : levels.back(); }
};

void active_vmult(Ctx &C, int which, double *dst, const double *src);   // operators of the active mesh, hanging nodes resolved

template <int DIM>
void level_A(const Level &L, Vec &d, const Vec &s) { op_vmult<DIM>(L, L.cd, M_A, s.data(), nullptr, d.data(), nullptr); }

// ------------------------------------------------------------------ transfer
// MGTransferMF for Q2^dim / Q1 on uniformly refined parents:
// per parent cell embed the coarse cell values into the (2k+1)^dim nodes of
// its children; shared fine nodes weighted by 1/multiplicity; coarse Dirichlet
// values zeroed before prolongation and after restriction.
template <int DIM>
void transfer(const Ctx &C, int which, int fine_level, bool prolong, Vec &dst, const Vec &src) {
  const Level &F = C.levels[fine_level], &Cs = C.levels[fine_level - 1];
  const bool vel = which == AGMG_OP_A;
  const int k = vel ? 2 : 1, n1 = k + 1, nf1 = 2 * k + 1;
  const int ncomp = vel ? DIM : 1;
  const int nloc = ipow(n1, DIM), npatch = ipow(nf1, DIM);
  const double *P1d = vel ? &T.P2[0][0] : &T.P1[0][0];
  const std::vector<uint32_t> &cnF = vel ? F.cnv : F.cnp, &cnC = vel ? Cs.cnv : Cs.cnp;
  const int nnF = vel ? F.n_nodes_v : F.n_nodes_p;
  // multiplicity of fine nodes over parent patches: occurrences (parent, patch position); on a
  // periodic mesh one cell wide the two faces of a patch are the same fine nodes
  std::vector<double> weight(nnF, 0.0);
  const int nchild = 1 << DIM;
  std::vector<std::vector<int>> children(Cs.n_cells, std::vector<int>(nchild, -1));
  for (int f = 0; f < F.n_cells; ++f) children[F.parent[f]][F.child[f]] = f;
  auto patch_nodes = [&](int parent, std::vector<uint32_t> &pn) {
    pn.assign(npatch, 0);
    for (int c = 0; c < nchild; ++c) {
      const int f = children[parent][c];
      for (int i = 0; i < nloc; ++i) {
        int rem = i, pidx = 0, stride = 1;
        for (int d = 0; d < DIM; ++d) {
          const int id = rem % n1; rem /= n1;
          pidx += (id + k * ((c >> d) & 1)) * stride;
          stride *= nf1;
        }
        pn[pidx] = cnF[size_t(f) * nloc + i];
      }
    }
  };
  std::vector<uint32_t> pn;
  // (local smoothing on adaptive meshes: a coarse cell that is active on its level has no children and takes no part)
  for (int p = 0; p < Cs.n_cells; ++p) {
    if (children[p][0] < 0) continue;
    patch_nodes(p, pn);
    for (int i = 0; i < npatch; ++i) weight[pn[i]] += 1.0;
  }
  for (auto &w : weight) w = w > 0.0 ? 1.0 / w : 0.0;
  auto coarse_constrained = [&](int node, int c) { return vel ? Cs.boundary_v(node, c) : Cs.boundary_p(node); };
  int n[3] = {n1, n1, DIM == 3 ? n1 : 1};
  std::vector<double> a(125), b(125), loc(27);
  for (int p = 0; p < Cs.n_cells; ++p) {
    if (children[p][0] < 0) continue;
    patch_nodes(p, pn);
    for (int c = 0; c < ncomp; ++c) {
      if (prolong) {
        for (int i = 0; i < nloc; ++i) {
          const uint32_t node = cnC[size_t(p) * nloc + i];
          loc[i] = coarse_constrained(node, c) ? 0.0 : src[size_t(node) * ncomp + c];
        }
        int nn[3] = {n[0], n[1], n[2]};
        const double *s = loc.data();
        double *bufs[2] = {a.data(), b.data()};
        int w = 0;
        for (int d = 0; d < DIM; ++d) { apply1d(P1d, nf1, n1, d, nn, s, bufs[w]); nn[d] = nf1; s = bufs[w]; w ^= 1; }
        for (int i = 0; i < npatch; ++i) dst[size_t(pn[i]) * ncomp + c] += weight[pn[i]] * s[i];
      } else {
        for (int i = 0; i < npatch; ++i) a[i] = weight[pn[i]] * src[size_t(pn[i]) * ncomp + c];
        int nn[3] = {nf1, nf1, DIM == 3 ? nf1 : 1};
        const double *s = a.data();
        double *bufs[2] = {b.data(), a.data()};
        int w = 0;
        for (int d = 0; d < DIM; ++d) { apply1d_T(P1d, nf1, n1, d, nn, s, bufs[w]); nn[d] = n1; s = bufs[w]; w ^= 1; }
        for (int i = 0; i < nloc; ++i) {
          const uint32_t node = cnC[size_t(p) * nloc + i];
          if (!coarse_constrained(node, c)) dst[size_t(node) * ncomp + c] += s[i];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ multigrid
template <int DIM>
struct MG {
  Ctx &C;
  int which;  // AGMG_OP_A or AGMG_OP_MASS
  std::vector<Vec> defect, solution, t;
  MG(Ctx &c, int w) : C(c), which(w) {}
  int size(int l) const { return which == AGMG_OP_A ? C.levels[l].ndv() : C.levels[l].ndp(); }
  void matvec(int l, Vec &d, const Vec &s) {
    const Level &L = C.levels[l];
    if (which == AGMG_OP_A) op_vmult<DIM>(L, L.cd, M_A, s.data(), nullptr, d.data(), nullptr);
    else mass_vmult<DIM>(L, L.cd, s.data(), d.data());
  }
  const Chebyshev &cheb(int l) const { return which == AGMG_OP_A ? C.chebA[l] : C.chebS[l]; }
  int steps() const { return which == AGMG_OP_A ? 1 : C.opt.schur_smoother_steps; }
  void smoother_apply(int l, Vec &u, const Vec &rhs) {   // mg::SmootherRelaxation::apply
    cheb(l).vmult(u, rhs);
    for (int i = 1; i < steps(); ++i) cheb(l).step(u, rhs);
  }
  void smoother_smooth(int l, Vec &u, const Vec &rhs) {  // ::smooth
    for (int i = 0; i < steps(); ++i) cheb(l).step(u, rhs);
  }
  // MGInterfaceOperator on a level with a refinement edge (MatrixFreeOperators::Base::vmult_interface_down / _up):
  // the operator with the boundary rows alone constrained, applied to the source without (down) or with only (up) its
  // refinement-edge entries, of which the edge rows (down) or all other rows (up) are kept
  void interface(int l, bool down, Vec &d, const Vec &s) {
    Level &L = C.levels[l];
    const bool vel = which == AGMG_OP_A;
    const int nc = vel ? DIM : 1;
    const std::vector<uint8_t> &edge = vel ? L.edge_v : L.edge_p;
    Vec x(s);
    for (size_t i = 0; i < x.size(); ++i) if ((edge[i / nc] != 0) == down) x[i] = 0.0;
    std::swap(L.cmv, L.bnd_v); std::swap(L.cmp, L.bnd_p);
    matvec(l, d, x);
    std::swap(L.cmv, L.bnd_v); std::swap(L.cmp, L.bnd_p);
    for (size_t i = 0; i < d.size(); ++i) if ((edge[i / nc] != 0) != down) d[i] = 0.0;
    // (rows of boundary nodes: matvec returned the identity there; the reference's apply_add writes nothing)
    for (size_t i = 0; i < d.size(); ++i) if (vel ? L.boundary_v(int(i / nc), int(i % nc)) : L.boundary_p(int(i))) d[i] = 0.0;
  }
  // Multigrid::level_v_step with set_edge_matrices(interface, interface) (local_smoothing.cc:1045, 1053)
  void level_v_step(int l) {
    if (l == 0) { smoother_apply(0, solution[0], defect[0]); return; }
    const bool edge = C.levels[l].has_edge;
    smoother_apply(l, solution[l], defect[l]);
    matvec(l, t[l], solution[l]);
    if (edge) {                                  // edge_out->vmult_add
      Vec e(t[l].size());
      interface(l, true, e, solution[l]);
      axpy(t[l], 1.0, e);
    }
    for (size_t i = 0; i < t[l].size(); ++i) t[l][i] = defect[l][i] - t[l][i];
    transfer<DIM>(C, which, l, false, defect[l - 1], t[l]);
    level_v_step(l - 1);
    transfer<DIM>(C, which, l, true, solution[l], solution[l - 1]);
    if (edge) {                                  // edge_in->Tvmult: defect -= (interface up) solution
      Vec e(t[l].size());
      interface(l, false, e, solution[l]);
      axpy(defect[l], -1.0, e);
    }
    smoother_smooth(l, solution[l], defect[l]);
  }
  void vcycle(Vec &dst, const Vec &src) {   // PreconditionMG::vmult
    const int nl = int(C.levels.size());
    defect.resize(nl); solution.resize(nl); t.resize(nl);
    for (int l = 0; l < nl; ++l) { defect[l].assign(size(l), 0.0); solution[l].assign(size(l), 0.0); t[l].assign(size(l), 0.0); }
    if (!C.has_active) {
      defect[nl - 1] = src;
      level_v_step(nl - 1);
      dst = solution[nl - 1];
      return;
    }
    // copy_to_mg / copy_from_mg through the copy indices of every level (MGLevelGlobalTransfer)
    const int sp = which == AGMG_OP_A ? 0 : 1, nc = which == AGMG_OP_A ? DIM : 1;
    for (int l = 0; l < nl; ++l) {
      const Level &L = C.levels[l];
      for (size_t i = 0; i < L.copy_act[sp].size(); ++i)
        for (int c = 0; c < nc; ++c) defect[l][size_t(L.copy_lev[sp][i]) * nc + c] = src[size_t(L.copy_act[sp][i]) * nc + c];
    }
    level_v_step(nl - 1);
    dst.assign(src.size(), 0.0);
    for (int l = 0; l < nl; ++l) {
      const Level &L = C.levels[l];
      for (size_t i = 0; i < L.copy_act[sp].size(); ++i)
        for (int c = 0; c < nc; ++c) dst[size_t(L.copy_act[sp][i]) * nc + c] = solution[l][size_t(L.copy_lev[sp][i]) * nc + c];
    }
  }
};

template <int DIM>
void setup_smoothers(Ctx &C) {
  const int nl = int(C.levels.size());
  C.chebA.assign(nl, Chebyshev());
  C.chebS.assign(nl, Chebyshev());
  for (int l = 0; l < nl; ++l) {
    Level *L = &C.levels[l];
    for (int w = 0; w < 2; ++w) {
      Chebyshev &ch = w == 0 ? C.chebA[l] : C.chebS[l];
      if (w == 0) ch.A = [L](Vec &d, const Vec &s) { op_vmult<DIM>(*L, L->cd, M_A, s.data(), nullptr, d.data(), nullptr); };
      else ch.A = [L](Vec &d, const Vec &s) { mass_vmult<DIM>(*L, L->cd, s.data(), d.data()); };
      ch.dinv = w == 0 ? &L->inv_diag_A : &L->inv_diag_M;
      ch.n = w == 0 ? L->ndv() : L->ndp();
      if (l > 0) { ch.degree = C.opt.smoother_degree; ch.range = C.opt.smoothing_range; ch.eig_its = C.opt.eig_cg_n_iterations; }
      else { ch.degree = C.opt.coarse_degree; ch.range = C.opt.coarse_range; ch.eig_its = C.opt.coarse_eig_cg_n_iterations; }
      ch.estimate();
    }
  }
  C.smoothers_ready = true;
}

// ------------------------------------------------------------------ block preconditioner
template <int DIM>
struct BlockPrec {
  Ctx &C;
  bool expensive;
  MG<DIM> mgA, mgS;
  int nA = 0, nS = 0;
  int status = 0;
  BlockPrec(Ctx &c, bool e) : C(c), expensive(e), mgA(c, AGMG_OP_A), mgS(c, AGMG_OP_MASS) {}
  // BlockSchurGMGPreconditioner::vmult (stokes_matrix_free.h:237-354)
  void vmult(Vec &dst, const Vec &src) {
    Level &L = C.act();
    const int nu = L.ndv(), np = L.ndp();
    Vec sp(src.begin() + nu, src.end()), su(src.begin(), src.begin() + nu);
    Vec dp(np, 0.0), du(nu, 0.0);
    if (expensive) {
      const double nrm = norm(sp);
      if (nrm > 1e-50) {
        LinOp A = [&](Vec &d, const Vec &s) { active_vmult(C, AGMG_OP_MASS, d.data(), s.data()); };
        LinOp P = [&](Vec &d, const Vec &s) { mgS.vcycle(d, s); };
        CGResult r = solver_cg(A, P, dp, sp, 100, nrm * C.opt.S_block_tolerance, false, false);
        nS += r.its;
        if (!r.converged) status = -2;
      }
    } else { mgS.vcycle(dp, sp); nS += 1; }
    for (auto &v : dp) v *= -1.0;
    // utmp = src_u - B^T dst_p
    Vec utmp(nu);
    {
      Vec xin(nu + np, 0.0), yout(nu + np);
      std::copy(dp.begin(), dp.end(), xin.begin() + nu);
      active_vmult(C, AGMG_OP_BT, yout.data(), xin.data());
      for (int i = 0; i < nu; ++i) utmp[i] = su[i] - yout[i];
    }
    if (expensive) {
      const double nrm = norm(utmp);
      LinOp A = [&](Vec &d, const Vec &s) { active_vmult(C, AGMG_OP_A, d.data(), s.data()); };
      LinOp P = [&](Vec &d, const Vec &s) { mgA.vcycle(d, s); };
      CGResult r = C.opt.A_block_is_symmetric ? solver_cg(A, P, du, utmp, 1000, nrm * C.opt.A_block_tolerance, false, false)
                                              : solver_bicgstab(A, P, du, utmp, 1000, nrm * C.opt.A_block_tolerance);
      nA += r.its;
      if (!r.converged) status = -3;
    } else { mgA.vcycle(du, utmp); nA += 1; }
    dst.resize(nu + np);
    std::copy(du.begin(), du.end(), dst.begin());
    std::copy(dp.begin(), dp.end(), dst.begin() + nu);
  }
};

// ------------------------------------------------------------------ Krylov
struct Control { int max_steps; double tol; int last_step = 0; double last_value = 0; std::vector<double> history; };
int check(Control &c, int step, double v) {
  c.last_step = step; c.last_value = v; c.history.push_back(v);
  if (v <= c.tol) return ST_SUCCESS;
  if (step >= c.max_steps || !std::isfinite(v)) return ST_FAILURE;
  return ST_ITERATE;
}

// deal.II SolverGMRES with right preconditioning; basis size `restart`;
// x += P^-1 (V y) at convergence/restart (one extra preconditioner call).
int solver_gmres(const LinOp &A, const LinOp &P, Vec &x, const Vec &b, int restart, Control &ctl) {
  const size_t n = b.size();
  std::vector<Vec> V;
  Vec p(n), w(n);
  int acc = 0, state = ST_ITERATE;
  do {
    V.clear();
    A(p, x);
    Vec v0(n);
    for (size_t i = 0; i < n; ++i) v0[i] = b[i] - p[i];
    double rho = norm(v0);
    state = check(ctl, acc, rho);
    if (state != ST_ITERATE) { if (acc > 0 || state == ST_SUCCESS) ; break; }
    // the step-0 check is logged once only
    for (auto &e : v0) e /= rho;
    V.push_back(v0);
    std::vector<double> gamma(restart + 1, 0.0), ci(restart), si(restart);
    std::vector<std::vector<double>> H(restart + 1, std::vector<double>(restart, 0.0));
    gamma[0] = rho;
    int dim = 0;
    for (int j = 0; j < restart && state == ST_ITERATE; ++j) {
      ++acc;
      P(p, V[j]);
      A(w, p);
      std::vector<double> h(j + 2, 0.0);
      for (int pass = 0; pass < 2; ++pass)       // Gram-Schmidt with one re-orthogonalisation
        for (int i = 0; i <= j; ++i) { const double hi = dot(w, V[i]); h[i] += hi; axpy(w, -hi, V[i]); }
      const double s = norm(w);
      h[j + 1] = s;
      if (s != 0) for (auto &e : w) e /= s;
      V.push_back(w);
      for (int i = 0; i < j; ++i) {               // previous Givens rotations
        const double t = h[i];
        h[i] = ci[i] * t + si[i] * h[i + 1];
        h[i + 1] = -si[i] * t + ci[i] * h[i + 1];
      }
      const double r = std::hypot(h[j], h[j + 1]);
      ci[j] = h[j] / r; si[j] = h[j + 1] / r;
      h[j] = r;
      gamma[j + 1] = -si[j] * gamma[j];
      gamma[j] *= ci[j];
      for (int i = 0; i <= j; ++i) H[i][j] = h[i];
      dim = j + 1;
      rho = std::fabs(gamma[dim]);
      state = check(ctl, acc, rho);
    }
    std::vector<double> y(dim);
    for (int i = dim - 1; i >= 0; --i) {
      double s = gamma[i];
      for (int k = i + 1; k < dim; ++k) s -= H[i][k] * y[k];
      y[i] = s / H[i][i];
    }
    std::fill(w.begin(), w.end(), 0.0);
    for (int i = 0; i < dim; ++i) axpy(w, y[i], V[i]);
    P(p, w);
    axpy(x, 1.0, p);
  } while (state == ST_ITERATE);
  return state;
}

// deal.II SolverFGMRES(max_basis_size).  Iteration counting: one count per
// column of the Hessenberg matrix entering the least-squares solve.
int solver_fgmres(const LinOp &A, const LinOp &P, Vec &x, const Vec &b, int basis, Control &ctl) {
  const size_t n = b.size();
  Vec aux(n);
  int acc = 0, state = ST_ITERATE;
  do {
    A(aux, x);
    for (size_t i = 0; i < n; ++i) aux[i] = b[i] - aux[i];
    const double beta = norm(aux);
    state = check(ctl, acc, beta);
    if (state != ST_ITERATE) break;
    std::vector<Vec> V, Z;
    std::vector<double> gamma(basis + 1, 0.0), ci(basis), si(basis);
    std::vector<std::vector<double>> H(basis + 1, std::vector<double>(basis, 0.0));
    gamma[0] = beta;
    Vec v(n);
    for (size_t i = 0; i < n; ++i) v[i] = aux[i] / beta;
    int dim = 0;
    for (int j = 0; j < basis && state == ST_ITERATE; ++j) {
      V.push_back(v);
      Vec z(n);
      P(z, V[j]);
      Z.push_back(z);
      A(aux, z);
      std::vector<double> h(j + 2, 0.0);
      for (int i = 0; i <= j; ++i) { h[i] = dot(aux, V[i]); axpy(aux, -h[i], V[i]); }
      const double a = norm(aux);
      h[j + 1] = a;
      for (size_t i = 0; i < n; ++i) v[i] = a != 0 ? aux[i] / a : 0.0;
      for (int i = 0; i < j; ++i) {
        const double t = h[i];
        h[i] = ci[i] * t + si[i] * h[i + 1];
        h[i + 1] = -si[i] * t + ci[i] * h[i + 1];
      }
      const double r = std::hypot(h[j], h[j + 1]);
      ci[j] = h[j] / r; si[j] = h[j + 1] / r;
      h[j] = r;
      gamma[j + 1] = -si[j] * gamma[j];
      gamma[j] *= ci[j];
      for (int i = 0; i <= j; ++i) H[i][j] = h[i];
      dim = j + 1;
      state = check(ctl, ++acc, std::fabs(gamma[dim]));
    }
    std::vector<double> y(dim);
    for (int i = dim - 1; i >= 0; --i) {
      double s = gamma[i];
      for (int k = i + 1; k < dim; ++k) s -= H[i][k] * y[k];
      y[i] = s / H[i][i];
    }
    for (int i = 0; i < dim; ++i) axpy(x, y[i], Z[i]);
  } while (state == ST_ITERATE);
  return state;
}

// deal.II SolverIDR(s) (right-preconditioned, van Gijzen & Sonneveld biortho variant)
int solver_idr(const LinOp &A, const LinOp &P, Vec &x, const Vec &b, int s, Control &ctl) {
  const size_t n = b.size();
  Vec r(n), v(n), vhat(n), uhat(n), ghat(n);
  A(r, x);
  for (size_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
  double res = norm(r);
  int step = 0;
  int state = check(ctl, step, res);
  if (state != ST_ITERATE) return state;
  std::vector<Vec> G(s, Vec(n, 0.0)), U(s, Vec(n, 0.0)), Q(s, Vec(n));
  std::vector<std::vector<double>> M(s, std::vector<double>(s, 0.0));
  std::mt19937 rng(5489u);
  std::normal_distribution<double> nd(0.0, 1.0);
  for (int i = 0; i < s; ++i) {
    if (i != 0) for (auto &e : Q[i]) e = nd(rng); else Q[i] = r;
    for (int j = 0; j < i; ++j) {
      const double a = dot(Q[j], Q[i]) / dot(Q[j], Q[j]);
      axpy(Q[i], -a, Q[j]);
    }
    if (i != 0) { const double nn = norm(Q[i]); for (auto &e : Q[i]) e /= nn; }
    M[i][i] = 1.0;
  }
  double omega = 1.0;
  bool early = false;
  while (state == ST_ITERATE) {
    ++step;
    std::vector<double> phi(s);
    for (int i = 0; i < s; ++i) phi[i] = dot(Q[i], r);
    for (int k = 0; k < s; ++k) {
      std::vector<double> gamma(s - k);
      for (int i = k; i < s; ++i) {   // forward substitution on M(k:s,k:s)
        double sum = phi[i];
        for (int j = k; j < i; ++j) sum -= M[i][j] * gamma[j - k];
        gamma[i - k] = sum / M[i][i];
      }
      v = r;
      for (int i = k; i < s; ++i) axpy(v, -gamma[i - k], G[i]);
      P(vhat, v);
      for (size_t i = 0; i < n; ++i) uhat[i] = omega * vhat[i];
      for (int i = k; i < s; ++i) axpy(uhat, gamma[i - k], U[i]);
      A(ghat, uhat);
      if (k > 0) {
        double alpha = dot(Q[0], ghat) / M[0][0];
        for (int i = 1; i < k; ++i) {
          axpy(ghat, -alpha, G[i - 1]); axpy(uhat, -alpha, U[i - 1]);
          alpha = dot(Q[i], ghat) / M[i][i];
        }
        axpy(ghat, -alpha, G[k - 1]); axpy(uhat, -alpha, U[k - 1]);
      }
      G[k].swap(ghat); U[k].swap(uhat);
      for (int i = k; i < s; ++i) M[i][k] = dot(Q[i], G[k]);
      const double beta = phi[k] / M[k][k];
      axpy(r, -beta, G[k]);
      axpy(x, beta, U[k]);
      res = norm(r);
      state = check(ctl, step, res);
      if (state != ST_ITERATE) { early = true; break; }
      if (k + 1 < s) {
        for (int i = 0; i < k + 1; ++i) phi[i] = 0.0;
        for (int i = k + 1; i < s; ++i) phi[i] -= beta * M[i][k];
      }
    }
    if (early) break;
    P(vhat, r);
    A(v, vhat);
    omega = dot(v, r) / dot(v, v);
    axpy(r, -omega, v);
    res = norm(r);
    axpy(x, omega, vhat);
    state = check(ctl, step, res);
  }
  return state;
}

// ------------------------------------------------------------------ solve
template <int DIM>
int do_solve(Ctx &C, const double *rhs_in, double *x_io, int newton, agmg_solve_report *rep) {
  Level &L = C.act();
  const int nu = L.ndv(), np = L.ndp(), n = nu + np;
  if (!C.smoothers_ready) setup_smoothers<DIM>(C);
  Vec rhs(rhs_in, rhs_in + n), x(x_io, x_io + n);
  // current_constraints.set_zero(initial guess), local_smoothing.cc:1116
  for (int nd = 0; nd < L.n_nodes_v; ++nd)
    for (int c = 0; c < DIM; ++c) if (L.constrained_v(nd, c)) x[size_t(nd) * DIM + c] = 0.0;
  // (hanging-node entries: zero in the start vector like every constrained entry, identity rows in the operator)
  for (int sp = 0; sp < 2; ++sp)
    for (uint32_t h : L.hang_n[sp])
      for (int c = 0; c < (sp == 0 ? DIM : 1); ++c) x[sp == 0 ? size_t(h) * DIM + c : size_t(nu) + h] = 0.0;
  LinOp A = [&](Vec &d, const Vec &s) {
    d.resize(n);
    active_vmult(C, AGMG_OP_STOKES, d.data(), s.data());
  };
  double tol;
  Vec tmp(n);
  if (!newton) {
    A(tmp, x);
    double s = 0;
    for (int i = 0; i < n; ++i) { const double d = rhs[i] - tmp[i]; s += d * d; }
    rep->initial_nonlinear_residual = std::sqrt(s);
    Vec x0 = x;
    std::fill(x0.begin(), x0.begin() + nu, 0.0);
    A(tmp, x0);
    double ru = 0, rp = 0;
    for (int i = 0; i < nu; ++i) { const double d = rhs[i] - tmp[i]; ru += d * d; }
    for (int i = nu; i < n; ++i) rp += rhs[i] * rhs[i];
    tol = C.opt.linear_tolerance * std::sqrt(ru + rp);
  } else {
    double s = 0;
    for (int i = 0; i < n; ++i) s += rhs[i] * rhs[i];
    tol = C.opt.linear_tolerance * std::sqrt(s);
    rep->initial_nonlinear_residual = std::sqrt(s);
  }
  rep->solver_tolerance = tol;
  BlockPrec<DIM> cheap(C, false), expensive(C, true);
  LinOp Pc = [&](Vec &d, const Vec &s) { cheap.vmult(d, s); };
  LinOp Pe = [&](Vec &d, const Vec &s) { expensive.vmult(d, s); };
  Control cc{C.opt.n_cheap_steps, tol}, ce{C.opt.n_expensive_steps, tol};
  int status = 0;
  int state = ST_FAILURE;
  if (C.opt.n_cheap_steps > 0) {
    if (C.opt.krylov_type == AGMG_KRYLOV_GMRES) state = solver_gmres(A, Pc, x, rhs, C.opt.gmres_restart, cc);
    else state = solver_idr(A, Pc, x, rhs, C.opt.idr_s, cc);
  }
  rep->n_cheap_iterations = cc.last_step;
  rep->n_expensive_iterations = 0;
  rep->final_linear_residual = cc.last_value;
  if (state != ST_SUCCESS) {
    if (C.opt.n_expensive_steps == 0) status = 1;
    else {
      state = solver_fgmres(A, Pe, x, rhs, C.opt.gmres_restart, ce);
      rep->n_expensive_iterations = ce.last_step;
      rep->final_linear_residual = ce.last_value;
      if (state != ST_SUCCESS) status = 1;
      if (expensive.status) status = 1;
    }
  }
  rep->n_A_preconditioner_cheap = cheap.nA; rep->n_S_preconditioner_cheap = cheap.nS;
  rep->n_A_preconditioner_expensive = expensive.nA; rep->n_S_preconditioner_expensive = expensive.nS;
  rep->n_history = 0;
  for (double v : cc.history) if (rep->n_history < 2048) rep->history[rep->n_history++] = v;
  for (double v : ce.history) if (rep->n_history < 2048) rep->history[rep->n_history++] = v;
  // current_constraints.distribute(solution): hanging nodes take the values of their constraints
  for (int sp = 0; sp < 2; ++sp) {
    const int nc = sp == 0 ? DIM : 1;
    const size_t off = sp == 0 ? 0 : nu;
    for (size_t i = 0; i < L.hang_n[sp].size(); ++i)
      for (int c = 0; c < nc; ++c) {
        double v = 0;
        for (int k = L.hang_ptr[sp][i]; k < L.hang_ptr[sp][i + 1]; ++k) v += L.hang_w[sp][k] * x[off + size_t(L.hang_m[sp][k]) * nc + c];
        x[off + size_t(L.hang_n[sp][i]) * nc + c] = v;
      }
  }
  std::copy(x.begin(), x.end(), x_io);
  return status;
}

#define DISPATCH_DIM(C, expr2, expr3) ((C)->opt.dim == 2 ? (expr2) : (expr3))

}  // namespace



// ====================================================================== C API
// correct_stokes_rhs (local_smoothing.cc:761-898): rhs += -M u0 with u0 read WITHOUT the constraints
// (read_dof_values_plain, :813, 817), no Newton terms, no dilation LHS term (only the deviatoric part,
// :835-838); integrate_scatter drops the rows of constrained DoFs; the face term of the free-surface
// stabilisation likewise with plain reads (:853-888).
template <int DIM>
static void rhs_correction_impl(Ctx &C, const double *u0, double *rhs) {
  const Level &L = C.act();
  CellData cd = C.active_cd.set ? C.active_cd : L.cd;
  cd.is_compressible = cd.is_compressible || cd.dilation;
  cd.newton = 0; cd.dilation = 0;
  const int NV = L.nv, NP = L.np, nu = L.ndv();
  const double *su = u0, *sp = u0 + nu;
  double *ru = rhs, *rp = rhs + nu;
  for (int cell = 0; cell < L.n_cells; ++cell) {
    double uin[3][27], pin[8], uout[3][27], pout[8];
    const uint32_t *nv = &L.cnv[size_t(cell) * NV];
    const uint32_t *np = &L.cnp[size_t(cell) * NP];
    for (int i = 0; i < NV; ++i)
      for (int c = 0; c < DIM; ++c) uin[c][i] = su[size_t(nv[i]) * DIM + c];
    for (int i = 0; i < NP; ++i) pin[i] = sp[np[i]];
    cell_apply<DIM>(L, cd, cell, M_STOKES, uin, pin, uout, pout);
    for (int i = 0; i < NV; ++i)
      for (int c = 0; c < DIM; ++c)
        if (!L.constrained_v(nv[i], c)) ru[size_t(nv[i]) * DIM + c] -= uout[c][i];
    for (int i = 0; i < NP; ++i)
      if (!L.constrained_p(np[i])) rp[np[i]] -= pout[i];
  }
  if (cd.n_faces) {   // face_terms() with plain reads
    constexpr int NF = DIM == 3 ? 9 : 3, NQF = NF;
    for (int f = 0; f < cd.n_faces; ++f) {
      const uint32_t *nv = &L.cnv[size_t(cd.face_cell[f]) * L.nv];
      int node[NF];
      double u[NF][DIM], out[NF][DIM];
      for (int a = 0; a < NF; ++a) {
        node[a] = nv[face_node_local<DIM>(cd.face_no[f], a)];
        for (int c = 0; c < DIM; ++c) { u[a][c] = su[size_t(node[a]) * DIM + c]; out[a][c] = 0.0; }
      }
      for (int q = 0; q < NQF; ++q) {
        double shape[NF];
        for (int a = 0; a < NF; ++a) shape[a] = DIM == 3 ? T.N2[q % 3][a % 3] * T.N2[q / 3][a / 3] : T.N2[q][a];
        double n[3], jxw, gu = 0.0;
        face_geometry<DIM>(L, cd, f, q, n, jxw);
        for (int c = 0; c < DIM; ++c) {
          double uq = 0.0;
          for (int a = 0; a < NF; ++a) uq += shape[a] * u[a][c];
          gu += cd.face_stab[(size_t(f) * NQF + q) * DIM + c] * uq;
        }
        for (int a = 0; a < NF; ++a)
          for (int c = 0; c < DIM; ++c) out[a][c] += shape[a] * (-gu * n[c]) * jxw;
      }
      for (int a = 0; a < NF; ++a)
        for (int c = 0; c < DIM; ++c)
          if (!L.constrained_v(node[a], c)) ru[size_t(node[a]) * DIM + c] -= out[a][c];
    }
  }
}
struct oracle_ctx { Ctx c; };

extern "C" {

const char *oracle_last_error(void) { return g_error.c_str(); }

void oracle_default_options(agmg_options *o) {
  std::memset(o, 0, sizeof(*o));
  o->dim = 3; o->degree_v = 2; o->device = 0; o->n_levels = 1;
  o->linear_tolerance = 1e-7; o->n_cheap_steps = 1000; o->n_expensive_steps = 1000;
  o->gmres_restart = 100; o->krylov_type = AGMG_KRYLOV_GMRES; o->idr_s = 2;
  o->A_block_tolerance = 1e-2; o->S_block_tolerance = 1e-6; o->A_block_is_symmetric = 1;
  o->coarse_on_host = 1;
  o->smoother_degree = 4; o->smoothing_range = 15.; o->eig_cg_n_iterations = 10;
  o->coarse_degree = 8; o->coarse_range = 1e-3; o->coarse_eig_cg_n_iterations = 100;
  o->schur_smoother_steps = 4;
  o->world_size = 1; o->rank = 0;
}

oracle_ctx *oracle_create(const agmg_options *opt) {
  if (opt->dim != 2 && opt->dim != 3) { g_error = "dim must be 2 or 3"; return nullptr; }
  if (opt->degree_v != 2) { g_error = "only degree_v = 2"; return nullptr; }
  auto *o = new oracle_ctx();
  o->c.opt = *opt;
  o->c.levels.resize(opt->n_levels);
  return o;
}
void oracle_destroy(oracle_ctx *o) { delete o; }

int oracle_set_level(oracle_ctx *o, int level, const agmg_level_desc *d) {
  Ctx &C = o->c;
  if (level == AGMG_ACTIVE && !d) { C.has_active = false; return 0; }
  if (level != AGMG_ACTIVE && (level < 0 || level >= int(C.levels.size()))) { g_error = "bad level"; return -1; }
  if (level == AGMG_ACTIVE) { C.active = Level(); C.has_active = true; }
  Level &L = level == AGMG_ACTIVE ? C.active : C.levels[level];
  const int dim = C.opt.dim;
  L.dim = dim; L.nv = ipow(3, dim); L.np = ipow(2, dim); L.nq = ipow(3, dim);
  L.n_cells = d->n_cells; L.n_nodes_v = d->n_nodes_v; L.n_nodes_p = d->n_nodes_p;
  L.cnv.assign(d->cell_nodes_v, d->cell_nodes_v + size_t(L.n_cells) * L.nv);
  L.cnp.assign(d->cell_nodes_p, d->cell_nodes_p + size_t(L.n_cells) * L.np);
  L.cmv.assign(d->constrained_v, d->constrained_v + L.n_nodes_v);
  if (d->constrained_p) L.cmp.assign(d->constrained_p, d->constrained_p + L.n_nodes_p);
  else L.cmp.assign(L.n_nodes_p, 0);
  L.geom = d->geometry_kind;
  if (L.geom == AGMG_GEOM_CARTESIAN) L.cell_h.assign(d->cell_h, d->cell_h + size_t(L.n_cells) * dim);
  else {
    L.invJ.assign(d->inv_jacobian, d->inv_jacobian + size_t(L.n_cells) * L.nq * dim * dim);
    L.jxw.assign(d->jxw, d->jxw + size_t(L.n_cells) * L.nq);
  }
  if (d->parent_cell) {
    L.parent.assign(d->parent_cell, d->parent_cell + L.n_cells);
    L.child.assign(d->child_index, d->child_index + L.n_cells);
  }
  for (int sp = 0; sp < 2; ++sp) {
    const int nh = sp == 0 ? d->n_hanging_v : d->n_hanging_p, nn = sp == 0 ? L.n_nodes_v : L.n_nodes_p;
    const uint32_t *hn = sp == 0 ? d->hanging_nodes_v : d->hanging_nodes_p, *hm = sp == 0 ? d->hanging_master_v : d->hanging_master_p;
    const int32_t *hp = sp == 0 ? d->hanging_ptr_v : d->hanging_ptr_p;
    const double *hw = sp == 0 ? d->hanging_weight_v : d->hanging_weight_p;
    L.hang_n[sp].clear(); L.hang_m[sp].clear(); L.hang_ptr[sp].clear(); L.hang_w[sp].clear();
    if (nh <= 0) continue;
    if (level != AGMG_ACTIVE && C.levels.size() != 1) { g_error = "hanging-node constraints belong to the active mesh (multigrid levels have none)"; return -2; }
    if (!hn || !hp || !hm || !hw) { g_error = "incomplete hanging-node tables"; return -1; }
    L.hang_n[sp].assign(hn, hn + nh); L.hang_ptr[sp].assign(hp, hp + nh + 1);
    L.hang_m[sp].assign(hm, hm + hp[nh]); L.hang_w[sp].assign(hw, hw + hp[nh]);
    std::vector<uint8_t> is_h(nn, 0);
    for (uint32_t h : L.hang_n[sp]) { if (int(h) >= nn) { g_error = "hanging node out of range"; return -1; } is_h[h] = 1; }
    for (uint32_t m : L.hang_m[sp]) if (int(m) >= nn || is_h[m]) { g_error = "hanging-node master out of range or hanging itself"; return -1; }
  }
  // refinement edge + copy indices of a level of local smoothing
  L.has_edge = false; L.bnd_v.clear(); L.bnd_p.clear(); L.edge_v.clear(); L.edge_p.clear();
  if (level != AGMG_ACTIVE && (d->refinement_edge_v || d->refinement_edge_p)) {
    if (!d->refinement_edge_v || !d->refinement_edge_p) { g_error = "refinement-edge masks come for both spaces"; return -1; }
    L.edge_v.assign(d->refinement_edge_v, d->refinement_edge_v + L.n_nodes_v);
    L.edge_p.assign(d->refinement_edge_p, d->refinement_edge_p + L.n_nodes_p);
    bool any = false;
    for (uint8_t e : L.edge_v) any |= e != 0;
    for (uint8_t e : L.edge_p) any |= e != 0;
    if (any) {
      L.has_edge = true;
      L.bnd_v = L.cmv; L.bnd_p = L.cmp;
      const uint8_t all = uint8_t((1 << dim) - 1);
      for (int i = 0; i < L.n_nodes_v; ++i) if (L.edge_v[i]) L.cmv[i] |= all;
      for (int i = 0; i < L.n_nodes_p; ++i) if (L.edge_p[i]) L.cmp[i] |= 1;
    }
  }
  for (int sp = 0; sp < 2; ++sp) {
    const int nc = sp == 0 ? d->n_copy_v : d->n_copy_p, nn = sp == 0 ? L.n_nodes_v : L.n_nodes_p;
    const uint32_t *ca = sp == 0 ? d->copy_active_v : d->copy_active_p, *cl = sp == 0 ? d->copy_level_v : d->copy_level_p;
    L.copy_act[sp].clear(); L.copy_lev[sp].clear();
    if (nc <= 0 || level == AGMG_ACTIVE) continue;
    if (!ca || !cl) { g_error = "incomplete copy tables"; return -1; }
    L.copy_act[sp].assign(ca, ca + nc); L.copy_lev[sp].assign(cl, cl + nc);
    for (uint32_t v : L.copy_lev[sp]) if (int(v) >= nn) { g_error = "copy table: level node out of range"; return -1; }
  }
  for (size_t i = 0; i < L.cnv.size(); ++i) if (int(L.cnv[i]) >= L.n_nodes_v) { g_error = "cell_nodes_v out of range"; return -1; }
  for (size_t i = 0; i < L.cnp.size(); ++i) if (int(L.cnp[i]) >= L.n_nodes_p) { g_error = "cell_nodes_p out of range"; return -1; }
  L.set = true;
  C.smoothers_ready = false;
  return 0;
}

int oracle_set_coefficients(oracle_ctx *o, int level, const agmg_cell_data *d) {
  Ctx &C = o->c;
  CellData *cd;
  int nc;
  if (level == AGMG_ACTIVE) { cd = &C.active_cd; nc = C.act().n_cells; }
  else { cd = &C.levels[level].cd; nc = C.levels[level].n_cells; }
  cd->is_compressible = d->is_compressible; cd->newton = 0; cd->dilation = 0; cd->symmetrize = d->symmetrize_newton_system;
  cd->dil_lhs.clear(); cd->dil_de.clear(); cd->dil_dp.clear();
  if (d->enable_prescribed_dilation) {  // active level only (local_smoothing.cc:456-462)
    if (level != AGMG_ACTIVE) { g_error = "dilation tables exist on the active level only"; return -1; }
    if (!d->dilation_lhs_term) { g_error = "dilation_lhs_term missing"; return -1; }
    const int dim = C.opt.dim, nq = ipow(3, dim), ns = dim * (dim + 1) / 2;
    cd->dil_lhs.assign(d->dilation_lhs_term, d->dilation_lhs_term + size_t(nc) * nq);
    if (d->enable_newton_derivatives && d->dilation_derivative_wrt_strain_rate && d->dilation_derivative_wrt_pressure) {
      cd->dil_de.assign(d->dilation_derivative_wrt_strain_rate, d->dilation_derivative_wrt_strain_rate + size_t(nc) * nq * ns);
      cd->dil_dp.assign(d->dilation_derivative_wrt_pressure, d->dilation_derivative_wrt_pressure + size_t(nc) * nq);
    }
    cd->dilation = 1;
  }
  if (d->enable_newton_derivatives) {  // active level only (local_smoothing.cc:447-663)
    if (level != AGMG_ACTIVE) { g_error = "Newton tables exist on the active level only"; return -1; }
    if (!d->strain_rate_table || !d->newton_factor_wrt_strain_rate || !d->newton_factor_wrt_pressure) { g_error = "Newton tables missing"; return -1; }
    const int dim = C.opt.dim, nq = ipow(3, dim), ns = dim * (dim + 1) / 2;
    cd->strain.assign(d->strain_rate_table, d->strain_rate_table + size_t(nc) * nq * ns);
    cd->dEde.assign(d->newton_factor_wrt_strain_rate, d->newton_factor_wrt_strain_rate + size_t(nc) * nq * ns);
    cd->dEdp.assign(d->newton_factor_wrt_pressure, d->newton_factor_wrt_pressure + size_t(nc) * nq);
    cd->newton = 1;
  }
  cd->lambda = d->pressure_scaling; cd->nqv = d->n_q_viscosity;
  cd->eta.assign(d->viscosity, d->viscosity + size_t(nc) * cd->nqv);
  cd->set = true;
  C.smoothers_ready = false;
  return 0;
}

int oracle_set_free_surface_faces(oracle_ctx *o, const agmg_face_data *d) {
  Ctx &C = o->c;
  CellData &cd = C.active_cd;
  const int dim = C.opt.dim, nqf = ipow(3, dim - 1);
  cd.n_faces = 0;
  cd.face_cell.clear(); cd.face_no.clear(); cd.face_stab.clear(); cd.face_normal.clear(); cd.face_jxw.clear();
  if (!d || d->n_faces == 0) return 0;
  if (!d->face_cell || !d->face_number || !d->stabilization_term) { g_error = "free-surface face tables missing"; return -1; }
  if ((d->normal == nullptr) != (d->jxw == nullptr)) { g_error = "face normal and jxw come together"; return -1; }
  if (!d->normal && C.act().geom != AGMG_GEOM_CARTESIAN) { g_error = "general geometry needs face normals and JxW"; return -1; }
  for (int f = 0; f < d->n_faces; ++f)
    if (int(d->face_cell[f]) >= C.act().n_cells || d->face_number[f] >= 2 * dim) { g_error = "face table out of range"; return -1; }
  cd.n_faces = d->n_faces;
  cd.face_cell.assign(d->face_cell, d->face_cell + d->n_faces);
  cd.face_no.assign(d->face_number, d->face_number + d->n_faces);
  cd.face_stab.assign(d->stabilization_term, d->stabilization_term + size_t(d->n_faces) * nqf * dim);
  if (d->normal) {
    cd.face_normal.assign(d->normal, d->normal + size_t(d->n_faces) * nqf * dim);
    cd.face_jxw.assign(d->jxw, d->jxw + size_t(d->n_faces) * nqf);
  }
  return 0;
}

int oracle_compute_diagonals(oracle_ctx *o) {
  for (auto &L : o->c.levels) { if (o->c.opt.dim == 2) compute_diagonals<2>(L); else compute_diagonals<3>(L); }
  o->c.smoothers_ready = false;
  return 0;
}

static void vmult_impl(Ctx &C, int which, int level, double *dst, const double *src, bool dense);

// MatrixFree's treatment of hanging-node constraints around an operator that treats every node as its own DoF
// (read_dof_values resolves the constraints, distribute_local_to_global applies their transpose, constrained rows are
// identity rows): y = C^T A C x on the unconstrained DoFs.  Masters with a homogeneous Dirichlet bit contribute zero and
// receive nothing (AffineConstraints merges the two kinds of constraints that way).
static void vmult_hanging(Ctx &C, int which, int level, double *dst, const double *src, bool dense) {
  Level &L = const_cast<Level &>(level == AGMG_ACTIVE ? C.act() : C.levels[level]);
  const int dim = L.dim, nu = L.ndv(), np_ = L.ndp();
  const bool has_u = which != AGMG_OP_MASS, has_p = which != AGMG_OP_A;
  const size_t n = (has_u ? size_t(nu) : 0) + (has_p ? size_t(np_) : 0);     // A: u; mass: p; Stokes, B^T: [u; p]
  Vec xs(src, src + n), ys(n, 0.0);
  const size_t poff = has_u ? nu : 0;                        // offset of the pressure block in the vector
  auto resolve = [&](int sp, size_t off, int nc) {
    for (size_t i = 0; i < L.hang_n[sp].size(); ++i) {
      const uint32_t h = L.hang_n[sp][i];
      for (int c = 0; c < nc; ++c) {
        double v = 0;
        for (int k = L.hang_ptr[sp][i]; k < L.hang_ptr[sp][i + 1]; ++k) {
          const uint32_t m = L.hang_m[sp][k];
          const bool con = sp == 0 ? L.constrained_v(m, c) : L.constrained_p(m);
          if (!con) v += L.hang_w[sp][k] * src[off + size_t(m) * nc + c];
        }
        xs[off + size_t(h) * nc + c] = v;
      }
    }
  };
  auto distribute = [&](int sp, size_t off, int nc) {
    for (size_t i = 0; i < L.hang_n[sp].size(); ++i) {
      const uint32_t h = L.hang_n[sp][i];
      for (int c = 0; c < nc; ++c) {
        const bool con_h = sp == 0 ? L.constrained_v(h, c) : L.constrained_p(h);
        if (!con_h)
          for (int k = L.hang_ptr[sp][i]; k < L.hang_ptr[sp][i + 1]; ++k) {
            const uint32_t m = L.hang_m[sp][k];
            const bool con = sp == 0 ? L.constrained_v(m, c) : L.constrained_p(m);
            if (!con) ys[off + size_t(m) * nc + c] += L.hang_w[sp][k] * ys[off + size_t(h) * nc + c];
          }
        ys[off + size_t(h) * nc + c] = src[off + size_t(h) * nc + c];      // constrained row: identity
      }
    }
  };
  if (has_u) resolve(0, 0, dim);
  if (has_p) resolve(1, poff, 1);
  // the operator itself, on a level without the tables (every node its own DoF)
  std::vector<uint32_t> hn[2] = {std::move(L.hang_n[0]), std::move(L.hang_n[1])};
  L.hang_n[0].clear(); L.hang_n[1].clear();
  vmult_impl(C, which, level, ys.data(), xs.data(), dense);
  L.hang_n[0] = std::move(hn[0]); L.hang_n[1] = std::move(hn[1]);
  if (has_u) distribute(0, 0, dim);
  if (which == AGMG_OP_MASS || which == AGMG_OP_STOKES) distribute(1, poff, 1);     // (B^T writes no pressure rows)
  std::copy(ys.begin(), ys.end(), dst);
}

static void vmult_impl(Ctx &C, int which, int level, double *dst, const double *src, bool dense) {
  const Level &L = level == AGMG_ACTIVE ? C.act() : C.levels[level];
  const CellData &cd = level == AGMG_ACTIVE ? C.active_cd : L.cd;
  const int nu = L.ndv();
  if (L.has_hanging()) { vmult_hanging(C, which, level, dst, src, dense); return; }
  if (dense) { if (C.opt.dim == 2) dense_vmult<2>(L, cd, which, src, dst); else dense_vmult<3>(L, cd, which, src, dst); return; }
  if (C.opt.dim == 2) {
    if (which == AGMG_OP_MASS) mass_vmult<2>(L, cd, src, dst);
    else if (which == AGMG_OP_A) op_vmult<2>(L, cd, M_A, src, nullptr, dst, nullptr);
    else op_vmult<2>(L, cd, which == AGMG_OP_BT ? M_BT : M_STOKES, src, src + nu, dst, dst + nu);
  } else {
    if (which == AGMG_OP_MASS) mass_vmult<3>(L, cd, src, dst);
    else if (which == AGMG_OP_A) op_vmult<3>(L, cd, M_A, src, nullptr, dst, nullptr);
    else op_vmult<3>(L, cd, which == AGMG_OP_BT ? M_BT : M_STOKES, src, src + nu, dst, dst + nu);
  }
}
extern "C++" {
namespace {
void active_vmult(Ctx &C, int which, double *dst, const double *src) { vmult_impl(C, which, AGMG_ACTIVE, dst, src, false); }
}
}
int oracle_rhs_correction(oracle_ctx *o, const double *u0, double *rhs) {
  if (o->c.opt.dim == 2) rhs_correction_impl<2>(o->c, u0, rhs); else rhs_correction_impl<3>(o->c, u0, rhs);
  return 0;
}

int oracle_vmult(oracle_ctx *o, int which, int level, double *dst, const double *src) { vmult_impl(o->c, which, level, dst, src, false); return 0; }
int oracle_vmult_dense(oracle_ctx *o, int which, int level, double *dst, const double *src) { vmult_impl(o->c, which, level, dst, src, true); return 0; }

// ---- CPU baseline leg of bench.py: the same operators with SIMD across cells (oracle/cpu_simd.cpp), OpenMP
// threads over the cell batches of a colour.  Returns the SIMD lanes used, or -1 if the level is not covered
// (general geometry, Newton / dilation / face terms, 2-D).
extern "C" {
void cpu_simd_cells_v3(int, int, int, const int *, const uint32_t *, const uint32_t *, const uint8_t *, const uint8_t *,
                       const double *, const double *, int, double, const double *, const double *, double *, double *);
void cpu_simd_cells_v4(int, int, int, const int *, const uint32_t *, const uint32_t *, const uint8_t *, const uint8_t *,
                       const double *, const double *, int, double, const double *, const double *, double *, double *);
}
int oracle_vmult_simd(oracle_ctx *o, int which, int level, double *dst, const double *src) {
  Ctx &C = o->c;
  const Level &L = level == AGMG_ACTIVE ? C.act() : C.levels[level];
  const CellData &cd = level == AGMG_ACTIVE ? C.active_cd : L.cd;
  if (C.opt.dim != 3 || L.geom != AGMG_GEOM_CARTESIAN || cd.newton || cd.dilation || cd.n_faces || which == AGMG_OP_MASS) return -1;
  const bool wide = __builtin_cpu_supports("avx512f");
  const int lanes = wide ? 8 : 4;
  auto kernel = wide ? cpu_simd_cells_v4 : cpu_simd_cells_v3;
  const int mode = which == AGMG_OP_STOKES ? 0 : (which == AGMG_OP_A ? 1 : 2);
  const int nu = L.ndv();
  const double *su = src, *sp = which == AGMG_OP_A ? nullptr : src + nu;
  double *du = dst, *dp = which == AGMG_OP_A ? nullptr : dst + nu;
  const auto &colours = cell_colours(L);
  const int nthreads = oracle_threads();
  const int ndp = dp ? L.ndp() : 0;
  // one parallel region per application: zeroing, the colours one after the other (a barrier between two colours;
  // every thread takes the same contiguous run of a colour's batches in every application, so its cells stay in its
  // caches), the identity rows
#pragma omp parallel num_threads(nthreads)
  {
#pragma omp for schedule(static)
    for (int i = 0; i < nu; ++i) du[i] = 0.0;
#pragma omp for schedule(static)
    for (int i = 0; i < ndp; ++i) dp[i] = 0.0;
    for (const auto &col : colours) {
      const int n = (int)col.size();
      const int chunk = 8 * lanes, n_batches = (n + chunk - 1) / chunk;
#pragma omp for schedule(static)
      for (int b = 0; b < n_batches; ++b) {
        const int first = b * chunk, cnt = std::min(chunk, n - first);
        kernel(mode, cd.is_compressible ? 1 : 0, cnt, col.data() + first, L.cnv.data(), L.cnp.data(), L.cmv.data(), L.cmp.data(),
               L.cell_h.data(), cd.eta.data(), cd.nqv, cd.lambda, su, sp, du, dp);
      }
    }
    // MatrixFreeOperators::Base::vmult: identity rows (B^T: of the velocity block, reading src_u)
#pragma omp for schedule(static)
    for (int n = 0; n < L.n_nodes_v; ++n)
      if (L.cmv[n])
        for (int c = 0; c < 3; ++c)
          if (L.constrained_v(n, c)) du[size_t(n) * 3 + c] += su[size_t(n) * 3 + c];
    if (dp && sp) {
#pragma omp for schedule(static)
      for (int n = 0; n < L.n_nodes_p; ++n)
        if (L.constrained_p(n)) dp[n] += sp[n];
    }
  }
  return lanes;
}

// The reference's harness protocol inside the library (local_smoothing.cc:1236-1264: warm-up, then the mean over
// n_repeat applications), on buffers allocated once: seconds per application.  Returns the SIMD lanes, or -1.
int oracle_bench_vmult_simd(oracle_ctx *o, int which, int level, int n_repeat, const double *src, double *seconds) {
  Ctx &C = o->c;
  const Level &L = level == AGMG_ACTIVE ? C.act() : C.levels[level];
  const size_t n = (size_t)L.ndv() + (which == AGMG_OP_A ? 0 : (size_t)L.ndp());
  std::vector<double> dst(n);
  int lanes = oracle_vmult_simd(o, which, level, dst.data(), src);
  if (lanes < 0 || n_repeat < 1) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_repeat; ++i) lanes = oracle_vmult_simd(o, which, level, dst.data(), src);
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n_repeat;
  return lanes;
}

int oracle_get_inverse_diagonal(oracle_ctx *o, int which, int level, double *dst) {
  const Level &L = o->c.levels[level];
  const Vec &v = which == AGMG_OP_A ? L.inv_diag_A : L.inv_diag_M;
  std::copy(v.begin(), v.end(), dst);
  return 0;
}

int oracle_prolongate_add(oracle_ctx *o, int which, int to_level, double *dst, const double *src) {
  Ctx &C = o->c;
  const int nd = which == AGMG_OP_A ? C.levels[to_level].ndv() : C.levels[to_level].ndp();
  const int ns = which == AGMG_OP_A ? C.levels[to_level - 1].ndv() : C.levels[to_level - 1].ndp();
  Vec d(dst, dst + nd), s(src, src + ns);
  if (C.opt.dim == 2) transfer<2>(C, which, to_level, true, d, s); else transfer<3>(C, which, to_level, true, d, s);
  std::copy(d.begin(), d.end(), dst);
  return 0;
}
int oracle_restrict_add(oracle_ctx *o, int which, int from_level, double *dst, const double *src) {
  Ctx &C = o->c;
  const int ns = which == AGMG_OP_A ? C.levels[from_level].ndv() : C.levels[from_level].ndp();
  const int nd = which == AGMG_OP_A ? C.levels[from_level - 1].ndv() : C.levels[from_level - 1].ndp();
  Vec d(dst, dst + nd), s(src, src + ns);
  if (C.opt.dim == 2) transfer<2>(C, which, from_level, false, d, s); else transfer<3>(C, which, from_level, false, d, s);
  std::copy(d.begin(), d.end(), dst);
  return 0;
}

int oracle_setup_smoothers(oracle_ctx *o) {
  if (o->c.opt.dim == 2) setup_smoothers<2>(o->c); else setup_smoothers<3>(o->c);
  return 0;
}
int oracle_get_eigenvalue_estimates(oracle_ctx *o, int which, int level, double *lmin, double *lmax) {
  const Chebyshev &c = which == AGMG_OP_A ? o->c.chebA[level] : o->c.chebS[level];
  *lmin = c.lmin; *lmax = c.lmax;
  return 0;
}

int oracle_vcycle(oracle_ctx *o, int which, double *dst, const double *src) {
  Ctx &C = o->c;
  const int n = which == AGMG_OP_A ? C.act().ndv() : C.act().ndp();
  Vec d(n), s(src, src + n);
  if (C.opt.dim == 2) { if (!C.smoothers_ready) setup_smoothers<2>(C); MG<2> mg(C, which); mg.vcycle(d, s); }
  else { if (!C.smoothers_ready) setup_smoothers<3>(C); MG<3> mg(C, which); mg.vcycle(d, s); }
  std::copy(d.begin(), d.end(), dst);
  return 0;
}

int oracle_precondition(oracle_ctx *o, int expensive, double *dst, const double *src) {
  Ctx &C = o->c;
  const int n = C.act().ndv() + C.act().ndp();
  Vec d(n), s(src, src + n);
  if (C.opt.dim == 2) { if (!C.smoothers_ready) setup_smoothers<2>(C); BlockPrec<2> p(C, expensive != 0); p.vmult(d, s); }
  else { if (!C.smoothers_ready) setup_smoothers<3>(C); BlockPrec<3> p(C, expensive != 0); p.vmult(d, s); }
  std::copy(d.begin(), d.end(), dst);
  return 0;
}

int oracle_solve(oracle_ctx *o, const double *rhs, double *x, int newton, agmg_solve_report *rep) {
  std::memset(rep, 0, sizeof(*rep));
  return o->c.opt.dim == 2 ? do_solve<2>(o->c, rhs, x, newton, rep) : do_solve<3>(o->c, rhs, x, newton, rep);
}

}  // extern "C"
