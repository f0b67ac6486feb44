// kernels_ops.hip -- Q2-Q1 sum-factorised cell kernels and node-owner pull kernels (gfx950).
//
// What these replace in the reference (tjhei/aspect 3.1.0-pre):
//   k_cells<OP_A>       ABlockOperator::local_apply        matrix_free_operators.cc:705-783
//   k_cells<OP_STOKES>  StokesOperator::local_apply        matrix_free_operators.cc:171-278
//   k_cells<OP_BT>      BTBlockOperator::local_apply       matrix_free_operators.cc:399-432
//   k_mass              MassMatrixOperator::local_apply    matrix_free_operators.cc:518-576
//   k_diag_*            compute_diagonal                   matrix_free_operators.cc:592-639, 799-829
//   k_pull              deal.II MatrixFreeOperators::Base::vmult wrapper (dst = 0, constrained rows
//                       identity) fused with the Chebyshev / residual vector update
//
// Design: one lane per (cell, velocity component); 21 cells (3-D) or 32 cells (2-D)
// per wavefront.  All 1-D contractions (3x3) run in registers; the only cross-lane step is the
// transpose of the velocity gradient (sym grad needs du_d/dx_c from the lane of component d),
// staged through a wave-private LDS tile, one quadrature plane at a time.  Scatter is atomic-free
// and deterministic: every cell writes its contributions to a slot array laid out
// [local dof][cell][component] (fully coalesced), and a node-owner pull kernel sums the slots of
// each DoF in a fixed order and applies the vector epilogue.
#include "agmg_internal.h"
#include "kernels_cell.h"

#include <cstdio>

namespace agmg {

namespace {

// slot variant: every cell writes its contributions to slots [local dof][cell][component]
template <int DIM, int OP, bool COMP, bool PERQ, bool GEN, bool NEWTON = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, NEWTON ? 1 : AGMG_CELLS_WAVES) k_cells(const CellArgs A) {
  constexpr int NV = KDims<DIM>::NV, NP = KDims<DIM>::NP, CPW = KDims<DIM>::CPW, LROW = KDims<DIM>::LROW;
  __shared__ double xbuf[WAVES_PER_BLOCK][3][LROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int j = lane / DIM;
  const int c = lane - j * DIM;
  const bool lane_ok = j < CPW;
  if (!lane_ok) j = CPW - 1;
  const long cell_l = ((long)blockIdx.x * WAVES_PER_BLOCK + wave) * CPW + j;
  const bool valid = lane_ok && cell_l < A.n_cells;
  const int cell = cell_l < A.n_cells ? (int)cell_l : A.n_cells - 1;
  const size_t cs = (size_t)A.cstride;
  double G[NV], pout[NP];
  double u[NV];
  WaveXch<DIM> ex{xbuf[wave], lane, j};
  cell_compute<DIM, OP, COMP, PERQ, false, true, GEN, NEWTON>(A, cell, c, j, lane, ex, G, pout, nullptr, u);
  if (OP == KOP_STOKES) sum_over_components<DIM, NP>(pout);
  if (valid) {
#pragma unroll
    for (int a = 0; a < NV; ++a)
#if AGMG_ABLATE & 1
      if (G[a] == 12345.678)
#endif
      A.slots_u[((size_t)a * cs + cell) * DIM + c] = G[a];
    if (OP == KOP_STOKES && c == 0) {
#pragma unroll
      for (int b = 0; b < NP; ++b) A.slots_p[(size_t)b * cs + cell] = pout[b];
    }
  }
}

// ---- pressure mass matrix: one lane per cell ---------------------------------------------
template <int DIM, bool PERQ, bool GEN>
__global__ void __launch_bounds__(256) k_mass(const CellArgs A) {
  constexpr int NP = (DIM == 3) ? 8 : 4;
  constexpr int NQ = (DIM == 3) ? 27 : 9;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= A.n_cells) return;
  const size_t cs = (size_t)A.cstride;
  double det = 1.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) det *= GEN ? 1.0 : A.hT[d * cs + cell];
  const double scale = A.lambda * A.lambda * det;
  double pin[NP];
#pragma unroll
  for (int b = 0; b < NP; ++b) {
    const uint32_t idx = A.cnpT[b * cs + cell];
    const double v = A.src_p[idx & NODE_MASK];
    pin[b] = ((idx >> 29) & 1u) ? 0.0 : v;
  }
  const double inv_eta_c = PERQ ? 1.0 : 1.0 / A.etaT[cell];
  double pout[NP];
#pragma unroll
  for (int b = 0; b < NP; ++b) pout[b] = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int q0 = q % 3, q1 = (q / 3) % 3, q2 = q / 9;
    double v = 0.0;
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      const double w = n1(q0, b & 1) * n1(q1, (b >> 1) & 1) * (DIM == 3 ? n1(q2, (b >> 2) & 1) : 1.0);
      v = fma(w, pin[b], v);
    }
    const double wq = wq1(q0) * wq1(q1) * (DIM == 3 ? wq1(q2) : 1.0);
    // per-lane division as in matrix_free_operators.cc:556-563
    const double ie = PERQ ? 1.0 / A.etaT[(size_t)q * cs + cell] : inv_eta_c;
    v *= (GEN ? A.jxwT[(size_t)q * cs + cell] : wq) * scale * ie;
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      const double w = n1(q0, b & 1) * n1(q1, (b >> 1) & 1) * (DIM == 3 ? n1(q2, (b >> 2) & 1) : 1.0);
      pout[b] = fma(w, v, pout[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < NP; ++b) A.slots_p[(size_t)b * cs + cell] = pout[b];
}

// ---- assembled rows of the pressure mass matrix ------------------------------------------------
// The Q1 mass operator has at most 3^DIM entries per row: 12 B per entry, i.e. no more bytes per
// application than the per-q coefficient table of the matrix-free form, and one node-owner launch
// with the vector epilogue fused instead of cell kernel + slots + pull.  Rows are assembled on the
// device whenever the coefficients change: one thread per row walks the cells around its node in
// the fixed order of the pull list (deterministic), computes its row of each cell matrix
// sum_q N_b N_j (JxW lambda^2 / eta_q) and adds it to the row's entries.
struct EllArgs {
  const uint32_t *col;
  double *val;
  int n_nodes, stride;
  const uint32_t *ptr, *list;
};
template <int DIM, bool PERQ, bool GEN>
__global__ void __launch_bounds__(256) k_mass_ell_assemble(const CellArgs A, const EllArgs M) {
  constexpr int NP = (DIM == 3) ? 8 : 4, NQ = (DIM == 3) ? 27 : 9, W = NQ;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= M.n_nodes) return;
  const size_t cs = (size_t)A.cstride, st = (size_t)M.stride;
  uint32_t col[W];
  double val[W];
#pragma unroll
  for (int kk = 0; kk < W; ++kk) { col[kk] = M.col[kk * st + n]; val[kk] = 0.0; }
  for (uint32_t k = M.ptr[n]; k < M.ptr[n + 1]; ++k) {
    const uint32_t ent = M.list[k];
    const int b = (int)(ent / cs);
    if (b >= NP) continue;   // (distributed levels: an entry of the receive buffer, not a cell)
    const size_t cell = ent - (size_t)b * cs;
    double det = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) det *= GEN ? 1.0 : A.hT[d * cs + cell];
    const double scale = A.lambda * A.lambda * det;
    const double inv_eta_c = PERQ ? 1.0 : 1.0 / A.etaT[cell];
    double m[NP];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) m[jj] = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int q0 = q % 3, q1 = (q / 3) % 3, q2 = q / 9;
      const double wq = wq1(q0) * wq1(q1) * (DIM == 3 ? wq1(q2) : 1.0);
      const double ie = PERQ ? 1.0 / A.etaT[(size_t)q * cs + cell] : inv_eta_c;
      const double wb = n1(q0, b & 1) * n1(q1, (b >> 1) & 1) * (DIM == 3 ? n1(q2, (b >> 2) & 1) : 1.0);
      const double f = wb * (GEN ? A.jxwT[(size_t)q * cs + cell] : wq) * scale * ie;
#pragma unroll
      for (int jj = 0; jj < NP; ++jj)
        m[jj] = fma(n1(q0, jj & 1) * n1(q1, (jj >> 1) & 1) * (DIM == 3 ? n1(q2, (jj >> 2) & 1) : 1.0), f, m[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
      const uint32_t idx = A.cnpT[jj * cs + cell];
      if ((idx >> 29) & 1u) continue;   // constrained source entries read as zero
      const uint32_t nj = idx & NODE_MASK;
      // (unused entries of the row repeat the row's own node n: only the first match may take the value)
      bool placed = false;
#pragma unroll
      for (int kk = 0; kk < W; ++kk) {
        const bool hit = !placed && col[kk] == nj;
        val[kk] += hit ? m[jj] : 0.0;
        placed = placed || hit;
      }
    }
  }
#pragma unroll
  for (int kk = 0; kk < W; ++kk) M.val[kk * st + n] = val[kk];
}

struct EllMulArgs {
  const uint32_t *col;
  const double *val;
  int n_nodes, stride;
  const uint8_t *cmask;
  const double *src;
  EpiArgs e;
  // distributed levels: rows [node0, node0 + n_nodes).  ghost_out != nullptr: the raw sums of the ghost rows (the
  // shares of this rank's cells) go to ghost_out[row - node0] for the compress step; else the rows are owned rows and
  // those another rank ghosts (xrow[n] >= 0) add the received shares recv[xent[k]], k in [xptr[xrow], xptr[xrow + 1])
  int node0;
  double *ghost_out;
  const int32_t *xrow;
  const uint32_t *xptr, *xent;
  const double *recv;
  // symmetric exchange (SymExchange): srow != nullptr: a shared row takes its own share from ypart and the others'
  // from the receive buffer, in rank order
  const int32_t *srow;
  const uint32_t *sptr, *spos;
  const uint8_t *ssplit;
  const double *ypart, *srecv;
  // columns by class (DeviceLevel::ell_cls): column k of row n = n + cls_off[cls[n]][k]
  const uint16_t *cls;
  const int32_t *cls_off;
};
struct EllSymArgs {
  const uint32_t *col;
  const double *val;
  int n_sh, stride;
  const double *src;
  const uint32_t *sh_nodes, *sptr, *spos;
  double *ypart, *sendbuf;
};
// shares of the rows other ranks hold too: into ypart, and into every message that carries the row
template <int W>
__global__ void __launch_bounds__(256) k_mass_ell_sym_a(const EllSymArgs M) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= M.n_sh) return;
  const uint32_t n = M.sh_nodes[s];
  const size_t st = (size_t)M.stride;
  uint32_t col[W];
  double val[W];
#pragma unroll
  for (int k = 0; k < W; ++k) { col[k] = M.col[k * st + n]; val[k] = M.val[k * st + n]; }
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) sum = fma(val[k], M.src[col[k]], sum);
  M.ypart[s] = sum;
  for (uint32_t k = M.sptr[s]; k < M.sptr[s + 1]; ++k) M.sendbuf[M.spos[k]] = sum;
}
template <int W>
__global__ void __launch_bounds__(256) k_mass_ell(const EllMulArgs M) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M.n_nodes) return;
  const int n = M.node0 + i;
  const size_t st = (size_t)M.stride;
  const int sr = M.srow ? M.srow[n] : -1;
  double sum = 0.0;
  if (sr >= 0) {
    const uint32_t s0 = M.sptr[sr], s1 = M.sptr[sr + 1], mine = s0 + M.ssplit[sr];
    const double own = M.ypart[sr];
    for (uint32_t k = s0; k < s1; ++k) {
      if (k == mine) sum += own;
      sum += M.srecv[M.spos[k]];
    }
    if (mine == s1) sum += own;
  } else {
    uint32_t col[W];
    double val[W];
    if (M.cls) {
      const int32_t *off = M.cls_off + (size_t)M.cls[n] * W;
#pragma unroll
      for (int k = 0; k < W; ++k) { col[k] = (uint32_t)(n + off[k]); val[k] = M.val[k * st + n]; }
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) { col[k] = M.col[k * st + n]; val[k] = M.val[k * st + n]; }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) sum = fma(val[k], M.src[col[k]], sum);
  }
  if (M.ghost_out) { M.ghost_out[i] = sum; return; }
  if (M.xrow) {
    const int xr = M.xrow[n];
    if (xr >= 0)
      for (uint32_t k = M.xptr[xr]; k < M.xptr[xr + 1]; ++k) sum += M.recv[M.xent[k]];
  }
  const bool con = M.cmask ? (M.cmask[n] & 1u) : false;
  double out;
  if (epi_value(M.e, (size_t)n, sum, con, out)) M.e.dst[n] = out;
}

// ---- fused Chebyshev smoother of the mass operator, one workgroup (agmg_internal.h) ----------
struct MassSmArgs {
  const uint32_t *ptr, *list;
  const uint8_t *cmask;
  const double *dinv, *b, *x_in;
  double *x_out, *resid_out;
  int n_nodes;
};
template <int DIM, bool PERQ, bool GEN>
__global__ void __launch_bounds__(MS_THREADS) k_mass_smoother(const CellArgs A, const MassSmArgs M, const SmootherProgram P) {
  constexpr int NP = (DIM == 3) ? 8 : 4;
  constexpr int NQ = (DIM == 3) ? 27 : 9;
  __shared__ double slots[MS_MAX_SLOTS];
  __shared__ double X[MS_MAX_NODES];
  const int t = threadIdx.x;
  const size_t cs = (size_t)A.cstride;
  const int CS = A.n_cells;   // slot row stride
  const bool is_cell = t < A.n_cells, is_node = t < M.n_nodes;
  // cell role: node indices and quadrature coefficients of my cell, kept in registers
  uint32_t idx[NP];
  constexpr bool TAB = PERQ || GEN;      // a coefficient per q-point (else weight * one cell factor)
  double coef[TAB ? NQ : 1];
  double scale = 0.0, ie_c = 0.0;
  if (is_cell) {
#pragma unroll
    for (int b = 0; b < NP; ++b) idx[b] = A.cnpT[b * cs + t];
    double det = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) det *= GEN ? 1.0 : A.hT[d * cs + t];
    scale = A.lambda * A.lambda * det;
    ie_c = PERQ ? 1.0 : 1.0 / A.etaT[t];
    if (TAB) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int q0 = q % 3, q1 = (q / 3) % 3, q2 = q / 9;
        const double wq = wq1(q0) * wq1(q1) * (DIM == 3 ? wq1(q2) : 1.0);
        const double ie = PERQ ? 1.0 / A.etaT[(size_t)q * cs + t] : ie_c;
        coef[q] = (GEN ? A.jxwT[(size_t)q * cs + t] : wq) * scale * ie;   // product order of k_mass
      }
    }
  }
  // node role: my row of the pull CSR (as LDS slot indices) and my vector entries
  int ent[8], cnt = 0;
  double x = 0.0, xo = 0.0, bv = 0.0, dv = 0.0;
  bool con = false;
  if (is_node) {
    const uint32_t s0 = M.ptr[t];
    cnt = (int)(M.ptr[t + 1] - s0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ent[k] = 0;
      if (k < cnt) {
        const uint32_t e = M.list[s0 + k];
        const uint32_t a = e / (uint32_t)cs;
        ent[k] = (int)(a * CS + (e - a * (uint32_t)cs));
      }
    }
    bv = M.b[t];
    dv = M.dinv[t];
    con = M.cmask ? (M.cmask[t] & 1u) : false;
    if (M.x_in) x = M.x_in[t];
    X[t] = x;
  }
  __syncthreads();
  for (int st = 0; st < P.n; ++st) {
    const SmootherStep S = P.s[st];
    if (S.kind == SM_POINTWISE) {
      if (is_node) { x = S.f2 * dv * bv; X[t] = x; }
      __syncthreads();
      continue;
    }
    if (is_cell) {   // MassMatrixOperator::local_apply of my cell, as k_mass
      double pin[NP], pout[NP];
#pragma unroll
      for (int b = 0; b < NP; ++b) {
        const double v = X[idx[b] & NODE_MASK];
        pin[b] = ((idx[b] >> 29) & 1u) ? 0.0 : v;
        pout[b] = 0.0;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int q0 = q % 3, q1 = (q / 3) % 3, q2 = q / 9;
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < NP; ++b) {
          const double w = n1(q0, b & 1) * n1(q1, (b >> 1) & 1) * (DIM == 3 ? n1(q2, (b >> 2) & 1) : 1.0);
          v = fma(w, pin[b], v);
        }
        if (TAB) v *= coef[q];
        else v *= wq1(q0) * wq1(q1) * (DIM == 3 ? wq1(q2) : 1.0) * scale * ie_c;
#pragma unroll
        for (int b = 0; b < NP; ++b) {
          const double w = n1(q0, b & 1) * n1(q1, (b >> 1) & 1) * (DIM == 3 ? n1(q2, (b >> 2) & 1) : 1.0);
          pout[b] = fma(w, v, pout[b]);
        }
      }
#pragma unroll
      for (int b = 0; b < NP; ++b) slots[b * CS + t] = pout[b];
    }
    __syncthreads();
    if (is_node) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < cnt) sum += slots[ent[k]];
      const double Ax = con ? x : sum;
      if (S.kind == SM_RESID) {
        M.resid_out[t] = bv - Ax;
      } else {
        double v = fma(S.f2 * dv, bv - Ax, (1.0 + S.f1) * x);
        if (S.use_xold) v = fma(-S.f1, xo, v);
        xo = x;
        x = v;
        X[t] = x;
      }
    }
    __syncthreads();
  }
  if (is_node) M.x_out[t] = x;
}

// the same smoother sequence from the assembled rows (k_mass_ell's arithmetic): the thread's row stays
// in registers, the iterate ping-pongs between two LDS copies, one barrier per step
template <int W>
__global__ void __launch_bounds__(MS_THREADS) k_mass_smoother_ell(const EllMulArgs E, const MassSmArgs M,
                                                                   const SmootherProgram P) {
  __shared__ double X[2][MS_MAX_NODES];
  const int t = threadIdx.x;
  const bool is_node = t < M.n_nodes;
  const size_t st = (size_t)E.stride;
  uint32_t col[W];
  double val[W];
  double x = 0.0, xo = 0.0, bv = 0.0, dv = 0.0;
  bool con = false;
  if (is_node) {
#pragma unroll
    for (int k = 0; k < W; ++k) { col[k] = E.col[k * st + t]; val[k] = E.val[k * st + t]; }
    bv = M.b[t];
    dv = M.dinv[t];
    con = M.cmask ? (M.cmask[t] & 1u) : false;
    if (M.x_in) x = M.x_in[t];
    X[0][t] = x;
  }
  int cur = 0;
  __syncthreads();
  for (int sidx = 0; sidx < P.n; ++sidx) {
    const SmootherStep S = P.s[sidx];
    if (is_node) {
      if (S.kind == SM_POINTWISE) {
        x = S.f2 * dv * bv;
      } else {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) sum = fma(val[k], X[cur][col[k]], sum);
        const double Ax = con ? x : sum;
        if (S.kind == SM_RESID) {
          M.resid_out[t] = bv - Ax;
        } else {
          double v = fma(S.f2 * dv, bv - Ax, (1.0 + S.f1) * x);
          if (S.use_xold) v = fma(-S.f1, xo, v);
          xo = x;
          x = v;
        }
      }
      X[cur ^ 1][t] = x;
    }
    cur ^= 1;
    __syncthreads();
  }
  if (is_node) M.x_out[t] = x;
}

// ---- diagonals -------------------------------------------------------------------------
// velocity: lane per (cell, local node); entry (a, c) = sum_q JxW eta (|grad phi|^2 + (d_c phi)^2
//           [- 2/3 (d_c phi)^2 if compressible])
template <int DIM>
__global__ void __launch_bounds__(256) k_diag_v(const CellArgs A, int comp, int nqv, int gen) {
  constexpr int NV = (DIM == 3) ? 27 : 9;
  constexpr int NQ = NV;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int cell = (int)(t / NV), a = (int)(t - (long)cell * NV);
  if (cell >= A.n_cells) return;
  const size_t cs = (size_t)A.cstride;
  double h[3] = {1, 1, 1}, det = 1.0;
  if (!gen)
    for (int d = 0; d < DIM; ++d) { h[d] = A.hT[d * cs + cell]; det *= h[d]; }
  const int ai[3] = {a % 3, (a / 3) % 3, a / 9};
  double acc[3] = {0, 0, 0};
  for (int q = 0; q < NQ; ++q) {
    const int qi[3] = {q % 3, (q / 3) % 3, q / 9};
    double g[3] = {0, 0, 0}, wq = 1.0;
    for (int e = 0; e < DIM; ++e) {
      double v = 1.0;
      for (int d = 0; d < DIM; ++d) v *= (d == e) ? d2(qi[d], ai[d]) : n2(qi[d], ai[d]);
      g[e] = v / h[e];
      wq *= wq1(qi[e]);
    }
    if (gen) {  // reference gradient -> physical gradient, JxW from the table
      double gp[3] = {0, 0, 0};
      for (int dd = 0; dd < DIM; ++dd)
        for (int e = 0; e < DIM; ++e) gp[dd] += g[e] * A.invJT[((size_t)q * (DIM * DIM) + e * DIM + dd) * cs + cell];
      for (int dd = 0; dd < DIM; ++dd) g[dd] = gp[dd];
      wq = A.jxwT[(size_t)q * cs + cell];
    }
    const double eta = A.etaT[(nqv == 1 ? 0 : (size_t)q * cs) + cell];
    double n2g = 0.0;
    for (int e = 0; e < DIM; ++e) n2g += g[e] * g[e];
    for (int cc = 0; cc < DIM; ++cc) {
      double v = n2g + g[cc] * g[cc];
      if (comp) v -= (2.0 / 3.0) * g[cc] * g[cc];
      acc[cc] += wq * det * eta * v;
    }
  }
  for (int cc = 0; cc < DIM; ++cc) A.slots_u[((size_t)a * cs + cell) * DIM + cc] = acc[cc];
}

template <int DIM>
__global__ void __launch_bounds__(256) k_diag_p(const CellArgs A, int nqv, int gen) {
  constexpr int NP = (DIM == 3) ? 8 : 4;
  constexpr int NQ = (DIM == 3) ? 27 : 9;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int cell = (int)(t / NP), b = (int)(t - (long)cell * NP);
  if (cell >= A.n_cells) return;
  const size_t cs = (size_t)A.cstride;
  double det = 1.0;
  if (!gen)
    for (int d = 0; d < DIM; ++d) det *= A.hT[d * cs + cell];
  double acc = 0.0;
  for (int q = 0; q < NQ; ++q) {
    const int q0 = q % 3, q1 = (q / 3) % 3, q2 = q / 9;
    const double w = n1(q0, b & 1) * n1(q1, (b >> 1) & 1) * (DIM == 3 ? n1(q2, (b >> 2) & 1) : 1.0);
    const double wq = gen ? A.jxwT[(size_t)q * cs + cell] : wq1(q0) * wq1(q1) * (DIM == 3 ? wq1(q2) : 1.0);
    const double eta = A.etaT[(nqv == 1 ? 0 : (size_t)q * cs) + cell];
    acc += wq * det * A.lambda * A.lambda / eta * w * w;
  }
  A.slots_p[(size_t)b * cs + cell] = acc;
}

template <int NC, int NPARTS>
__global__ void __launch_bounds__(256) k_pull(const PullArgs P) { pull_body<NC, NPARTS>(P, blockIdx.x); }

// ---- block scatter: one workgroup = WAVES_PER_BLOCK batches of cells ("block").  Every lane
// parks its 3^DIM contributions in an LDS slot tile; after one barrier the workgroup sums, per
// node of the block and in a fixed order, the slots that hit it (block-local CSR), stores nodes
// no other block touches straight to dst with the vector epilogue fused, and hands the partial
// sums of shared nodes to the fix-up kernel.  No atomics: results are bitwise reproducible.
#if AGMG_ABLATE & 1024
#define AGMG_STAMP(k)                                                                      \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    unsigned long long t_;                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (lane == 0) B.dbg[((size_t)blockIdx.x * WAVES_PER_BLOCK + wave) * 8 + (k)] = t_;    \
  } while (0)
#else
#define AGMG_STAMP(k)
#endif
struct BlkArgs {
  unsigned long long *dbg;
  const uint4 *blk_hdr;      // per block {unique nodes, exclusive nodes, first shared slot, 0}
  const uint32_t *unodes;    // [n_blocks][max_u]: node | cmask << 29
  const uint4 *uent8;        // [n_blocks][max_u]: 8 LDS slot indices (16 bit each) per local node
  const uint16_t *lidT;      // [cstride][LID_ROW]: local id (within its block) of each cell node
  double *slots_sh;
  const uint32_t *blk_list;  // != nullptr: the launch covers blocks blk_list[blk_first + 0 .. gridDim.x)
  int blk_first;
};

// LDS carve-up of k_cells_blk (bytes); max_u = largest number of unique nodes of any block
template <int DIM> struct BlkLds {
  static constexpr int BC = KDims<DIM>::CPW * WAVES_PER_BLOCK;
  static constexpr int WTILE = blk_wave_tile(DIM);                          // doubles per wavefront
  static constexpr int ZERO_SLOT = WAVES_PER_BLOCK * WTILE;                 // slot index of a zero triple
  static constexpr size_t slots = ((size_t)ZERO_SLOT + 4) * sizeof(double);
  static constexpr int ETA_OFF = KDims<DIM>::XB;                             // eta tile behind the exchange tile
  static_assert(KDims<DIM>::XB + KDims<DIM>::NV * KDims<DIM>::CPW <= WTILE && KDims<DIM>::NV * BLK_SLOT_ROW <= WTILE,
                "exchange + eta tiles and the slot tile must fit the wave's tile");
  __host__ __device__ static size_t unodes_off(int max_u) { return slots + (size_t)max_u * 16; }
  __host__ __device__ static size_t total(int max_u) { return unodes_off(max_u); }   // (the node words live in registers)
};

// node-owner sums of the block kernel, specialised on the epilogue (EPI) so the item loop has no
// switch: one thread per node of the block and all DIM components; the 8 slot entries of a node
// are LDS slot indices in doubles (missing ones point at the zero triple)
template <int DIM, int EPI>
__device__ __forceinline__ void blk_node_sums(const char *slots_c, const uint4 *l_ent, const uint32_t *nd /* [MAXU_PT] */,
                                              const int U, const int nex, const EpiArgs &E, double *slots_sh,
                                              const size_t sbase) {
  constexpr int NT = 64 * WAVES_PER_BLOCK;
  constexpr int MAXU_PT = (KDims<DIM>::CPW * WAVES_PER_BLOCK * KDims<DIM>::NV / 2 + NT - 1) / NT;
  // (thread t finishes the nodes t, t + NT, ...: their words are the ones it holds in registers)
  uint4 e4[MAXU_PT];
#pragma unroll
  for (int i = 0; i < MAXU_PT; ++i) e4[i] = l_ent[min((int)threadIdx.x + i * NT, U - 1)];
  double sum[MAXU_PT][DIM];
#pragma unroll
  for (int i = 0; i < MAXU_PT; ++i) {
    const uint32_t ew[4] = {e4[i].x, e4[i].y, e4[i].z, e4[i].w};
#pragma unroll
    for (int cc = 0; cc < DIM; ++cc) sum[i][cc] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t off = (k & 1) ? (ew[k >> 1] >> 16) : (ew[k >> 1] & 0xFFFFu);
      const double *p = reinterpret_cast<const double *>(slots_c) + off;
#pragma unroll
      for (int cc = 0; cc < DIM; ++cc) sum[i][cc] += p[cc];
    }
  }
#pragma unroll
  for (int i = 0; i < MAXU_PT; ++i) {
    const int lid = threadIdx.x + i * NT;
    if (lid < U) {
      if (lid < nex) {
        const size_t g0 = (size_t)(nd[i] & NODE_MASK) * DIM;
        double out[DIM];
#pragma unroll
        for (int cc = 0; cc < DIM; ++cc) {
          const bool con = (nd[i] >> (29 + cc)) & 1u;
          const double sm = sum[i][cc];
          const size_t t = g0 + cc;
          if (EPI == EPI_STORE) out[cc] = con ? (E.x ? E.x[t] : 0.0) : sm;
          else if (EPI == EPI_RESID) out[cc] = E.b[t] - (con ? (E.x ? E.x[t] : 0.0) : sm);
          else {  // EPI_CHEB
            const double x = E.x[t];
            double v = fma(E.f2 * E.dinv[t], E.b[t] - (con ? x : sm), (1.0 + E.f1) * x);
            if (E.xold) v = fma(-E.f1, E.xold[t], v);
            out[cc] = v;
          }
        }
#pragma unroll
        for (int cc = 0; cc < DIM; ++cc) E.dst[g0 + cc] = out[cc];
      } else {
#pragma unroll
        for (int cc = 0; cc < DIM; ++cc) slots_sh[sbase + (size_t)(lid - nex) * DIM + cc] = sum[i][cc];
      }
    }
  }
}

template <int DIM, int OP, bool COMP, bool PERQ, bool GEN>
__device__ __forceinline__ void cells_blk_body(const CellArgs &A, const BlkArgs &B, const EpiArgs &E, const int max_u, const int blk) {
  constexpr int NV = KDims<DIM>::NV, NP = KDims<DIM>::NP, CPW = KDims<DIM>::CPW, LROW = KDims<DIM>::LROW;
  constexpr int BC = CPW * WAVES_PER_BLOCK, NT = 64 * WAVES_PER_BLOCK;
  constexpr int MAXU_PT = (BC * NV / 2 + NT - 1) / NT;  // nodes per thread (U <= BC*NV/2, checked on the host)
  constexpr int NETA = NV * CPW, NETA_PL = (NETA + 63) / 64;
  constexpr bool HAS_U = (OP != KOP_BT);
  constexpr bool ETA_LDS = PERQ && HAS_U;
  extern __shared__ double smem[];
  char *const smem_c = reinterpret_cast<char *>(smem);
  double *const slots = smem;                                      // [wave][WTILE] + zero triple
  uint4 *const l_ent = reinterpret_cast<uint4 *>(smem_c + BlkLds<DIM>::slots);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Life of the slot area: (1) copy of the source vector at the block's nodes, gathered from by
  // all waves; (2) per wave: gradient-exchange tile + viscosity tile; (3) per wave: slot tile
  // with the lane results.  Barriers separate (1)|(2) and (3)|node sums; (2)|(3) is wave-local.
  double *const wtile = slots + (size_t)wave * BlkLds<DIM>::WTILE;
  double (*xb)[LROW] = reinterpret_cast<double (*)[LROW]>(wtile);
  double *const eta_tile = wtile + BlkLds<DIM>::ETA_OFF;
  int j = lane / DIM;
  const int c = lane - j * DIM;
  const bool lane_ok = j < CPW;
  if (!lane_ok) j = CPW - 1;
  // XCD affinity: workgroups with the same blockIdx % 8 share an XCD (and its L2), so each of the
  // 8 groups takes a contiguous run of blocks, i.e. a compact part of the mesh whose blocks re-read
  // each other's halo nodes (a speed choice only: any placement is correct)
  const int c0 = blk * BC;
  const int cell_l = c0 + wave * CPW + j;
  const int cell = cell_l < A.n_cells ? cell_l : A.n_cells - 1;
  const size_t cs = (size_t)A.cstride;
  AGMG_STAMP(0);
  // ---- all global reads of the block are issued here, coalesced.  First the node words of the block's
  // nodes (registers; thread t holds entries t, t + NT, ..., which the host pads with node 0 up to max_u):
  // the gather of the source vector depends on them, and vector loads return in the order issued, so
  // it can start as soon as they are back, while the rest of this batch is still in flight.  The slot
  // entries of the node sums (fixed stride max_u per block: no dependent address) go straight to LDS
  // (LDS-DMA, no registers): wave w moves entries [i*NT + 64w, +64)
  const size_t mbase = (size_t)blk * max_u;
  uint32_t m_node[MAXU_PT];
#pragma unroll
  for (int i = 0; i < MAXU_PT; ++i) {
    const int t = threadIdx.x + i * NT;
    m_node[i] = t < max_u ? B.unodes[mbase + t] : 0u;
  }
#pragma unroll
  for (int i = 0; i < MAXU_PT; ++i) {
    const int t = threadIdx.x + i * NT;
    if (t < max_u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(B.uent8 + mbase + t),
                                       (__attribute__((address_space(3))) void *)(l_ent + i * NT + wave * 64), 16, 0, 0);
  }
  const uint4 hdr = B.blk_hdr[blk];
  const int U = (int)hdr.x, nex = (int)hdr.y;
  uint32_t lids[HAS_U ? NV : 1];
  if (HAS_U) {
#pragma unroll
    for (int a = 0; a < NV; ++a) lids[a] = 0;
    {
      const uint4 *row = reinterpret_cast<const uint4 *>(B.lidT + (size_t)cell * LID_ROW);
      uint32_t w[LID_ROW / 2];
#pragma unroll
      for (int k = 0; k < (NV + 7) / 8; ++k) {
        const uint4 v = row[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int a = 0; a < NV; ++a) lids[a] = (a & 1) ? (w[a >> 1] >> 16) : (w[a >> 1] & 0xFFFFu);
    }
  }
  double m_eta[ETA_LDS ? NETA_PL : 1];
  if (ETA_LDS) {
#pragma unroll
    for (int k = 0; k < NETA_PL; ++k) {
      const int idx = lane + 64 * k;
      const int q = idx / CPW, jj = idx - q * CPW;
      const int cx = min(c0 + wave * CPW + jj, A.cstride - 1);
      m_eta[k] = A.etaT[(size_t)min(q, NV - 1) * cs + cx];
    }
  }
  if (threadIdx.x < 4) slots[BlkLds<DIM>::ZERO_SLOT + threadIdx.x] = 0.0;
  double u[NV];
  if (HAS_U) {
    // source vector at the block's nodes -> LDS (one thread per node, constrained entries zeroed)
    double sv[MAXU_PT][DIM];
#pragma unroll
    for (int i = 0; i < MAXU_PT; ++i) {   // all loads first
      const double *sp = A.src_u + (size_t)(m_node[i] & NODE_MASK) * DIM;
#pragma unroll
      for (int cc = 0; cc < DIM; ++cc) sv[i][cc] = sp[cc];
    }
#pragma unroll
    for (int i = 0; i < MAXU_PT; ++i) {
      const int t = threadIdx.x + i * NT;
      if (t < U) {
#pragma unroll
        for (int cc = 0; cc < DIM; ++cc) slots[t * DIM + cc] = ((m_node[i] >> (29 + cc)) & 1u) ? 0.0 : sv[i][cc];
      }
    }
    AGMG_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA of the slot entries: not tracked by the compiler)
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NV; ++a) u[a] = slots[lids[a] * DIM + c];
    __syncthreads();
    AGMG_STAMP(7);
  }
  if (ETA_LDS) {
#pragma unroll
    for (int k = 0; k < NETA_PL; ++k) {
      const int idx = lane + 64 * k;
      if (idx < NETA) eta_tile[idx] = m_eta[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  {
    double G[NV], pout[NP];
    AGMG_STAMP(1);
    WaveXch<DIM> ex{xb, lane, j};
#if AGMG_ABLATE & 64
    // (diagnostic: no cell arithmetic)
#pragma unroll
    for (int a = 0; a < NV; ++a) G[a] = HAS_U ? u[a] : 1.0;
#pragma unroll
    for (int b = 0; b < NP; ++b) pout[b] = 0.0;
    (void)ex;
#else
    cell_compute<DIM, OP, COMP, PERQ, ETA_LDS, false, GEN>(A, cell, c, j, lane, ex, G, pout, eta_tile, u);
#endif
    AGMG_STAMP(2);
    double *my = wtile + lane;
#pragma unroll
    for (int a = 0; a < NV; ++a) my[a * BLK_SLOT_ROW] = G[a];
    if (OP == KOP_STOKES) sum_over_components<DIM, NP>(pout);
    if (OP == KOP_STOKES && c == 0 && lane_ok && cell_l < A.n_cells) {
#pragma unroll
      for (int bb = 0; bb < NP; ++bb) A.slots_p[(size_t)bb * cs + cell] = pout[bb];
    }
  }
  AGMG_STAMP(3);
  if (!HAS_U) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA of the slot entries, see above)
  __syncthreads();
  AGMG_STAMP(4);
#if AGMG_ABLATE & 128
  if (slots[threadIdx.x] != 12345.678) return;
#endif
  const size_t sbase = (size_t)hdr.z * DIM;
  if (E.epi == EPI_CHEB) blk_node_sums<DIM, EPI_CHEB>(smem_c, l_ent, m_node, U, nex, E, B.slots_sh, sbase);
  else if (E.epi == EPI_RESID) blk_node_sums<DIM, EPI_RESID>(smem_c, l_ent, m_node, U, nex, E, B.slots_sh, sbase);
  else blk_node_sums<DIM, EPI_STORE>(smem_c, l_ent, m_node, U, nex, E, B.slots_sh, sbase);
  AGMG_STAMP(5);
}

template <int DIM, int OP, bool COMP, bool PERQ, bool GEN>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, AGMG_CELLS_WAVES)
k_cells_blk(const CellArgs A, const BlkArgs B, const EpiArgs E, const int max_u) {
#if AGMG_ABLATE & 256
  const int blk = blockIdx.x;
#else
  const int item = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  const int blk = B.blk_list ? (int)B.blk_list[B.blk_first + item] : item;
#endif
  cells_blk_body<DIM, OP, COMP, PERQ, GEN>(A, B, E, max_u, blk);
}

// A level that is ONE block (<= 84 cells in 3-D: no node is shared with another workgroup, no fix-up pass): a whole
// Chebyshev sequence of the A block, and the residual behind it, in one launch.  The steps are the launches of
// cheb_vmult / cheb_step (same kernels' body, same epilogues, same vectors): between two of them the workgroup makes its
// stores visible (agent-scope fence: the next step gathers what other waves wrote) and meets at a barrier.  On these
// levels a launch is the latency chain of one workgroup (about 10 us) plus the gap to the next launch: eight to nine
// of them per smoother application, here one.
template <int DIM, bool COMP, bool PERQ, bool GEN>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, AGMG_CELLS_WAVES)
k_cells_blk_seq(const CellArgs A0, const BlkArgs B, const SeqArgs S, const int max_u) {
  constexpr int NT = 64 * WAVES_PER_BLOCK;
  for (int k = 0; k < S.n; ++k) {
    const SeqStep st = S.s[k];
    if (st.kind == SEQ_POINTWISE) {
      for (int t = threadIdx.x; t < S.n_dofs; t += NT) st.dst[t] = st.f2 * S.dinv[t] * S.b[t];
    } else {
      CellArgs A = A0;
      A.src_u = st.x;
      EpiArgs E;
      E.epi = st.kind == SEQ_CHEB ? EPI_CHEB : EPI_RESID;
      E.dst = st.dst; E.x = st.x; E.b = S.b; E.xold = st.xold; E.dinv = S.dinv; E.f1 = st.f1; E.f2 = st.f2;
      cells_blk_body<DIM, KOP_A, COMP, PERQ, GEN>(A, B, E, max_u, 0);
    }
    __threadfence();
    __syncthreads();
  }
}

struct FixArgs {
  const uint32_t *sh_nodes, *sh_ptr, *sh_list;
  const double *slots_sh;
  int i0, n_range;          // shared nodes [i0, i0 + n_range)
  uint32_t recv_base;       // sh_list entries >= recv_base address recv[(entry - recv_base) * NC + c]
  const double *recv;
  double *ghost_out;        // != nullptr: raw sums to ghost_out[(node - n_owned) * NC + c]
  int n_owned;
  EpiArgs e;
};
// One thread per shared node.  The loads are arranged in three dependent rounds instead of one
// per CSR entry: (row bounds, node word) -> (the first FIX_PRE list entries, clamped) + the
// epilogue's operands (they only need the node) -> the partial sums; rows with more entries finish
// in a loop.  The order of the additions is the CSR order as before: bitwise the same sums.
template <int NC>
__device__ __forceinline__ void fix_body(const FixArgs &F, const unsigned block) {
  constexpr int FIX_PRE = 4;
  const int il = (int)block * 256 + threadIdx.x;
  if (il >= F.n_range) return;
  const int i = F.i0 + il;
  const uint32_t s = F.sh_ptr[i], en = F.sh_ptr[i + 1];
  const uint32_t nd = F.sh_nodes[i];   // node | cmask << 29
  const size_t g0 = (size_t)(nd & NODE_MASK) * NC;
  uint32_t ent[FIX_PRE];
#pragma unroll
  for (int k = 0; k < FIX_PRE; ++k) ent[k] = F.sh_list[min(s + k, en - 1)];
  // epilogue operands of this node (velocity-row epilogues: STORE / RESID / CHEB; others load later)
  const EpiArgs &E = F.e;
  const bool pre = !F.ghost_out && (E.epi == EPI_STORE || E.epi == EPI_RESID || E.epi == EPI_CHEB);
  double xv[NC], bv[NC], dv[NC], ov[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    xv[c] = bv[c] = dv[c] = ov[c] = 0.0;
    if (pre) {
      const bool con = (nd >> (29 + c)) & 1u;
      if (E.epi == EPI_CHEB) {
        xv[c] = E.x[g0 + c]; bv[c] = E.b[g0 + c]; dv[c] = E.dinv[g0 + c];
        if (E.xold) ov[c] = E.xold[g0 + c];
      } else {
        if (con && E.x) xv[c] = E.x[g0 + c];
        if (E.epi == EPI_RESID) bv[c] = E.b[g0 + c];
      }
    }
  }
  double part[FIX_PRE][NC];
#pragma unroll
  for (int k = 0; k < FIX_PRE; ++k) {
    const double *p = ent[k] >= F.recv_base ? F.recv + (size_t)(ent[k] - F.recv_base) * NC : F.slots_sh + (size_t)ent[k] * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) part[k][c] = p[c];
  }
  double sum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) sum[c] = 0.0;
#pragma unroll
  for (int k = 0; k < FIX_PRE; ++k)
    if (s + k < en) {
#pragma unroll
      for (int c = 0; c < NC; ++c) sum[c] += part[k][c];
    }
  for (uint32_t k = s + FIX_PRE; k < en; ++k) {
    const uint32_t e = F.sh_list[k];
    const double *p = e >= F.recv_base ? F.recv + (size_t)(e - F.recv_base) * NC : F.slots_sh + (size_t)e * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) sum[c] += p[c];
  }
  if (F.ghost_out) {
    double *o = F.ghost_out + (size_t)((int)(nd & NODE_MASK) - F.n_owned) * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = sum[c];
    return;
  }
  if (pre) {
    double out[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool con = (nd >> (29 + c)) & 1u;
      const double Ax = con ? xv[c] : sum[c];
      if (E.epi == EPI_STORE) out[c] = Ax;
      else if (E.epi == EPI_RESID) out[c] = bv[c] - Ax;
      else {
        double v = fma(E.f2 * dv[c], bv[c] - Ax, (1.0 + E.f1) * xv[c]);
        if (E.xold) v = fma(-E.f1, ov[c], v);
        out[c] = v;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) E.dst[g0 + c] = out[c];
    return;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool con = (nd >> (29 + c)) & 1u;
    double out;
    if (epi_value(F.e, g0 + c, sum[c], con, out)) F.e.dst[g0 + c] = out;
  }
}
template <int NC>
__global__ void __launch_bounds__(256) k_fix(const FixArgs F) { fix_body<NC>(F, blockIdx.x); }
// ---- one message round per application (SymExchange of the velocity nodes)
struct FixSymArgs {
  const uint32_t *sh_nodes, *sh_ptr, *sh_list;   // block tables' shared nodes
  const double *slots_sh;
  uint32_t recv_base;                            // list entries >= recv_base belong to the two-round form: skipped
  int n_rows;                                    // part 0: shared rows of the exchange; part 1: shared nodes of the block tables
  const uint32_t *shidx;                         // part 0: row -> index among the block tables' shared nodes
  const int32_t *symrow;                         // part 1: the inverse, or -1
  const uint32_t *sptr, *spos;
  const uint8_t *ssplit;
  double *ypart, *sendbuf;
  const double *recvbuf;
  EpiArgs e;
};
template <int NC>
__device__ __forceinline__ void local_slot_sum(const FixSymArgs &F, const int i, double (&sum)[NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) sum[c] = 0.0;
  for (uint32_t k = F.sh_ptr[i]; k < F.sh_ptr[i + 1]; ++k) {
    const uint32_t e = F.sh_list[k];
    if (e >= F.recv_base) continue;
    const double *p = F.slots_sh + (size_t)e * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) sum[c] += p[c];
  }
}
template <int NC>
__global__ void __launch_bounds__(256) k_fix_sym_a(const FixSymArgs F) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= F.n_rows) return;
  double sum[NC];
  local_slot_sum<NC>(F, (int)F.shidx[r], sum);
#pragma unroll
  for (int c = 0; c < NC; ++c) F.ypart[(size_t)r * NC + c] = sum[c];
  for (uint32_t k = F.sptr[r]; k < F.sptr[r + 1]; ++k) {
    double *o = F.sendbuf + (size_t)F.spos[k] * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = sum[c];
  }
}
template <int NC>
__global__ void __launch_bounds__(256) k_fix_sym_b(const FixSymArgs F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.n_rows) return;
  const uint32_t nd = F.sh_nodes[i];
  const size_t g0 = (size_t)(nd & NODE_MASK) * NC;
  double sum[NC];
  const int r = F.symrow[i];
  if (r < 0) local_slot_sum<NC>(F, i, sum);
  else {
    const uint32_t s0 = F.sptr[r], s1 = F.sptr[r + 1], mine = s0 + F.ssplit[r];
#pragma unroll
    for (int c = 0; c < NC; ++c) sum[c] = 0.0;
    for (uint32_t k = s0; k < s1; ++k) {
      if (k == mine) {
#pragma unroll
        for (int c = 0; c < NC; ++c) sum[c] += F.ypart[(size_t)r * NC + c];
      }
      const double *p = F.recvbuf + (size_t)F.spos[k] * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c) sum[c] += p[c];
    }
    if (mine == s1) {
#pragma unroll
      for (int c = 0; c < NC; ++c) sum[c] += F.ypart[(size_t)r * NC + c];
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool con = (nd >> (29 + c)) & 1u;
    double out;
    if (epi_value(F.e, g0 + c, sum[c], con, out)) F.e.dst[g0 + c] = out;
  }
}
// Stokes: the fix-up of the velocity rows and the pull of the pressure rows are independent of
// each other (both only wait for the cell kernel): one launch, the grid split between them
template <int DIM>
__global__ void __launch_bounds__(256) k_fix_pull(const FixArgs F, const PullArgs P, const unsigned n_fix_blocks) {
  if (blockIdx.x < n_fix_blocks) fix_body<DIM>(F, blockIdx.x);
  else pull_body<1, 1>(P, blockIdx.x - n_fix_blocks);
}

// StokesOperator::local_apply_boundary_face (matrix_free_operators.cc:294-333), node-owner form: one
// thread per surface node sums, face by face in table order, phi_a(q) * (-(g_q . u_q)) * n_q JxW_q
// over the face q-points.  u_q is recomputed by every node of the face (9 x on a 2-D manifold:
// negligible next to the cell loop) so that no scatter, no atomics and no second pass are needed.
template <int DIM>
__global__ void __launch_bounds__(128) k_face_term(const FaceTerm F, const uint8_t *__restrict__ cmask,
                                                   const double *__restrict__ src, double *__restrict__ dst,
                                                   const double sign, double *__restrict__ ghost_out, const int n_owned,
                                                   const int plain) {
  constexpr int NF = DIM == 3 ? 9 : 3, NQF = NF;
  const int s = blockIdx.x * 128 + threadIdx.x;
  if (s >= F.n_nodes) return;
  const double sq = 0.7745966692414834;  // sqrt(3/5)
  double N2[3][3];                       // [q][i] Q2 values at the Gauss points of the unit interval
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double x = 0.5 + 0.5 * sq * (q - 1);
    N2[q][0] = (1 - x) * (1 - 2 * x); N2[q][1] = 4 * x * (1 - x); N2[q][2] = x * (2 * x - 1);
  }
  const uint32_t node = F.nodes[s];
  double val[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) val[c] = 0.0;
  for (uint32_t k = F.ptr[s]; k < F.ptr[s + 1]; ++k) {
    const uint32_t e = F.ent[k], f = e >> 4, a = e & 15u;
    double u[NF][DIM];
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      const uint32_t nb = F.fnode[(size_t)f * NF + b];
      const uint8_t m = cmask[nb];
#pragma unroll
      for (int c = 0; c < DIM; ++c) u[b][c] = (((m >> c) & 1) && !plain) ? 0.0 : src[(size_t)nb * DIM + c];
    }
#pragma unroll
    for (int q = 0; q < NQF; ++q) {
      const double *g = F.stab + ((size_t)f * NQF + q) * DIM;
      const double *nj = F.njxw + ((size_t)f * NQF + q) * DIM;
      double gu = 0.0, pa = 0.0;
#pragma unroll
      for (int b = 0; b < NF; ++b) {
        const double pb = DIM == 3 ? N2[q % 3][b % 3] * N2[q / 3][b / 3] : N2[q][b];
        if ((uint32_t)b == a) pa = pb;
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) t = fma(g[c], u[b][c], t);
        gu = fma(pb, t, gu);
      }
#pragma unroll
      for (int c = 0; c < DIM; ++c) val[c] = fma(-pa * gu, nj[c], val[c]);
    }
  }
  if (ghost_out && (int)node >= n_owned) {   // another rank owns the row: its share of the sum travels
#pragma unroll
    for (int c = 0; c < DIM; ++c) ghost_out[(size_t)((int)node - n_owned) * DIM + c] = val[c];
    return;
  }
  const uint8_t m = cmask[node];
#pragma unroll
  for (int c = 0; c < DIM; ++c)
    if (!((m >> c) & 1)) dst[(size_t)node * DIM + c] += sign * val[c];
}
template <int DIM>
__global__ void __launch_bounds__(128) k_face_add_received(const FaceTerm F, const uint8_t *__restrict__ cmask,
                                                           const double *__restrict__ recv, double *__restrict__ dst,
                                                           const double sign) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= F.n_exp_nodes) return;
  const uint32_t node = F.exp_nodes[i];
  double sum[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) sum[c] = 0.0;
  for (uint32_t k = F.exp_ptr[i]; k < F.exp_ptr[i + 1]; ++k) {
    const double *p = recv + (size_t)F.exp_ent[k] * DIM;
#pragma unroll
    for (int c = 0; c < DIM; ++c) sum[c] += p[c];
  }
  const uint8_t m = cmask[node];
#pragma unroll
  for (int c = 0; c < DIM; ++c)
    if (!((m >> c) & 1)) dst[(size_t)node * DIM + c] += sign * sum[c];
}

template <int DIM, int OP, bool GEN>
void launch_cells_t(hipStream_t s, const CellArgs &a, bool comp, bool perq) {
  const int cpb = (64 / DIM) * WAVES_PER_BLOCK;
  const dim3 grid((a.n_cells + cpb - 1) / cpb), block(64 * WAVES_PER_BLOCK);
  if (comp) {
    if (perq) hipLaunchKernelGGL((k_cells<DIM, OP, true, true, GEN>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_cells<DIM, OP, true, false, GEN>), grid, block, 0, s, a);
  } else {
    if (perq) hipLaunchKernelGGL((k_cells<DIM, OP, false, true, GEN>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_cells<DIM, OP, false, false, GEN>), grid, block, 0, s, a);
  }
}
template <int DIM, bool GEN>
void launch_cells_newton(hipStream_t s, const CellArgs &a, bool comp, bool perq) {
  const int cpb = (64 / DIM) * WAVES_PER_BLOCK;
  const dim3 grid((a.n_cells + cpb - 1) / cpb), block(64 * WAVES_PER_BLOCK);
  if (comp) {
    if (perq) hipLaunchKernelGGL((k_cells<DIM, KOP_STOKES, true, true, GEN, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_cells<DIM, KOP_STOKES, true, false, GEN, true>), grid, block, 0, s, a);
  } else {
    if (perq) hipLaunchKernelGGL((k_cells<DIM, KOP_STOKES, false, true, GEN, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_cells<DIM, KOP_STOKES, false, false, GEN, true>), grid, block, 0, s, a);
  }
}
template <int DIM, bool GEN>
void launch_cells_op(hipStream_t s, const CellArgs &a, int which, bool comp, bool perq, bool newton) {
  if (newton && which == AGMG_OP_STOKES) { launch_cells_newton<DIM, GEN>(s, a, comp, perq); return; }
  if (which == AGMG_OP_A) launch_cells_t<DIM, KOP_A, GEN>(s, a, comp, perq);
  else if (which == AGMG_OP_STOKES) launch_cells_t<DIM, KOP_STOKES, GEN>(s, a, comp, perq);
  else launch_cells_t<DIM, KOP_BT, GEN>(s, a, false, false);
}

template <int DIM, int OP, bool GEN>
int launch_blk_t(hipStream_t s, const CellArgs &a, const BlkArgs &b, const EpiArgs &e, int n_blocks, int max_u,
                 bool comp, bool perq) {
  const size_t lds = BlkLds<DIM>::total(max_u);
  const dim3 grid(n_blocks), block(64 * WAVES_PER_BLOCK);
#define AGMG_BLK_LAUNCH(C, Q)                                                                              \
  do {                                                                                                     \
    static size_t lds_set[64] = {0};   /* the attribute is per device */                                   \
    int dev_ = 0;                                                                                          \
    AGMG_HIP_CHECK(hipGetDevice(&dev_));                                                                   \
    if (dev_ < 0 || dev_ >= 64) { set_error("device ordinal out of range"); return -1; }                   \
    if (lds > lds_set[dev_]) {                                                                             \
      AGMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells_blk<DIM, OP, C, Q, GEN>),      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
      lds_set[dev_] = lds;                                                                                 \
    }                                                                                                      \
    hipLaunchKernelGGL((k_cells_blk<DIM, OP, C, Q, GEN>), grid, block, lds, s, a, b, e, max_u);                        \
  } while (0)
  if (comp) { if (perq) AGMG_BLK_LAUNCH(true, true); else AGMG_BLK_LAUNCH(true, false); }
  else { if (perq) AGMG_BLK_LAUNCH(false, true); else AGMG_BLK_LAUNCH(false, false); }
#undef AGMG_BLK_LAUNCH
  return 0;
}
template <int DIM, bool GEN>
int launch_blk_seq_t(hipStream_t s, const CellArgs &a, const BlkArgs &b, const SeqArgs &q, int max_u, bool comp, bool perq) {
  const size_t lds = BlkLds<DIM>::total(max_u);
  const dim3 grid(1), block(64 * WAVES_PER_BLOCK);
#define AGMG_SEQ_LAUNCH(C, Q)                                                                              \
  do {                                                                                                     \
    static size_t lds_set[64] = {0};   /* the attribute is per device */                                   \
    int dev_ = 0;                                                                                          \
    AGMG_HIP_CHECK(hipGetDevice(&dev_));                                                                   \
    if (dev_ < 0 || dev_ >= 64) { set_error("device ordinal out of range"); return -1; }                   \
    if (lds > lds_set[dev_]) {                                                                             \
      AGMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells_blk_seq<DIM, C, Q, GEN>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
      lds_set[dev_] = lds;                                                                                 \
    }                                                                                                      \
    hipLaunchKernelGGL((k_cells_blk_seq<DIM, C, Q, GEN>), grid, block, lds, s, a, b, q, max_u);            \
  } while (0)
  if (comp) { if (perq) AGMG_SEQ_LAUNCH(true, true); else AGMG_SEQ_LAUNCH(true, false); }
  else { if (perq) AGMG_SEQ_LAUNCH(false, true); else AGMG_SEQ_LAUNCH(false, false); }
#undef AGMG_SEQ_LAUNCH
  return 0;
}
template <int DIM, bool GEN>
int launch_blk_op(hipStream_t s, const CellArgs &a, const BlkArgs &b, const EpiArgs &e, int n_blocks, int max_u,
                  int which, bool comp, bool perq) {
  if (which == AGMG_OP_A) return launch_blk_t<DIM, KOP_A, GEN>(s, a, b, e, n_blocks, max_u, comp, perq);
  if (which == AGMG_OP_STOKES) return launch_blk_t<DIM, KOP_STOKES, GEN>(s, a, b, e, n_blocks, max_u, comp, perq);
  return launch_blk_t<DIM, KOP_BT, GEN>(s, a, b, e, n_blocks, max_u, false, false);
}

}  // namespace

int launch_face_term(hipStream_t s, int dim, const FaceTerm &ft, const uint8_t *cmask_v, const double *src_u,
                     double *dst_u, double sign, double *ghost_out, int n_owned, bool plain) {
  if (ft.n_nodes <= 0) return 0;
  const dim3 grid((ft.n_nodes + 127) / 128), block(128);
  if (dim == 3) hipLaunchKernelGGL((k_face_term<3>), grid, block, 0, s, ft, cmask_v, src_u, dst_u, sign, ghost_out, n_owned, plain ? 1 : 0);
  else hipLaunchKernelGGL((k_face_term<2>), grid, block, 0, s, ft, cmask_v, src_u, dst_u, sign, ghost_out, n_owned, plain ? 1 : 0);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_face_add_received(hipStream_t s, int dim, const FaceTerm &ft, const uint8_t *cmask_v, const double *recv,
                             double *dst_u, double sign) {
  if (ft.n_exp_nodes <= 0) return 0;
  const dim3 grid((ft.n_exp_nodes + 127) / 128), block(128);
  if (dim == 3) hipLaunchKernelGGL((k_face_add_received<3>), grid, block, 0, s, ft, cmask_v, recv, dst_u, sign);
  else hipLaunchKernelGGL((k_face_add_received<2>), grid, block, 0, s, ft, cmask_v, recv, dst_u, sign);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}


int launch_op_cells(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
                    const double *src_p, double *slots_u, double *slots_p, bool plain) {
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (L.n_cells <= 0) return 0;
  CellArgs a = make_cell_args(L, cd, src_u, src_p, slots_u, slots_p);
  a.plain = plain ? 1 : 0;
  const bool perq = cd.nqv > 1;
  // the compressible term only exists where the velocity is read (A, Stokes)
  const bool comp = (cd.is_compressible != 0 || cd.dilation) && which != AGMG_OP_BT;
  const bool gen = L.geom == AGMG_GEOM_GENERAL;
  if (L.dim == 3) {
    if (gen) launch_cells_op<3, true>(s, a, which, comp, perq, cd.newton || cd.dilation);
    else launch_cells_op<3, false>(s, a, which, comp, perq, cd.newton || cd.dilation);
  } else {
    if (gen) launch_cells_op<2, true>(s, a, which, comp, perq, cd.newton || cd.dilation);
    else launch_cells_op<2, false>(s, a, which, comp, perq, cd.newton || cd.dilation);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_mass_cells(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const double *src_p, double *slots_p) {
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (L.n_cells <= 0) return 0;
  const CellArgs a = make_cell_args(L, cd, nullptr, src_p, nullptr, slots_p);
  const dim3 grid((L.n_cells + 255) / 256), block(256);
  const bool perq = cd.nqv > 1;
  const bool gen = L.geom == AGMG_GEOM_GENERAL;
#define AGMG_MASS_LAUNCH(D, G)                                                        \
  do {                                                                                \
    if (perq) hipLaunchKernelGGL((k_mass<D, true, G>), grid, block, 0, s, a);         \
    else hipLaunchKernelGGL((k_mass<D, false, G>), grid, block, 0, s, a);             \
  } while (0)
  if (L.dim == 3) { if (gen) AGMG_MASS_LAUNCH(3, true); else AGMG_MASS_LAUNCH(3, false); }
  else { if (gen) AGMG_MASS_LAUNCH(2, true); else AGMG_MASS_LAUNCH(2, false); }
#undef AGMG_MASS_LAUNCH
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_mass_ell_assemble(hipStream_t s, const DeviceLevel &L, const CoeffData &cd) {
  if (!cd.set || !L.ell_col || !cd.ell_val) { set_error("mass rows: tables missing"); return -4; }
  if (L.ell_stride > L.n_nodes_p)   // padding rows
    AGMG_HIP_CHECK(hipMemsetAsync(cd.ell_val, 0, (size_t)L.ell_w * L.ell_stride * sizeof(double), s));
  const CellArgs a = make_cell_args(L, cd, nullptr, nullptr, nullptr, nullptr);
  const EllArgs m{L.ell_col, cd.ell_val, L.n_nodes_p, L.ell_stride, L.pull_ptr_p, L.pull_list_p};
  const dim3 grid((L.n_nodes_p + 255) / 256), block(256);
  const bool perq = cd.nqv > 1;
  const bool gen = L.geom == AGMG_GEOM_GENERAL;
#define AGMG_ELL_LAUNCH(D, G)                                                                      \
  do {                                                                                             \
    if (perq) hipLaunchKernelGGL((k_mass_ell_assemble<D, true, G>), grid, block, 0, s, a, m);      \
    else hipLaunchKernelGGL((k_mass_ell_assemble<D, false, G>), grid, block, 0, s, a, m);          \
  } while (0)
  if (L.dim == 3) { if (gen) AGMG_ELL_LAUNCH(3, true); else AGMG_ELL_LAUNCH(3, false); }
  else { if (gen) AGMG_ELL_LAUNCH(2, true); else AGMG_ELL_LAUNCH(2, false); }
#undef AGMG_ELL_LAUNCH
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_mass_ell(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const double *src_p, const EpiArgs &e, int phase) {
  if (!L.ell_col || !cd.ell_val || cd.ell_stride != L.ell_stride) { set_error("mass rows not assembled"); return -4; }
  EllMulArgs m{L.ell_col, cd.ell_val, L.n_nodes_p, L.ell_stride, L.cmask_p, src_p, e, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, L.ell_cls, L.ell_cls_off};
  if (phase == PH_GHOST) {
    m.node0 = L.n_owned_p; m.n_nodes = L.n_nodes_p - L.n_owned_p; m.ghost_out = L.xp.ghostbuf;
    if (m.n_nodes > 0 && !m.ghost_out) { set_error("launch_mass_ell: ghost rows without an exchange buffer"); return -1; }
  } else if (phase == PH_OWNED) {
    m.n_nodes = L.n_owned_p;
    if (L.xp.n_export > 0) {
      if (!L.ell_xrow) { set_error("launch_mass_ell: level has no table of exported rows"); return -1; }
      m.xrow = L.ell_xrow; m.xptr = L.ell_xptr; m.xent = L.ell_xent; m.recv = L.xp.expbuf;
    }
  } else if (phase != PH_ALL) { set_error("launch_mass_ell: bad phase"); return -1; }
  if (m.n_nodes <= 0) return 0;
  const dim3 grid((m.n_nodes + 255) / 256), block(256);
  if (L.dim == 3) hipLaunchKernelGGL((k_mass_ell<27>), grid, block, 0, s, m);
  else hipLaunchKernelGGL((k_mass_ell<9>), grid, block, 0, s, m);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_mass_ell_sym(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const double *src_p, const EpiArgs &e, int part) {
  if (!L.ell_col || !cd.ell_val || cd.ell_stride != L.ell_stride) { set_error("mass rows not assembled"); return -4; }
  const SymExchange &x = L.sp;
  if (!x.active() || !x.row) { set_error("launch_mass_ell_sym: level has no symmetric exchange plan"); return -1; }
  if (part == 0) {
    if (x.n_sh <= 0) return 0;
    const EllSymArgs a{L.ell_col, cd.ell_val, x.n_sh, L.ell_stride, src_p, x.sh_nodes, x.ptr, x.pos, x.ypart, x.sendbuf};
    const dim3 grid((x.n_sh + 255) / 256), block(256);
    if (L.dim == 3) hipLaunchKernelGGL((k_mass_ell_sym_a<27>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_mass_ell_sym_a<9>), grid, block, 0, s, a);
  } else {
    if (L.n_nodes_p <= 0) return 0;
    const EllMulArgs m{L.ell_col, cd.ell_val, L.n_nodes_p, L.ell_stride, L.cmask_p, src_p, e, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                       x.row, x.ptr, x.pos, x.split, x.ypart, x.recvbuf, L.ell_cls, L.ell_cls_off};
    const dim3 grid((L.n_nodes_p + 255) / 256), block(256);
    if (L.dim == 3) hipLaunchKernelGGL((k_mass_ell<27>), grid, block, 0, s, m);
    else hipLaunchKernelGGL((k_mass_ell<9>), grid, block, 0, s, m);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_fix_sym(hipStream_t s, const DeviceLevel &L, const double *slots_u, const EpiArgs &eu, int part) {
  const BlockTables &bt = L.bt;
  const SymExchange &x = L.sv;
  if (!bt.built || !x.active() || !L.bt_symrow || !L.sv_shidx) { set_error("launch_fix_sym: level has no symmetric exchange tables"); return -1; }
  FixSymArgs f{bt.sh_nodes, bt.sh_ptr, bt.sh_list, slots_u, (uint32_t)bt.n_shared_slots, part == 0 ? x.n_sh : bt.n_shared,
               L.sv_shidx, L.bt_symrow, x.ptr, x.pos, x.split, x.ypart, x.sendbuf, x.recvbuf, eu};
  if (f.n_rows <= 0) return 0;
  const dim3 grid((unsigned)((f.n_rows + 255) / 256)), block(256);
  if (part == 0) {
    if (L.dim == 3) hipLaunchKernelGGL((k_fix_sym_a<3>), grid, block, 0, s, f);
    else hipLaunchKernelGGL((k_fix_sym_a<2>), grid, block, 0, s, f);
  } else {
    if (L.dim == 3) hipLaunchKernelGGL((k_fix_sym_b<3>), grid, block, 0, s, f);
    else hipLaunchKernelGGL((k_fix_sym_b<2>), grid, block, 0, s, f);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_mass_smoother(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const SmootherProgram &prog,
                         const double *dinv, const double *b, const double *x_in, double *x_out, double *resid_out,
                         bool use_ell) {
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (!L.mass_fusable || prog.n < 1 || prog.n > SM_MAX_STEPS) { set_error("fused mass smoother: level or program does not fit"); return -1; }
  const CellArgs a = make_cell_args(L, cd, nullptr, nullptr, nullptr, nullptr);
  const MassSmArgs m{L.pull_ptr_p, L.pull_list_p, L.cmask_p, dinv, b, x_in, x_out, resid_out, L.n_nodes_p};
  const dim3 grid(1), block(MS_THREADS);
  if (use_ell) {
    if (!L.ell_col || !cd.ell_val || cd.ell_stride != L.ell_stride) { set_error("mass rows not assembled"); return -4; }
    const EllMulArgs em{L.ell_col, cd.ell_val, L.n_nodes_p, L.ell_stride, L.cmask_p, nullptr, EpiArgs()};
    if (L.dim == 3) hipLaunchKernelGGL((k_mass_smoother_ell<27>), grid, block, 0, s, em, m, prog);
    else hipLaunchKernelGGL((k_mass_smoother_ell<9>), grid, block, 0, s, em, m, prog);
    AGMG_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const bool perq = cd.nqv > 1, gen = L.geom == AGMG_GEOM_GENERAL;
#define AGMG_MS_LAUNCH(D, G)                                                                     \
  do {                                                                                           \
    if (perq) hipLaunchKernelGGL((k_mass_smoother<D, true, G>), grid, block, 0, s, a, m, prog);  \
    else hipLaunchKernelGGL((k_mass_smoother<D, false, G>), grid, block, 0, s, a, m, prog);      \
  } while (0)
  if (L.dim == 3) { if (gen) AGMG_MS_LAUNCH(3, true); else AGMG_MS_LAUNCH(3, false); }
  else { if (gen) AGMG_MS_LAUNCH(2, true); else AGMG_MS_LAUNCH(2, false); }
#undef AGMG_MS_LAUNCH
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_diag_cells(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, double *slots_u, double *slots_p) {
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (L.n_cells <= 0) return 0;
  const CellArgs a = make_cell_args(L, cd, nullptr, nullptr, slots_u, slots_p);
  const int nv = L.dim == 3 ? 27 : 9, np = L.dim == 3 ? 8 : 4;
  const int gen = L.geom == AGMG_GEOM_GENERAL;
  const dim3 block(256);
  const dim3 gv((unsigned)(((long)L.n_cells * nv + 255) / 256)), gp((unsigned)(((long)L.n_cells * np + 255) / 256));
  if (L.dim == 3) {
    hipLaunchKernelGGL((k_diag_v<3>), gv, block, 0, s, a, cd.is_compressible, cd.nqv, gen);
    hipLaunchKernelGGL((k_diag_p<3>), gp, block, 0, s, a, cd.nqv, gen);
  } else {
    hipLaunchKernelGGL((k_diag_v<2>), gv, block, 0, s, a, cd.is_compressible, cd.nqv, gen);
    hipLaunchKernelGGL((k_diag_p<2>), gp, block, 0, s, a, cd.nqv, gen);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

// node range and exchange buffers of one phase of a pull over a node space
static bool pull_phase(PullArgs &p, const Exchange &x, int n_nodes, int n_owned, uint32_t recv_base, int phase) {
  p.recv_base = 0xFFFFFFFFu; p.recv = nullptr; p.ghost_out = nullptr;
  if (phase == PH_GHOST) {
    p.node0 = n_owned; p.n_range = n_nodes - n_owned; p.ghost_out = x.ghostbuf;
  } else {
    p.node0 = 0; p.n_range = phase == PH_OWNED ? n_owned : n_nodes;
    if (x.n_export > 0) { p.recv_base = recv_base; p.recv = x.expbuf; }
  }
  return p.n_range > 0;
}

int launch_pull_v(hipStream_t s, const DeviceLevel &L, const double *slots, const EpiArgs &e, int phase) {
  PullArgs p{L.pull_ptr_v, L.pull_list_v, slots, L.cmask_v, 0, 0, 0, nullptr, nullptr, e};
  const uint32_t recv_base = (uint32_t)((L.dim == 3 ? 27 : 9) * (size_t)L.cstride);
  if (!pull_phase(p, L.xv, L.n_nodes_v, L.n_owned_v, recv_base, phase)) return 0;
  const long n = (long)p.n_range * L.dim;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (L.dim == 3) hipLaunchKernelGGL((k_pull<3, 1>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((k_pull<2, 1>), grid, block, 0, s, p);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_pull_p(hipStream_t s, const DeviceLevel &L, const double *slots, int n_parts, const EpiArgs &e,
                  int phase) {
  PullArgs p{L.pull_ptr_p, L.pull_list_p, slots, L.cmask_p, 0, 0, 0, nullptr, nullptr, e};
  const uint32_t recv_base = (uint32_t)((L.dim == 3 ? 8 : 4) * (size_t)L.cstride);
  if (!pull_phase(p, L.xp, L.n_nodes_p, L.n_owned_p, recv_base, phase)) return 0;
  const long n = p.n_range;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (n_parts == 1) hipLaunchKernelGGL((k_pull<1, 1>), grid, block, 0, s, p);
  else if (n_parts == 2) hipLaunchKernelGGL((k_pull<1, 2>), grid, block, 0, s, p);
  else if (n_parts == 3) hipLaunchKernelGGL((k_pull<1, 3>), grid, block, 0, s, p);
  else { set_error("bad n_parts"); return -1; }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

bool blk_seq_possible(const DeviceLevel &L, const CoeffData &cd) {
  return L.bt.built && L.bt.n_blocks == 1 && L.bt.n_shared_owned == 0 && L.bt.n_shared_slots == 0 && cd.set && L.n_cells > 0 &&
         !cd.newton && !cd.dilation;
}
int launch_op_seq(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const SeqArgs &q, double *slots_u) {
  if (!blk_seq_possible(L, cd)) { set_error("launch_op_seq: level is not a single block"); return -1; }
  if (q.n < 1 || q.n > SEQ_MAX_STEPS) { set_error("launch_op_seq: bad step count"); return -1; }
  const BlockTables &bt = L.bt;
  const CellArgs a = make_cell_args(L, cd, nullptr, nullptr, slots_u, nullptr);
  BlkArgs b{nullptr, reinterpret_cast<const uint4 *>(bt.blk_hdr), bt.unodes, reinterpret_cast<const uint4 *>(bt.uent8), bt.lidT, slots_u,
            nullptr, 0};
  const bool perq = cd.nqv > 1;
  const bool comp = cd.is_compressible != 0;
  const bool gen = L.geom == AGMG_GEOM_GENERAL;
  int rc;
  if (L.dim == 3) rc = gen ? launch_blk_seq_t<3, true>(s, a, b, q, bt.max_unodes, comp, perq) : launch_blk_seq_t<3, false>(s, a, b, q, bt.max_unodes, comp, perq);
  else rc = gen ? launch_blk_seq_t<2, true>(s, a, b, q, bt.max_unodes, comp, perq) : launch_blk_seq_t<2, false>(s, a, b, q, bt.max_unodes, comp, perq);
  if (rc) return rc;
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_op(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
              const double *src_p, double *slots_u, double *slots_p, const EpiArgs &eu, bool use_blocks, int phase,
              const EpiArgs *ep) {
  // ep (Stokes): also finish the pressure rows (pull of slots_p with epilogue *ep) -- all of them (PH_ALL), the ghost rows
  // into the compress buffer (PH_GHOST / PH_INTERFACE) or the owned rows with the received sums (PH_OWNED); on the block
  // path the velocity fix-up and the pressure pull of a phase share one launch
  if (ep && (which != AGMG_OP_STOKES || phase == PH_INTERIOR)) { set_error("launch_op: pressure epilogue misuse"); return -1; }
  const int pphase = phase == PH_INTERFACE ? PH_GHOST : phase;
  // (the Newton-derivative variant of the Stokes cell kernel exists on the slot path only: it runs
  // once per outer Krylov iteration)
  if (!use_blocks || !L.bt.built || (eu.epi != EPI_STORE && eu.epi != EPI_RESID && eu.epi != EPI_CHEB) ||
      ((cd.newton || cd.dilation) && which == AGMG_OP_STOKES)) {
    // the slot path has no interior / interface split: everything waits for the ghost import
    if (phase == PH_INTERIOR) return 0;
    if (phase == PH_CELLS) { set_error("launch_op: PH_CELLS needs the block path"); return -1; }
    if (phase == PH_INTERFACE) phase = PH_GHOST;
    if (phase != PH_OWNED) {
      const int rc = launch_op_cells(s, L, cd, which, src_u, src_p, slots_u, slots_p);
      if (rc) return rc;
    }
    const int rc = launch_pull_v(s, L, slots_u, eu, phase);
    if (rc || !ep) return rc;
    return launch_pull_p(s, L, slots_p, 1, *ep, pphase);
  }
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (L.n_cells <= 0) return 0;
  const BlockTables &bt = L.bt;
  // fix-up over the shared nodes of one phase: ghosts -> Exchange::ghostbuf, owned -> dst (+ received sums)
  auto fix = [&](int i0, int n_range, bool ghost) -> int {
    if (n_range < 0) n_range = 0;
    FixArgs f{bt.sh_nodes, bt.sh_ptr, bt.sh_list, slots_u, i0, n_range, 0xFFFFFFFFu, nullptr, nullptr, L.n_owned_v, eu};
    if (ghost) f.ghost_out = L.xv.ghostbuf;
    else if (L.xv.n_export > 0) { f.recv_base = (uint32_t)bt.n_shared_slots; f.recv = L.xv.expbuf; }
    const unsigned nf = (unsigned)((n_range + 255) / 256);
    if (!ep) {
      if (!nf) return 0;
      if (L.dim == 3) hipLaunchKernelGGL((k_fix<3>), dim3(nf), dim3(256), 0, s, f);
      else hipLaunchKernelGGL((k_fix<2>), dim3(nf), dim3(256), 0, s, f);
    } else {   // the pressure rows of the same phase in the same launch
      PullArgs p{L.pull_ptr_p, L.pull_list_p, slots_p, L.cmask_p, 0, 0, 0, nullptr, nullptr, *ep};
      const uint32_t recv_base = (uint32_t)((L.dim == 3 ? 8 : 4) * (size_t)L.cstride);
      const unsigned np = pull_phase(p, L.xp, L.n_nodes_p, L.n_owned_p, recv_base, pphase) ? (unsigned)((p.n_range + 255) / 256) : 0u;
      if (nf + np == 0) return 0;
      if (L.dim == 3) hipLaunchKernelGGL((k_fix_pull<3>), dim3(nf + np), dim3(256), 0, s, f, p, nf);
      else hipLaunchKernelGGL((k_fix_pull<2>), dim3(nf + np), dim3(256), 0, s, f, p, nf);
    }
    AGMG_HIP_CHECK(hipGetLastError());
    return 0;
  };
  if (phase == PH_OWNED) return fix(0, bt.n_shared_owned, false);
  const CellArgs a = make_cell_args(L, cd, src_u, src_p, slots_u, slots_p);
  static unsigned long long *dbg_buf = nullptr;
#if AGMG_ABLATE & 1024
  if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, (size_t)4000000 * 8 * 8);
#endif
  BlkArgs b{dbg_buf, reinterpret_cast<const uint4 *>(bt.blk_hdr), bt.unodes, reinterpret_cast<const uint4 *>(bt.uent8), bt.lidT, slots_u,
            nullptr, 0};
  int n_launch = bt.n_blocks;
  if (phase == PH_INTERIOR || phase == PH_INTERFACE) {
    if (!bt.blk_order) { set_error("launch_op: level has no interior / interface block lists"); return -1; }
    b.blk_list = bt.blk_order;
    b.blk_first = phase == PH_INTERIOR ? 0 : bt.n_interior;
    n_launch = phase == PH_INTERIOR ? bt.n_interior : bt.n_blocks - bt.n_interior;
  }
  const bool perq = cd.nqv > 1;
  const bool comp = (cd.is_compressible != 0 || cd.dilation) && which != AGMG_OP_BT;
  const int lds = bt.max_unodes;
  const bool gen = L.geom == AGMG_GEOM_GENERAL;
  int rc = 0;
  if (n_launch > 0) {
    if (L.dim == 3) rc = gen ? launch_blk_op<3, true>(s, a, b, eu, n_launch, lds, which, comp, perq)
                             : launch_blk_op<3, false>(s, a, b, eu, n_launch, lds, which, comp, perq);
    else rc = gen ? launch_blk_op<2, true>(s, a, b, eu, n_launch, lds, which, comp, perq)
                  : launch_blk_op<2, false>(s, a, b, eu, n_launch, lds, which, comp, perq);
  }
  if (rc) return rc;
  if (phase == PH_INTERIOR || phase == PH_CELLS) { AGMG_HIP_CHECK(hipGetLastError()); return 0; }
  AGMG_HIP_CHECK(hipGetLastError());
#if AGMG_ABLATE & 1024
  {
    (void)hipDeviceSynchronize();
    const size_t nw = (size_t)bt.n_blocks * WAVES_PER_BLOCK;
    std::vector<unsigned long long> h(nw * 8);
    (void)hipMemcpy(h.data(), b.dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double d[5] = {0, 0, 0, 0, 0};
    for (size_t w = 0; w < nw; ++w)
      for (int k = 0; k < 5; ++k) d[k] += (double)(h[w * 8 + k + 1] - h[w * 8 + k]);
    double p1 = 0, p2 = 0, p3 = 0;
    for (size_t w = 0; w < nw; ++w) {
      p1 += (double)(h[w * 8 + 6] - h[w * 8 + 0]);
      p2 += (double)(h[w * 8 + 7] - h[w * 8 + 6]);
      p3 += (double)(h[w * 8 + 1] - h[w * 8 + 7]);
    }
    fprintf(stderr, "stamps (mean cycles/wave): prologue %.0f [issue+src->LDS %.0f | barrier+gather+barrier %.0f | park %.0f] | cell_compute %.0f | park %.0f | barrier %.0f | pull %.0f\n",
            d[0] / nw, p1 / nw, p2 / nw, p3 / nw, d[1] / nw, d[2] / nw, d[3] / nw, d[4] / nw);
  }
#endif
  if (phase == PH_GHOST || phase == PH_INTERFACE) return fix(bt.n_shared_owned, bt.n_shared - bt.n_shared_owned, true);
  return fix(0, bt.n_shared, false);
}

}  // namespace agmg
