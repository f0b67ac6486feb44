// cpu_simd.cpp -- CPU baseline leg of bench.py: the velocity-row operators with SIMD ACROSS CELLS.
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (like oracle.cpp): nothing under aspect_amd/ links or loads this.
//
// What it restates: how the reference runs this path on a CPU.  deal.II's MatrixFree evaluates a BATCH of
// cells per call, one cell per SIMD lane (VectorizedArray<double>: 8 lanes with AVX-512, 4 with AVX2;
// matrix_free_operators.h:85-90, doc/sphinx/user/methods/geometric-multigrid.md:35-58), sum-factorised
// (FEEvaluation), threads off, one MPI rank per core.  Here: batches of LANES cells of one colour (cells of a
// colour share no node, so lanes and threads never write the same entry), the cell operation of
// matrix_free_operators.cc:171-278 / 399-432 / 705-740 on vector registers (GCC vector extensions; the
// interpolation to the Gauss points followed by the collocation derivative, as FEEvaluation does for
// degree + 1 points), OpenMP threads over the batches of a colour in place of MPI ranks.  Cartesian cells,
// no Newton / dilation / face terms: the workloads of bench.py; oracle.cpp stays the checker for everything.
//
// Built twice by oracle/Makefile (-march=x86-64-v4 with 8 lanes, -march=x86-64-v3 with 4) and chosen at run
// time by the host's CPUID, because the library is built in one container and timed on another machine.
#include <cstddef>
#include <cstdint>

#ifndef SIMD_LANES
#define SIMD_LANES 4
#endif
#ifndef SIMD_SUFFIX
#define SIMD_SUFFIX v3
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

namespace {

constexpr int W = SIMD_LANES;
typedef double vd __attribute__((vector_size(8 * SIMD_LANES)));

constexpr double SQ06 = 0.774596669241483377035853079956;
constexpr double XQ0 = 0.5 - 0.5 * SQ06;
constexpr double N00 = (1 - XQ0) * (1 - 2 * XQ0), N01 = 4 * XQ0 * (1 - XQ0), N02 = XQ0 * (2 * XQ0 - 1);
constexpr double M0 = 1 - XQ0, M1 = XQ0;
constexpr double INV_S = 1.0 / (0.5 * SQ06);
constexpr double WQ[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
constexpr double DC[3][3] = {{-1.5, 2.0, -0.5}, {-0.5, 0.0, 0.5}, {0.5, -2.0, 1.5}};   // collocation derivative * spacing
constexpr double N1[3][2] = {{M0, M1}, {0.5, 0.5}, {M1, M0}};

inline vd bc(double x) { vd v; for (int l = 0; l < W; ++l) v[l] = x; return v; }

inline void interp3(vd &a, vd &b, vd &c) {
  const vd t0 = N00 * a + N01 * b + N02 * c, t2 = N02 * a + N01 * b + N00 * c;
  a = t0; c = t2;
}
inline void interp3T(vd &a, vd &b, vd &c) {
  const vd o0 = N00 * a + N02 * c, o1 = N01 * (a + c) + b, o2 = N02 * a + N00 * c;
  a = o0; b = o1; c = o2;
}

}  // namespace

// mode: 0 Stokes, 1 A block, 2 B^T.  cells[0..n): cells of ONE colour.  du / dp are added to.
extern "C" void CAT(cpu_simd_cells_, SIMD_SUFFIX)(int mode, int compressible, int n, const int *cells, const uint32_t *cnv,
                                                   const uint32_t *cnp, const uint8_t *cmv, const uint8_t *cmp,
                                                   const double *cell_h, const double *eta, int nqv, double lambda,
                                                   const double *su, const double *sp, double *du, double *dp) {
  const bool has_u = mode != 2, has_p = mode != 1;
  for (int b0 = 0; b0 < n; b0 += W) {
    const int nl = n - b0 < W ? n - b0 : W;
    int cell[W];
    for (int l = 0; l < W; ++l) cell[l] = cells[b0 + (l < nl ? l : 0)];
    vd u[3][27], G[3][27], p8[8], pq[27], pout[8];
    vd ih[3], det = bc(1.0);
    for (int d = 0; d < 3; ++d) {
      vd h;
      for (int l = 0; l < W; ++l) h[l] = cell_h[(size_t)cell[l] * 3 + d];
      det *= h;
      ih[d] = INV_S / h;
    }
    if (has_u)
      for (int i = 0; i < 27; ++i)
        for (int l = 0; l < W; ++l) {
          const uint32_t node = cnv[(size_t)cell[l] * 27 + i];
          const uint8_t m = cmv[node];
          for (int c = 0; c < 3; ++c) u[c][i][l] = ((m >> c) & 1) ? 0.0 : su[(size_t)node * 3 + c];
        }
    if (has_p) {
      for (int i = 0; i < 8; ++i)
        for (int l = 0; l < W; ++l) {
          const uint32_t node = cnp[(size_t)cell[l] * 8 + i];
          p8[i][l] = (cmp[node] & 1) ? 0.0 : sp[node];
        }
      for (int q = 0; q < 27; ++q) {
        const int qx = q % 3, qy = (q / 3) % 3, qz = q / 9;
        vd s = bc(0.0);
        for (int i = 0; i < 8; ++i) s += (N1[qx][i & 1] * N1[qy][(i >> 1) & 1] * N1[qz][i >> 2]) * p8[i];
        pq[q] = s;
      }
    }
    // nodal values -> Gauss points
    if (has_u)
      for (int c = 0; c < 3; ++c) {
        for (int l = 0; l < 9; ++l) interp3(u[c][3 * l], u[c][3 * l + 1], u[c][3 * l + 2]);
        for (int l = 0; l < 9; ++l) { const int b = (l % 3) + 9 * (l / 3); interp3(u[c][b], u[c][b + 3], u[c][b + 6]); }
        for (int l = 0; l < 9; ++l) interp3(u[c][l], u[c][l + 9], u[c][l + 18]);
      }
    for (int c = 0; c < 3; ++c)
      for (int i = 0; i < 27; ++i) G[c][i] = bc(0.0);
    for (int i = 0; i < 8; ++i) pout[i] = bc(0.0);
    for (int q = 0; q < 27; ++q) {
      const int qi[3] = {q % 3, (q / 3) % 3, q / 9};
      const int st[3] = {1, 3, 9};
      const double wq = WQ[qi[0]] * WQ[qi[1]] * WQ[qi[2]];
      vd e;
      for (int l = 0; l < W; ++l) e[l] = eta[(size_t)cell[l] * nqv + (nqv == 1 ? 0 : q)];
      const vd jxw = wq * det;
      vd grad[3][3];   // physical d u_c / d x_d
      vd tau[3][3];
      for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) tau[c][d] = bc(0.0);
      vd div = bc(0.0);
      if (has_u) {
        for (int c = 0; c < 3; ++c)
          for (int d = 0; d < 3; ++d) {
            const int base = q - qi[d] * st[d];
            vd s = bc(0.0);
            for (int m = 0; m < 3; ++m)
              if (DC[qi[d]][m] != 0.0) s += DC[qi[d]][m] * u[c][base + m * st[d]];
            grad[c][d] = s * ih[d];
          }
        for (int c = 0; c < 3; ++c)
          for (int d = 0; d < 3; ++d) tau[c][d] = e * (grad[c][d] + grad[d][c]);
        div = grad[0][0] + grad[1][1] + grad[2][2];
        if (compressible)
          for (int d = 0; d < 3; ++d) tau[d][d] -= (2.0 / 3.0) * e * div;
      }
      if (has_p)
        for (int d = 0; d < 3; ++d) tau[d][d] -= lambda * pq[q];
      // test with the gradient: reference flux * JxW, transposed collocation derivative
      for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) {
          const vd f = tau[c][d] * ih[d] * jxw;
          const int base = q - qi[d] * st[d];
          for (int m = 0; m < 3; ++m)
            if (DC[qi[d]][m] != 0.0) G[c][base + m * st[d]] += DC[qi[d]][m] * f;
        }
      if (mode == 0) {
        const vd pt = -lambda * div * jxw;
        for (int i = 0; i < 8; ++i) pout[i] += (N1[qi[0]][i & 1] * N1[qi[1]][(i >> 1) & 1] * N1[qi[2]][i >> 2]) * pt;
      }
    }
    for (int c = 0; c < 3; ++c) {
      for (int l = 0; l < 9; ++l) interp3T(G[c][3 * l], G[c][3 * l + 1], G[c][3 * l + 2]);
      for (int l = 0; l < 9; ++l) { const int b = (l % 3) + 9 * (l / 3); interp3T(G[c][b], G[c][b + 3], G[c][b + 6]); }
      for (int l = 0; l < 9; ++l) interp3T(G[c][l], G[c][l + 9], G[c][l + 18]);
    }
    for (int l = 0; l < nl; ++l) {
      for (int i = 0; i < 27; ++i) {
        const uint32_t node = cnv[(size_t)cell[l] * 27 + i];
        const uint8_t m = cmv[node];
        for (int c = 0; c < 3; ++c)
          if (!((m >> c) & 1)) du[(size_t)node * 3 + c] += G[c][i][l];
      }
      if (mode == 0)
        for (int i = 0; i < 8; ++i) {
          const uint32_t node = cnp[(size_t)cell[l] * 8 + i];
          if (!(cmp[node] & 1)) dp[node] += pout[i][l];
        }
    }
  }
}
