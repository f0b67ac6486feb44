// kernels_brickq.hip -- velocity-row operators on structured 3-D levels, brick by brick, with one QUAD of
// lanes per cell (gfx950).
//
// What this replaces in the reference (tjhei/aspect 3.1.0-pre):
//   ABlockOperator / StokesOperator / BTBlockOperator ::local_apply + the deal.II cell_loop and
//   MatrixFreeOperators::Base::vmult around it (matrix_free_operators.cc:171-278, 399-432, 705-795)
//
// Design.  Same brick decomposition as kernels_brick.hip (BrickTables: 8 x 8 x Z cells, a
// 17 x 17 x (2Z+1) node lattice, first-brick ownership, partial sums of the brick surface finished by
// k_brick_fix), different work mapping:
//   * a workgroup = 4 wavefronts = one 8 x 8 layer; a cell is a QUAD of lanes (4 q + c): lanes c = 0, 1, 2 hold
//     the three velocity components, lane 3 idles (it shadows component 0 and never stores).  Each wavefront
//     owns a 4 x 4 patch of cells.
//   * ROTATED FRAMES: lane c works in local axes (0, 1, 2) = global axes (c, c+1, c+2) mod 3, i.e. its 27 nodal
//     values are gathered with the index triple rotated by c.  The symmetric gradient needs d u_d / d x_c from
//     the lane of component d: in rotated frames that is local derivative 2 of the NEXT lane of the quad and
//     local derivative 1 of the one after it -- the same register in every lane, so the exchange is two
//     `v_mov_b32 ... quad_perm` per value: no LDS, no barrier.  The lanes of a quad walk the Gauss points in
//     their own (rotated) orders; the points are therefore processed in ORBITS of the cyclic shift
//     (x, y, z) -> (y, z, x): within an orbit the three lanes visit the same three global points.
//   * the layer's inputs (two lattice planes of the source vector, the node words, the brick-ordered
//     coefficient rows) are loaded into registers right after the gather of the previous layer and written to
//     LDS at the top of the next one (issue early / write late): their latency hides under the arithmetic.
//   * after the cell operation a lane un-rotates its 27 results (selects by c), then the x / y neighbour
//     hand-over, the carried top plane and the node-wise write-out with the vector epilogue work in the global
//     frame exactly as in kernels_brick.hip.
// Fixed order of every sum: bitwise reproducible.
#include "agmg_internal.h"
#include "kernels_cell.h"

#include <cstdio>
#include <vector>

namespace agmg {

namespace {

constexpr int BQ_T = 256;
constexpr int BQ_PLANE_D = BRICK_PLANE * 3;      // doubles of one lattice plane [node][component]
constexpr int BQ_OUT_D = 2 * BQ_PLANE_D;         // out tile: two planes
constexpr int BQ_FACE_D = 9 * 64 * 3;            // hand-over / carry tile [m][cell][component]
constexpr int BQ_WRING = 5;                      // planes of node words kept in LDS
constexpr int QP_NEXT = 0xC9;                    // quad_perm:[1,2,0,3]: lane c reads lane (c+1) mod 3
constexpr int QP_NEXT2 = 0xD2;                   // quad_perm:[2,0,1,3]: lane c reads lane (c+2) mod 3

template <int CTRL> __device__ __forceinline__ double quad_fetch(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// the cyclic shift of a local point index p = x + 3 y + 9 z:  (x, y, z) -> (y, z, x)
__host__ __device__ constexpr int sig(const int p) { return (p / 3) % 3 + 3 * (p / 9) + 9 * (p % 3); }

// phase stamps of the layer loop (diagnostic build -DAGMG_ABLATE=4096 only)
#if AGMG_ABLATE & 4096
#define BQ_STAMP(k)                                                               \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long t_;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
    __builtin_amdgcn_sched_barrier(0);                                            \
    st_acc[k] += t_ - st_last;                                                    \
    st_last = t_;                                                                 \
  } while (0)
#else
#define BQ_STAMP(k)
#endif

struct BqArgs {
  unsigned long long *dbg;
  const uint32_t *gnode, *bcell;
  const double *geo;          // [brick][layer][rows][64]: viscosity rows, then the three cell extents
  const double *src_u, *src_p;
  const uint32_t *cnpT;
  double *partials, *slots_p;
  double lambda;
  int Z, latn, npb, rows, cstride;
  int brick0;                 // first brick of this launch (distributed levels: interior bricks, then interface bricks)
};

// 3-way select by the lane's component
__device__ __forceinline__ double sel3(const int c, const double a0, const double a1, const double a2) {
  return c == 0 ? a0 : (c == 1 ? a1 : a2);
}

template <int OP, bool COMP, bool PERQ>
__global__ void __launch_bounds__(BQ_T, 2)
k_brickq(const BqArgs B, const EpiArgs E) {
  constexpr int NQ = PERQ ? 27 : 1;
  constexpr int ROWS = NQ + 4;
  constexpr int GEO_D = ROWS * 64 > BQ_FACE_D ? ROWS * 64 : BQ_FACE_D;   // also hosts the y-face tile
  constexpr int GEO_PT = (ROWS * 32 + BQ_T - 1) / BQ_T;                  // double2 pieces per thread
  __shared__ double src_t[3 * BQ_PLANE_D];
  __shared__ __attribute__((aligned(16))) double geo_t[GEO_D];
  __shared__ double tile[BQ_OUT_D];      // a_x = 2 faces / the parked carry
  __shared__ double out_t[BQ_OUT_D];     // finished planes [plane][node][component]
  __shared__ uint32_t wring[BQ_WRING * BRICK_PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = lane >> 2, cq = lane & 3;
  const bool live = cq < 3;
  const int c_lane = live ? cq : 0;
  const int ii_lane = 4 * (wave & 1) + (quad & 3), jj_lane = 4 * (wave >> 1) + (quad >> 2);
  const int cl_lane = ii_lane + 8 * jj_lane;
  const int brick = B.brick0 + (int)xcd_contiguous(blockIdx.x, gridDim.x);
  const uint32_t *const gn = B.gnode + (size_t)brick * B.latn;
  const size_t pbase = (size_t)brick * B.npb;
  const int nplanes_b = 2 * B.Z + 1;
  // ---- staging registers: node words and source values of two lattice planes (thread t: nodes t, t + 256,
  // t + 512 of the 578), the coefficient rows of a layer (double2 pieces)
  uint32_t sw[3], swn[3];
  double sv[3][3];
  double gva[GEO_PT], gvb[GEO_PT];
  // (every staging load is unconditional, with clamped indices: a load under a condition meets the old value of
  // its register at the join, and the copy that merges them waits for the load)
  auto words_load = [&](uint32_t (&w)[3], const int K0_) {
    const int K0 = min(K0_, nplanes_b - 2);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int n = min(tid + BQ_T * m, 2 * BRICK_PLANE - 1);
      w[m] = gn[BRICK_PLANE * K0 + n];
    }
  };
  auto src_load = [&](const uint32_t (&w)[3]) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const double *sp = B.src_u + (size_t)(w[m] & BRICK_NODE_MASK) * 3;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) sv[m][cc] = sp[cc];
    }
  };
  auto geo_load = [&](const int layer) {
    const double2 *g2 = reinterpret_cast<const double2 *>(B.geo + ((size_t)brick * B.Z + min(layer, B.Z - 1)) * (ROWS * 64));
#pragma unroll
    for (int m = 0; m < GEO_PT; ++m) {
      const int n = min(tid + BQ_T * m, ROWS * 32 - 1);
      const double2 v2 = g2[n];
      gva[m] = v2.x; gvb[m] = v2.y;
    }
  };
  auto stage_write = [&](const uint32_t (&w)[3], const int K0) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int n = tid + BQ_T * m;
      if (n < 2 * BRICK_PLANE) {
        const int pl = n >= BRICK_PLANE ? 1 : 0, r = n - BRICK_PLANE * pl;
        const int K = K0 + pl;
        if (OP != KOP_BT) {
          double *d = src_t + (K % 3) * BQ_PLANE_D + r * 3;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) d[cc] = ((w[m] >> (29 + cc)) & 1u) ? 0.0 : sv[m][cc];
        }
        wring[(K % BQ_WRING) * BRICK_PLANE + r] = w[m];
      }
    }
#pragma unroll
    for (int m = 0; m < GEO_PT; ++m) {
      const int n = tid + BQ_T * m;
      if (n < ROWS * 32) reinterpret_cast<double2 *>(geo_t)[n] = make_double2(gva[m], gvb[m]);
    }
  };

  // ---- prologue: plane 0, then planes 1 and 2 with the first layer's coefficient rows
  {
    uint32_t w0[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int n = tid + BQ_T * m;
      w0[m] = gn[min(n, BRICK_PLANE - 1)];
    }
    words_load(sw, 1);
    words_load(swn, 3);
    if (OP != KOP_BT) src_load(w0);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int n = tid + BQ_T * m;
      if (n < BRICK_PLANE) {
        if (OP != KOP_BT) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) src_t[n * 3 + cc] = ((w0[m] >> (29 + cc)) & 1u) ? 0.0 : sv[m][cc];
        }
        wring[n] = w0[m];
      }
    }
    if (OP != KOP_BT) src_load(sw);
    geo_load(0);
  }

  double carry[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) carry[m] = 0.0;

#if AGMG_ABLATE & 4096
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
#pragma unroll 1
  for (int k = 0; k < B.Z; ++k) {
    BQ_STAMP(7);
    stage_write(sw, 2 * k + 1);
    __syncthreads();
    BQ_STAMP(0);
    // (the lane's address arithmetic is re-derived per layer from opaque copies: hoisted out of the layer loop
    // it would occupy some sixty registers across the cell operation)
    int c = c_lane, cl = cl_lane;
    asm volatile("" : "+v"(c), "+v"(cl));
    int gax[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) gax[r] = (c + r >= 3) ? c + r - 3 : c + r;
    const int ii = cl & 7, jj = cl >> 3;
    double *const my_face = tile + cl * 3 + c;            // + m * 192
    double *const my_yface = geo_t + cl * 3 + c;
    // ---- gather in the rotated frame: u[a0 + 3 a1 + 9 a2] = value at the lattice node whose coordinate along
    // global axis gax[r] is 2 * (cell coordinate) + a_r
    double u[27];
    if (OP != KOP_BT) {
      // (index arithmetic with 0 / 1 masks: nested selects here came out as divergent branches)
      int zoff[3], T[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a) zoff[a] = ((2 * k + a) % 3) * BQ_PLANE_D;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int m1 = gax[r] == 1 ? 1 : 0, m2 = gax[r] == 2 ? 1 : 0;
        const int str = (3 + 48 * m1) * (1 - m2);      // in-plane stride of the axis: 3 (x), 51 (y), 0 (z)
#pragma unroll
        for (int a = 0; a < 3; ++a) T[r][a] = a * str + m2 * zoff[a];
      }
      const int base0 = ((2 * jj) * BRICK_LAT + 2 * ii) * 3 + c;
#pragma unroll
      for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
        for (int a1 = 0; a1 < 3; ++a1) {
          const int b12 = base0 + T[1][a1] + T[2][a2];
#pragma unroll
          for (int a0 = 0; a0 < 3; ++a0) u[a0 + 3 * a1 + 9 * a2] = src_t[b12 + T[0][a0]];
        }
    }
    // geometry and coefficients of this lane's cell, in the rotated frame
    double ih[3];    // 1 / (Gauss point spacing * cell extent) along the local axes
#pragma unroll
    for (int r = 0; r < 3; ++r) ih[r] = geo_t[(NQ + gax[r]) * 64 + cl];
    const double det = geo_t[(NQ + 3) * 64 + cl];
    const double eta_c = PERQ ? 1.0 : geo_t[cl];
    int es[3];   // stride of the viscosity tile along local axis r
#pragma unroll
    for (int r = 0; r < 3; ++r) es[r] = 64 * (1 + 2 * gax[r] * gax[r]);   // 64 * 3^gax
    // the carried plane waits in LDS while the cell operation needs the registers
    if (live) {
#pragma unroll
      for (int m = 0; m < 9; ++m) my_face[m * 192] = carry[m];
    }

    BQ_STAMP(1);
    // ---- cell operation (rotated frame)
    double G[27];
    {
      const double ce = det * eta_c;
      // flux along local axis r:  f * (P[r] * dref_r u_c + Q[r] * dref_c u_{c+r})   (r = 0: both are the same)
      const double P0 = 2.0 * ce * ih[0] * ih[0], P1 = ce * ih[1] * ih[1], P2 = ce * ih[2] * ih[2];
      const double Q1 = ce * ih[1] * ih[0], Q2 = ce * ih[2] * ih[0];
      // compressible: - 2/3 eta div u on the diagonal:  f * C0 * (ih0 d0 + ih1 z1 + ih2 z2)
      const double C0 = -(2.0 / 3.0) * ce * ih[0];
#pragma unroll
      for (int l = 0; l < 9; ++l) interp3(u[3 * l], u[3 * l + 1], u[3 * l + 2]);
#pragma unroll
      for (int l = 0; l < 9; ++l) {
        const int b = (l % 3) + 9 * (l / 3);
        interp3(u[b], u[b + 3], u[b + 6]);
      }
#pragma unroll
      for (int l = 0; l < 9; ++l) interp3(u[l], u[l + 9], u[l + 18]);
#pragma unroll
      for (int a = 0; a < 27; ++a) G[a] = 0.0;
      const double *const eta_p = geo_t + cl;
#pragma unroll
      for (int p = 0; p < 27; ++p) {
        const int p1 = sig(p), p2 = sig(p1);
        if (p1 < p || p2 < p) continue;          // orbits are entered at their smallest point
        const int n = p1 == p ? 1 : 3;
        const int pts[3] = {p, p1, p2};
        __builtin_amdgcn_sched_barrier(0);   // (orbit by orbit: keeps the register pressure of the unrolled loop down)
        double D0[3], D1[3], D2[3], et[3];
#pragma unroll
        for (int t = 0; t < n; ++t) {
          const int x = pts[t] % 3, y = (pts[t] / 3) % 3, z = pts[t] / 9;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            if (dcol(x, m) != 0.0) s0 = fma(dcol(x, m), u[m + 3 * y + 9 * z], s0);
            if (dcol(y, m) != 0.0) s1 = fma(dcol(y, m), u[x + 3 * m + 9 * z], s1);
            if (dcol(z, m) != 0.0) s2 = fma(dcol(z, m), u[x + 3 * y + 9 * m], s2);
          }
          D0[t] = s0; D1[t] = s1; D2[t] = s2;
          if (PERQ) et[t] = eta_p[x * es[0] + y * es[1] + z * es[2]];
        }
#pragma unroll
        for (int t = 0; t < n; ++t) {
          const int x = pts[t] % 3, y = (pts[t] / 3) % 3, z = pts[t] / 9;
          const double wq = wq1(x) * wq1(y) * wq1(z);
          const double f = PERQ ? wq * et[t] : wq;
          // d u_{c+1} / d x_c and d u_{c+2} / d x_c at this point, from the next two lanes of the quad
          const double Y1 = quad_fetch<QP_NEXT>(D2[(t + 1) % n]);
          const double Y2 = quad_fetch<QP_NEXT2>(D1[(t + 2) % n]);
          double F0 = (f * P0) * D0[t];
          const double F1 = f * fma(P1, D1[t], Q1 * Y1);
          const double F2 = f * fma(P2, D2[t], Q2 * Y2);
          if (COMP) {
            const double Z1 = quad_fetch<QP_NEXT>(D0[(t + 1) % n]);
            const double Z2 = quad_fetch<QP_NEXT2>(D0[(t + 2) % n]);
            F0 = fma(f * C0, fma(ih[0], D0[t], fma(ih[1], Z1, ih[2] * Z2)), F0);
          }
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            if (dcol(x, m) != 0.0) G[m + 3 * y + 9 * z] = fma(dcol(x, m), F0, G[m + 3 * y + 9 * z]);
            if (dcol(y, m) != 0.0) G[x + 3 * m + 9 * z] = fma(dcol(y, m), F1, G[x + 3 * m + 9 * z]);
            if (dcol(z, m) != 0.0) G[x + 3 * y + 9 * m] = fma(dcol(z, m), F2, G[x + 3 * y + 9 * m]);
          }
        }
      }
#pragma unroll
      for (int l = 0; l < 9; ++l) interp3T(G[3 * l], G[3 * l + 1], G[3 * l + 2]);
#pragma unroll
      for (int l = 0; l < 9; ++l) {
        const int b = (l % 3) + 9 * (l / 3);
        interp3T(G[b], G[b + 3], G[b + 6]);
      }
#pragma unroll
      for (int l = 0; l < 9; ++l) interp3T(G[l], G[l + 9], G[l + 18]);
    }
    BQ_STAMP(2);
    // ---- the next layer's inputs: issued here (behind the cell operation, whose registers they would crowd),
    // written to LDS at the top of the next iteration; the hand-over and the write-out below hide their latency
    if (OP != KOP_BT) src_load(swn);
#pragma unroll
    for (int m = 0; m < 3; ++m) sw[m] = swn[m];
    geo_load(k + 1);
    words_load(swn, 2 * k + 5);
    // ---- back to the global frame: Gg[i, j, k] = G[i, j, k] | G[j, k, i] | G[k, i, j] for c = 0 | 1 | 2
    double Gg[27];
#pragma unroll
    for (int a = 0; a < 27; ++a) {
      const int i = a % 3, j = (a / 3) % 3, kk = a / 9;
      Gg[a] = sel3(c, G[i + 3 * j + 9 * kk], G[j + 3 * kk + 9 * i], G[kk + 3 * i + 9 * j]);
    }
#pragma unroll
    for (int m = 0; m < 9; ++m) carry[m] = my_face[m * 192];
    // ---- sums over the cells of the layer that share a node, owner by owner: a lane owns the nodes a_x, a_y < 2
    // of its cell (the upper ring too at i = 7 / j = 7).  Every lane publishes its a_x = 2 face and its a_y = 2
    // face; an owner adds, in this order, the left cell's (2, a_y), the lower cell's (a_x, 2) and the diagonal
    // cell's (2, 2).  (Sums of nodes a lane does not own pick up stray terms and are never stored.)
    // (the a_y = 2 faces go where the coefficient rows are: every wavefront must be done with its cell operation,
    // which reads the viscosity rows to its end)
    __syncthreads();
    if (live) {
#pragma unroll
      for (int m = 0; m < 9; ++m) my_face[m * 192] = Gg[2 + 3 * m];              // m = a_y + 3 a_z
#pragma unroll
      for (int az = 0; az < 3; ++az)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) my_yface[(az * 3 + ax) * 192] = Gg[ax + 6 + 9 * az];
    }
    __syncthreads();
    BQ_STAMP(3);
    if (ii > 0) {
#pragma unroll
      for (int m = 0; m < 9; ++m) Gg[3 * m] += my_face[m * 192 - 3];
    }
    if (jj > 0) {
#pragma unroll
      for (int az = 0; az < 3; ++az)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) Gg[ax + 9 * az] += my_yface[(az * 3 + ax) * 192 - 24];
    }
    if (ii > 0 && jj > 0) {
#pragma unroll
      for (int az = 0; az < 3; ++az) Gg[9 * az] += my_face[(2 + 3 * az) * 192 - 27];
    }
    // the lower plane is complete with the previous layer's upper plane; the upper plane waits
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      Gg[m] += carry[m];
      carry[m] = Gg[18 + m];
    }
    BQ_STAMP(4);
    // finished planes -> out tile [plane][node][component]: the lane owns a_x, a_y < 2 (the upper ring too
    // for the lanes at i = 7 / j = 7)
    if (live) {
#pragma unroll
      for (int az = 0; az < 2; ++az)
#pragma unroll
        for (int ay = 0; ay < 3; ++ay)
#pragma unroll
          for (int ax = 0; ax < 3; ++ax)
            if ((ax < 2 || ii == 7) && (ay < 2 || jj == 7))
              out_t[(az * BRICK_PLANE + (2 * jj + ay) * BRICK_LAT + 2 * ii + ax) * 3 + c] = Gg[ax + 3 * ay + 9 * az];
    }
    __syncthreads();
    BQ_STAMP(5);
    // ---- node-wise write-out of planes 2k and 2k + 1 (thread t: nodes t, t + 256, t + 512 of the 578).  All
    // loads of a thread come before its first store, none of them under a lane-dependent condition (dst may
    // alias the epilogue's operands: the compiler keeps loads behind earlier stores, and a wait for a load also
    // waits for the stores in front of it)
    {
      uint32_t w[3];
      double v[3][3];
      bool act[3];
      bool any_con = false;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int n = tid + BQ_T * m;
        act[m] = n < 2 * BRICK_PLANE;
        const int nn = act[m] ? n : 0;
        const int pl = nn >= BRICK_PLANE ? 1 : 0, r = nn - BRICK_PLANE * pl;
        const int K = 2 * k + pl;
        w[m] = wring[(K % BQ_WRING) * BRICK_PLANE + r];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) v[m][cc] = out_t[nn * 3 + cc];
        any_con = any_con || ((w[m] >> 29) != 0 && !(w[m] & BRICK_PARTIAL));
      }
      // destination: the row itself, or this brick's partial-sum slot of a row another brick owns
      double *pp[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int nn = act[m] ? tid + BQ_T * m : 0;
        const int pl = nn >= BRICK_PLANE ? 1 : 0, r = nn - BRICK_PLANE * pl;
        const int K = 2 * k + pl;
        const int J = r / BRICK_LAT, I = r - BRICK_LAT * J;
        const int ring = J == 0 ? I : (J == BRICK_LAT - 1 ? 47 + I : 17 + 2 * (J - 1) + (I == BRICK_LAT - 1 ? 1 : 0));
        const int sidx = K == 0 ? r : BRICK_PLANE + 64 * (K - 1) + ring;
        const size_t t = (size_t)(w[m] & BRICK_NODE_MASK) * 3;
        pp[m] = (w[m] & BRICK_PARTIAL) ? B.partials + (pbase + (size_t)sidx) * 3 : E.dst + t;
      }
      if (E.epi == EPI_STORE && !__any(any_con)) {
        // plain product, no identity row among the wave's nodes: stores only
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (act[m]) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) pp[m][cc] = v[m][cc];
          }
      } else {
        double xv[3][3], bv[3][3], dv[3][3], ov[3][3];
        const bool need_x = E.epi == EPI_CHEB || E.x != nullptr, need_b = E.epi != EPI_STORE;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const size_t t = (size_t)(w[m] & BRICK_NODE_MASK) * 3;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            xv[m][cc] = need_x ? E.x[t + cc] : 0.0;
            bv[m][cc] = need_b ? E.b[t + cc] : 0.0;
            dv[m][cc] = E.epi == EPI_CHEB ? E.dinv[t + cc] : 0.0;
            ov[m][cc] = (E.epi == EPI_CHEB && E.xold) ? E.xold[t + cc] : 0.0;
          }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const bool fin = !(w[m] & BRICK_PARTIAL);
          double out[3];
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            const bool con = (w[m] >> (29 + cc)) & 1u;
            const double Ax = con ? xv[m][cc] : v[m][cc];
            double o;
            if (E.epi == EPI_STORE) o = Ax;
            else if (E.epi == EPI_RESID) o = bv[m][cc] - Ax;
            else o = fma(-E.f1, ov[m][cc], fma(E.f2 * dv[m][cc], bv[m][cc] - Ax, (1.0 + E.f1) * xv[m][cc]));
            out[cc] = fin ? o : v[m][cc];
          }
          if (act[m]) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) pp[m][cc] = out[cc];
          }
        }
      }
    }
  }
#if AGMG_ABLATE & 4096
  BQ_STAMP(6);
  if (lane == 0)
    for (int q = 0; q < 8; ++q) B.dbg[((size_t)brick * 4 + wave) * 8 + q] = st_acc[q];
#endif
  // ---- the brick's top plane
  __syncthreads();
  const int c = c_lane, ii = ii_lane, jj = jj_lane;
  if (live) {
#pragma unroll
    for (int ay = 0; ay < 3; ++ay)
#pragma unroll
      for (int ax = 0; ax < 3; ++ax)
        if ((ax < 2 || ii == 7) && (ay < 2 || jj == 7))
          out_t[((2 * jj + ay) * BRICK_LAT + 2 * ii + ax) * 3 + c] = carry[ax + 3 * ay];
  }
  __syncthreads();
  {
    const int K = 2 * B.Z;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int n = tid + BQ_T * m;
      if (n < BRICK_PLANE) {
        uint32_t w = wring[(K % BQ_WRING) * BRICK_PLANE + n];
        const bool fin = !(w & BRICK_PARTIAL);
        if (!fin) w = (w & 0xF0000000u) | (uint32_t)(BRICK_PLANE + 64 * (nplanes_b - 2) + n);
        const size_t t = (size_t)(w & BRICK_NODE_MASK) * 3;
        double *const pp = fin ? E.dst + t : B.partials + (pbase + (w & 0xFFFFu)) * 3;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const bool con = (w >> (29 + cc)) & 1u;
          const double val = out_t[n * 3 + cc];
          double o = val;
          if (fin) {
            const double xvv = (E.epi == EPI_CHEB || (con && E.x)) ? E.x[t + cc] : 0.0;
            const double Ax = con ? xvv : val;
            if (E.epi == EPI_STORE) o = Ax;
            else if (E.epi == EPI_RESID) o = E.b[t + cc] - Ax;
            else {
              o = fma(E.f2 * E.dinv[t + cc], E.b[t + cc] - Ax, (1.0 + E.f1) * xvv);
              if (E.xold) o = fma(-E.f1, E.xold[t + cc], o);
            }
          }
          pp[cc] = o;
        }
      }
    }
  }
}

// coefficient rows in brick order: geo[brick][layer][row][lane] = viscosity rows of the layer's cells, then
// 1 / (Gauss point spacing * extent) along x, y, z and the cell volume (MappingCartesian)
__global__ void __launch_bounds__(256) k_brick_geo(double *__restrict__ geo, const uint32_t *__restrict__ bcell,
                                                   const double *__restrict__ etaT, const double *__restrict__ hT,
                                                   const int nq, const size_t cs, const size_t n_layers) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int rows = nq + 4;
  if (t >= n_layers * rows * 64) return;
  const int ln = (int)(t & 63);
  const size_t lr = t >> 6;
  const size_t layer = lr / rows;
  const int row = (int)(lr - layer * rows);
  const uint32_t cell = bcell[layer * 64 + ln];
  double v;
  if (row < nq) v = etaT[(size_t)row * cs + cell];
  else if (row < nq + 3) v = INV_S / hT[(size_t)(row - nq) * cs + cell];
  else v = hT[cell] * hT[cs + cell] * hT[2 * cs + cell];
  geo[t] = v;
}

}  // namespace

bool brickq_supported(const DeviceLevel &L, const CoeffData &cd, int which) {
  return L.bk.built && L.dim == 3 && L.geom == AGMG_GEOM_CARTESIAN && cd.brick_geo != nullptr && which == AGMG_OP_A &&
         !cd.newton;
}

size_t brick_geo_doubles(const DeviceLevel &L, int nqv) { return (size_t)L.bk.n_bricks * L.bk.Z * (nqv + 4) * 64; }

int launch_brick_geo(hipStream_t s, const DeviceLevel &L, const CoeffData &cd) {
  if (!L.bk.built || !cd.brick_geo || !cd.etaT || !L.cell_hT) { set_error("brick coefficient rows: tables missing"); return -4; }
  const size_t n_layers = (size_t)L.bk.n_bricks * L.bk.Z;
  const size_t n = n_layers * (cd.nqv + 4) * 64;
  hipLaunchKernelGGL(k_brick_geo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cd.brick_geo, L.bk.bcell, cd.etaT, L.cell_hT,
                     cd.nqv, (size_t)L.cstride, n_layers);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_brickq(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u, const double *src_p,
                  double *partials, double *slots_p, const EpiArgs &eu, int brick0, int n_launch) {
  const BrickTables &bk = L.bk;
  if (n_launch < 0) n_launch = bk.n_bricks - brick0;
  if (brick0 < 0 || n_launch < 0 || brick0 + n_launch > bk.n_bricks) { set_error("launch_brickq: bad brick range"); return -1; }
  if (n_launch == 0) return 0;
  const bool perq = cd.nqv > 1;
  const bool comp = (cd.is_compressible != 0 || cd.dilation) && which != AGMG_OP_BT;
  static unsigned long long *dbg_buf = nullptr;
#if AGMG_ABLATE & 4096
  if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, (size_t)1 << 26);
#endif
  const BqArgs b{dbg_buf, bk.gnode, bk.bcell, cd.brick_geo, src_u, src_p, L.cnpT, partials, slots_p, cd.lambda,
                 bk.Z, bk.latn, bk.npb, cd.nqv + 4, L.cstride, brick0};
  const dim3 grid(n_launch), block(BQ_T);
  if (comp) {
    if (perq) hipLaunchKernelGGL((k_brickq<KOP_A, true, true>), grid, block, 0, s, b, eu);
    else hipLaunchKernelGGL((k_brickq<KOP_A, true, false>), grid, block, 0, s, b, eu);
  } else {
    if (perq) hipLaunchKernelGGL((k_brickq<KOP_A, false, true>), grid, block, 0, s, b, eu);
    else hipLaunchKernelGGL((k_brickq<KOP_A, false, false>), grid, block, 0, s, b, eu);
  }
  AGMG_HIP_CHECK(hipGetLastError());
#if AGMG_ABLATE & 4096
  {
    (void)hipDeviceSynchronize();
    const size_t nw = (size_t)bk.n_bricks * 4;
    std::vector<unsigned long long> h(nw * 8);
    (void)hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost);
    double d[8] = {0};
    for (size_t w = 0; w < nw; ++w)
      for (int q = 0; q < 8; ++q) d[q] += (double)h[w * 8 + q];
    fprintf(stderr, "brickq stamps (mean cycles per wave and layer, Z=%d): stage write+barrier %.0f | gather+issue+park %.0f | cell %.0f | unrotate+xface+barrier %.0f | x add, yface+barrier %.0f | y add, carry, put+barrier %.0f | write-out %.0f | loop head %.0f\n",
            bk.Z, d[0] / nw / bk.Z, d[1] / nw / bk.Z, d[2] / nw / bk.Z, d[3] / nw / bk.Z, d[4] / nw / bk.Z, d[5] / nw / bk.Z, d[6] / nw / bk.Z, d[7] / nw / bk.Z);
  }
#endif
  return 0;
}

}  // namespace agmg
