// kernels_brick.hip -- the velocity-row operators on structured 3-D levels, brick by brick (gfx950).
//
// What this replaces in the reference (tjhei/aspect 3.1.0-pre), as kernels_ops.hip does:
//   ABlockOperator / StokesOperator / BTBlockOperator ::local_apply + the deal.II cell_loop and
//   MatrixFreeOperators::Base::vmult around it (matrix_free_operators.cc:171-278, 399-432, 705-795)
//
// Design (DESIGN.md "bricks").  A level whose cells tile into bricks of 8 x 8 x Z cells (every
// uniformly refined tree from level 3 on) is applied with ONE workgroup per brick: three
// wavefronts, one per velocity component, lane = cell (i + 8 j) of the current 8 x 8 layer.  The
// workgroup marches up the brick layer by layer:
//   * a lane gathers the 27 nodal values of its cell through the brick's node lattice (a compact
//     17 x 17 x (2Z+1) index list, 4 B per node, instead of 27 indices per cell);
//   * cell_compute (kernels_cell.h) is the same register-resident sum-factorised cell operation as
//     in the other kernels; the velocity-gradient rows are exchanged between the three waves
//     through two alternating LDS tiles, one workgroup barrier per line of three q-points;
//   * contributions to a node shared by x / y neighbours of the layer are summed by handing the
//     upper face of each cell to the neighbouring LANE (wave-private LDS tile, no barrier); the top
//     plane of a layer is carried in registers into the next layer.  No slot tile, no node tables,
//     fixed order of the sums: bitwise reproducible;
//   * a lane then owns the 2 x 2 x 2 lower nodes of its cell (plus the brick's upper ring for
//     the lanes at i = 7 / j = 7): rows this brick owns are stored to dst with the vector epilogue
//     fused, rows owned by another brick go to a compact partial-sum array (the brick's surface),
//     and k_brick_fix adds those shares to the owners' rows afterwards (all epilogues are affine
//     in the sum, so the correction is  dst += coefficient * partial sums).
// Algorithmic traffic per cell (A block, per-q viscosity): 38 B lattice + 4 B cell ids + 230 B
// source + 216 B viscosity + 24 B extents + 192 B dst + ~160 B partial sums and their fix-up.
#include "agmg_internal.h"
#include "kernels_cell.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace agmg {

namespace {

struct BrickArgs {
  const uint32_t *gnode, *bcell;
  const uint16_t *pidx;
  double *partials;
  int Z, latn, npb;
  unsigned long long *dbg;
};

constexpr int BRICK_THREADS = 192;
#ifndef BRICK_PIPE
#define BRICK_PIPE false
#endif
#ifndef BRICK_NPT
#define BRICK_NPT 1
#endif
// phase stamps of the layer loop (tools/ablate.sh 2048; never defined in the product build)
#if AGMG_ABLATE & 2048
#define BRICK_STAMP(k)                                                            \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                   \
    __builtin_amdgcn_sched_barrier(0);                                            \
    st_acc[k] += t_ - st_last;                                                    \
    st_last = t_;                                                                 \
  } while (0)
#else
#define BRICK_STAMP(k)
#endif
constexpr int NB_TILE = 584;      // doubles of the neighbour tile of one wave (9 x 64 for the hand-over; two 17 x 17 planes as out tile)

// LDS image of two lattice planes of the source vector, one tile per (component, plane): even
// lattice coordinates first, then the odd ones, rows 24 doubles apart, so that the 64 cell lanes
// (lattice stride 2 in both directions) read conflict-free
constexpr int SRC_ROW = 24, SRC_PLANE = 17 * SRC_ROW;
__device__ __forceinline__ int src_pos(const int I) { return (I & 1) ? 9 + (I >> 1) : (I >> 1); }

// Write-out of finished lattice planes, one thread per lattice node and all three components (24
// contiguous bytes per access): the lanes have put their node sums into the out tile [comp][node].
// A row this brick owns gets the vector epilogue and goes to dst; a row another brick owns goes, as
// this brick's share of the sum, to the brick's partial-sum slots.  The node words come from the LDS
// word tile: node | constraint bits, or BRICK_PARTIAL | constraint bits | surface index.  ALL loads of
// a thread are issued before its first store: dst may alias the epilogue's operands (Chebyshev: dst
// == xold), so the compiler keeps loads behind earlier stores.
constexpr int OUT_TILE = 584;   // doubles per component: two planes of 17 x 17 nodes (+ padding)
template <int EPI, int NPT>
__device__ __forceinline__ void brick_writeout_t(const double *ot, const uint32_t *word_t, const int K0, const int nnodes,
                                                 const int tid, const size_t pbase, const BrickArgs &B, const EpiArgs &E) {
  uint32_t w[NPT];
  double v[NPT][3], xv[NPT][3], bv[NPT][3], dv[NPT][3], ov[NPT][3];
  bool act[NPT];
#pragma unroll
  for (int m = 0; m < NPT; ++m) {
    const int n = tid + BRICK_THREADS * m;
    act[m] = n < nnodes;
    const int nn = act[m] ? n : 0;
    const int pl = nn >= BRICK_PLANE ? 1 : 0, r = nn - BRICK_PLANE * pl;
    const int J = r / BRICK_LAT, I = r - BRICK_LAT * J;
    w[m] = word_t[((K0 + pl) % 3) * SRC_PLANE + src_pos(J) * SRC_ROW + src_pos(I)];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) v[m][cc] = ot[cc * OUT_TILE + nn];
  }
#pragma unroll
  for (int m = 0; m < NPT; ++m) {
    const bool fin = act[m] && !(w[m] & BRICK_PARTIAL);
    const size_t t = (size_t)(w[m] & BRICK_NODE_MASK) * 3;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      xv[m][cc] = bv[m][cc] = dv[m][cc] = ov[m][cc] = 0.0;
      if (EPI == EPI_CHEB) {
        if (fin) { xv[m][cc] = E.x[t + cc]; bv[m][cc] = E.b[t + cc]; dv[m][cc] = E.dinv[t + cc]; }
        if (fin && E.xold) ov[m][cc] = E.xold[t + cc];
      } else {
        if (fin && ((w[m] >> (29 + cc)) & 1u) && E.x) xv[m][cc] = E.x[t + cc];
        if (EPI == EPI_RESID && fin) bv[m][cc] = E.b[t + cc];
      }
    }
  }
#pragma unroll
  for (int m = 0; m < NPT; ++m) {
    const bool fin = !(w[m] & BRICK_PARTIAL);
    const size_t t = (size_t)(w[m] & BRICK_NODE_MASK) * 3;
    double *const p = fin ? E.dst + t : B.partials + (pbase + (w[m] & 0xFFFFu)) * 3;
    double out[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const bool con = (w[m] >> (29 + cc)) & 1u;
      const double Ax = con ? xv[m][cc] : v[m][cc];
      double o;
      if (EPI == EPI_STORE) o = Ax;
      else if (EPI == EPI_RESID) o = bv[m][cc] - Ax;
      else {
        o = fma(E.f2 * dv[m][cc], bv[m][cc] - Ax, (1.0 + E.f1) * xv[m][cc]);
        if (E.xold) o = fma(-E.f1, ov[m][cc], o);
      }
      out[cc] = fin ? o : v[m][cc];
    }
    if (act[m]) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) p[cc] = out[cc];
    }
  }
}
// planes K0 .. of `nnodes` lattice nodes in all (289 or 578), two nodes per thread and pass
__device__ __forceinline__ void brick_writeout(const double *ot, const uint32_t *word_t, const int K0, const int nnodes,
                                               const int tid, const size_t pbase, const BrickArgs &B, const EpiArgs &E) {
  for (int first = 0; first < nnodes; first += BRICK_NPT * BRICK_THREADS) {
    const int nn = min(nnodes - first, BRICK_NPT * BRICK_THREADS);
    // (the pass offset moves into the tile / plane arithmetic through tid)
    if (E.epi == EPI_CHEB) brick_writeout_t<EPI_CHEB, BRICK_NPT>(ot, word_t, K0, first + nn, tid + first, pbase, B, E);
    else if (E.epi == EPI_RESID) brick_writeout_t<EPI_RESID, BRICK_NPT>(ot, word_t, K0, first + nn, tid + first, pbase, B, E);
    else brick_writeout_t<EPI_STORE, BRICK_NPT>(ot, word_t, K0, first + nn, tid + first, pbase, B, E);
  }
}
// the lanes' node sums of one lattice plane -> out tile (plane slot pl): the lane owns a_x, a_y < 2, the
// lanes at i = 7 / j = 7 also the brick's upper ring
__device__ __forceinline__ void brick_put_plane(double *ot_c, const int pl, const double *v /* [9] */, const int i, const int j) {
#pragma unroll
  for (int ay = 0; ay < 3; ++ay)
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
      if ((ax < 2 || i == 7) && (ay < 2 || j == 7)) ot_c[pl * BRICK_PLANE + (2 * j + ay) * BRICK_LAT + 2 * i + ax] = v[ax + 3 * ay];
}

template <int OP, bool COMP, bool PERQ, bool GEN>
__global__ void __launch_bounds__(BRICK_THREADS, 2)
k_brick(const CellArgs A, const BrickArgs B, const EpiArgs E) {
  constexpr int NV = 27, NP = 8;
  constexpr bool HAS_U = (OP != KOP_BT);
  __shared__ double xch[2 * BrickXch::TILE];          // gradient exchange between the three component waves
  __shared__ double nbt[3 * NB_TILE];                 // neighbour hand-over, one tile per wave
  __shared__ double eta_t[PERQ ? 27 * 64 : 1];        // per-q viscosities of the layer [q][lane]
  __shared__ double src_t[HAS_U ? 6 * SRC_PLANE : 1]; // source vector on the layer's two upper planes [comp][plane]
  __shared__ uint32_t word_t[3 * SRC_PLANE];          // node words of three lattice planes, plane K in slot K % 3
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 7, j = lane >> 3;
  const int brick = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  const uint32_t *const gn = B.gnode + (size_t)brick * B.latn;
  double *const T = nbt + c * NB_TILE;
  const size_t cs = (size_t)A.cstride;
  BrickXch ex{xch, lane, 0, 0};
  const int nplanes_b = 2 * B.Z + 1;
  const size_t pbase = (size_t)brick * B.npb;
  double carry[9], ub[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) { carry[m] = 0.0; ub[m] = 0.0; }

  // ---- staging: every thread moves its share of a layer's inputs from global memory to LDS.
  // Source vector: one thread per lattice node (all components: 24 contiguous bytes), constrained
  // entries zeroed; viscosity: wave c moves the q rows c, c + 3, ...
  double sv[4][3];
  uint32_t sw[4];
  double ev[PERQ ? 9 : 1];
  auto src_load = [&](const int K0, const int nplanes) {   // planes K0 .. K0 + nplanes - 1 (nplanes <= 2)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int n = tid + BRICK_THREADS * m;
      sw[m] = (n < BRICK_PLANE * nplanes) ? gn[BRICK_PLANE * K0 + n] : 0u;
    }
    if (HAS_U) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double *sp = A.src_u + (size_t)(sw[m] & BRICK_NODE_MASK) * 3;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) sv[m][cc] = sp[cc];
      }
    }
  };
  auto src_write = [&](const int K0, const int nplanes, const int first_slot) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int n = tid + BRICK_THREADS * m;
      if (n < BRICK_PLANE * nplanes) {
        const int pl = n >= BRICK_PLANE ? 1 : 0, r = n - BRICK_PLANE * pl;
        const int J = r / BRICK_LAT, I = r - BRICK_LAT * J;
        const int pos = src_pos(J) * SRC_ROW + src_pos(I);
        if (HAS_U) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            src_t[cc * 2 * SRC_PLANE + (first_slot + pl) * SRC_PLANE + pos] = ((sw[m] >> (29 + cc)) & 1u) ? 0.0 : sv[m][cc];
        }
        // word for the store phase: a row another brick owns carries its surface index (the
        // enumeration of BrickTables::pidx: plane 0, the rings of the planes between, the last plane)
        const int K = K0 + pl;
        uint32_t w = sw[m];
        if (w & BRICK_PARTIAL) {
          const int ring = J == 0 ? I : (J == BRICK_LAT - 1 ? 47 + I : 17 + 2 * (J - 1) + (I == BRICK_LAT - 1 ? 1 : 0));
          const int sidx = K == 0 ? r : (K == nplanes_b - 1 ? BRICK_PLANE + 64 * (nplanes_b - 2) + r : BRICK_PLANE + 64 * (K - 1) + ring);
          w = (w & 0xF0000000u) | (uint32_t)sidx;
        }
        word_t[(K % 3) * SRC_PLANE + pos] = w;
      }
    }
  };
  auto eta_load = [&](const int cell) {
    if (PERQ) {
#pragma unroll
      for (int r = 0; r < 9; ++r) ev[r] = A.etaT[(size_t)(c + 3 * r) * cs + cell];
    }
  };
  auto eta_write = [&]() {
    if (PERQ) {
#pragma unroll
      for (int r = 0; r < 9; ++r) eta_t[(c + 3 * r) * 64 + lane] = ev[r];
    }
  };
  int cell = (int)B.bcell[((size_t)brick * B.Z) * 64 + lane];
  // plane 0 goes through the upper slot into the registers that carry a layer's top plane
  src_load(0, 1);
  src_write(0, 1, 1);
  if (HAS_U) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      const int ax = m % 3, ay = m / 3;
      ub[m] = src_t[(c * 2 + 1) * SRC_PLANE + (j + (ay == 1 ? 9 : ay >> 1)) * SRC_ROW + i + (ax == 1 ? 9 : ax >> 1)];
    }
    __syncthreads();
  }
  src_load(1, 2);
  eta_load(cell);
  src_write(1, 2, 0);
  eta_write();
  __syncthreads();

#if AGMG_ABLATE & 2048
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
  for (int k = 0; k < B.Z; ++k) {
    BRICK_STAMP(0);
    const bool more = k + 1 < B.Z;
    const int cell_next = more ? (int)B.bcell[((size_t)brick * B.Z + k + 1) * 64 + lane] : cell;
    double u[NV], G[NV], pout[NP];
    if (HAS_U) {
#pragma unroll
      for (int a = 0; a < NV; ++a) {
        const int ax = a % 3, ay = (a / 3) % 3, az = a / 9;
        u[a] = az == 0 ? ub[a]
                       : src_t[(c * 2 + az - 1) * SRC_PLANE + (j + (ay == 1 ? 9 : ay >> 1)) * SRC_ROW + i + (ax == 1 ? 9 : ax >> 1)];
      }
    }
    // (the carried plane waits in this wave's neighbour tile, unused until the hand-over below, while
    // the cell operation needs the registers)
#pragma unroll
    for (int m = 0; m < 9; ++m) T[m * 64 + lane] = carry[m];
    BRICK_STAMP(1);
    cell_compute<3, OP, COMP, PERQ, PERQ, false, GEN, false, BrickXch, BRICK_PIPE>(A, cell, c, lane, lane, ex, G, pout, eta_t, u, 64);
    BRICK_STAMP(2);
#pragma unroll
    for (int m = 0; m < 9; ++m) carry[m] = T[m * 64 + lane];
    if (HAS_U) {   // the layer's upper plane is the next layer's lower plane: out of the tile before it is restaged
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int ax = m % 3, ay = m / 3;
        ub[m] = src_t[(c * 2 + 1) * SRC_PLANE + (j + (ay == 1 ? 9 : ay >> 1)) * SRC_ROW + i + (ax == 1 ? 9 : ax >> 1)];
      }
    }

    // ---- sums over the cells of the layer that share a node: the a_x = 2 face goes to lane + 1, then
    // the a_y = 2 face (with what it just received) to lane + 8; wave-private tile, in-order LDS
#pragma unroll
    for (int m = 0; m < 9; ++m) T[m * 64 + lane] = G[2 + 3 * m];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const int from = lane > 0 ? lane - 1 : 0;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const double t = T[m * 64 + from];
        G[3 * m] += (i > 0) ? t : 0.0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int az = 0; az < 3; ++az)
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) T[(az * 3 + ax) * 64 + lane] = G[ax + 6 + 9 * az];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const int from = lane >= 8 ? lane - 8 : 0;
#pragma unroll
      for (int az = 0; az < 3; ++az)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const double t = T[(az * 3 + ax) * 64 + from];
          G[ax + 9 * az] += (j > 0) ? t : 0.0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the lower plane is complete with the previous layer's upper plane; the upper plane waits
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      G[m] += carry[m];
      carry[m] = G[18 + m];
    }
    BRICK_STAMP(3);
    // finished planes -> out tile (this wave's neighbour tile, free again) -> node-wise write-out
    brick_put_plane(T, 0, G, i, j);
    brick_put_plane(T, 1, G + 9, i, j);
    __syncthreads();
    BRICK_STAMP(4);
    brick_writeout(nbt, word_t, 2 * k, 2 * BRICK_PLANE, tid, pbase, B, E);
    BRICK_STAMP(5);
    // the next layer's inputs (in registers only from here to the tile writes below)
    if (more) {
      src_load(2 * k + 3, 2);
      eta_load(cell_next);
    }

    if (OP == KOP_STOKES) {
      // pressure rows: the three component waves' shares summed (c = 0, 1, 2) into the cell slots;
      // the tile the next line would write is free (its last readers passed this line's barrier)
      double *const PP = xch + ex.wpar * BrickXch::TILE;
#pragma unroll
      for (int b = 0; b < NP; ++b) PP[(c * NP + b) * 64 + lane] = pout[b];
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NP; ++b)
        if (b % 3 == c) A.slots_p[(size_t)b * cs + cell] = (PP[b * 64 + lane] + PP[(NP + b) * 64 + lane]) + PP[(2 * NP + b) * 64 + lane];
    }
    if (more) {
      __syncthreads();           // every wave is done with this layer's tiles
      BRICK_STAMP(6);
      src_write(2 * k + 3, 2, 0);
      eta_write();
      __syncthreads();
      BRICK_STAMP(7);
    } else if (OP == KOP_STOKES) {
      __syncthreads();           // the pressure tile is read before the next line overwrites it
    }
    cell = cell_next;
  }
  __syncthreads();   // (the last layer's write-out is done with the out tile)
  brick_put_plane(T, 0, carry, i, j);
  __syncthreads();
  brick_writeout(nbt, word_t, 2 * B.Z, BRICK_PLANE, tid, pbase, B, E);
#if AGMG_ABLATE & 2048
  if (lane == 0)
    for (int q = 0; q < 8; ++q) B.dbg[((size_t)brick * 3 + c) * 8 + q] = st_acc[q];
#endif
}

// ---- fix-up: dst[owner row] += coefficient * (partial sums of the other bricks), fixed order
struct BrickFixArgs {
  const uint32_t *fx_nodes, *fx_ptr, *fx_list;
  const double *partials;
  int n_fix;
  EpiArgs e;
};
__device__ __forceinline__ void brick_fix_body(const BrickFixArgs &F, const unsigned block) {
  const int n = (int)block * 256 + threadIdx.x;
  if (n >= F.n_fix) return;
  const uint32_t nd = F.fx_nodes[n];
  const uint32_t s = F.fx_ptr[n], en = F.fx_ptr[n + 1];
  double sum[3] = {0.0, 0.0, 0.0};
  for (uint32_t k = s; k < en; ++k) {
    const double *p = F.partials + (size_t)F.fx_list[k] * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) sum[c] += p[c];
  }
  const size_t g0 = (size_t)(nd & NODE_MASK) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if ((nd >> (29 + c)) & 1u) continue;   // identity row: independent of the sum
    const size_t t = g0 + c;
    double coef = 1.0;
    if (F.e.epi == EPI_RESID) coef = -1.0;
    else if (F.e.epi == EPI_CHEB) coef = -F.e.f2 * F.e.dinv[t];
    F.e.dst[t] = fma(coef, sum[c], F.e.dst[t]);
  }
}
__global__ void __launch_bounds__(256) k_brick_fix(const BrickFixArgs F) { brick_fix_body(F, blockIdx.x); }
__global__ void __launch_bounds__(256) k_brick_fix_pull(const BrickFixArgs F, const PullArgs P, const unsigned n_fix_blocks) {
  if (blockIdx.x < n_fix_blocks) brick_fix_body(F, blockIdx.x);
  else pull_body<1, 1>(P, blockIdx.x - n_fix_blocks);
}

template <int OP, bool GEN>
void launch_brick_t(hipStream_t s, const CellArgs &a, const BrickArgs &b, const EpiArgs &e, int n_bricks, bool comp, bool perq) {
  const dim3 grid(n_bricks), block(BRICK_THREADS);
  if (comp) {
    if (perq) hipLaunchKernelGGL((k_brick<OP, true, true, GEN>), grid, block, 0, s, a, b, e);
    else hipLaunchKernelGGL((k_brick<OP, true, false, GEN>), grid, block, 0, s, a, b, e);
  } else {
    if (perq) hipLaunchKernelGGL((k_brick<OP, false, true, GEN>), grid, block, 0, s, a, b, e);
    else hipLaunchKernelGGL((k_brick<OP, false, false, GEN>), grid, block, 0, s, a, b, e);
  }
}
template <bool GEN>
void launch_brick_op(hipStream_t s, const CellArgs &a, const BrickArgs &b, const EpiArgs &e, int n_bricks, int which, bool comp, bool perq) {
  if (which == AGMG_OP_A) launch_brick_t<KOP_A, GEN>(s, a, b, e, n_bricks, comp, perq);
  else if (which == AGMG_OP_STOKES) launch_brick_t<KOP_STOKES, GEN>(s, a, b, e, n_bricks, comp, perq);
  else hipLaunchKernelGGL((k_brick<KOP_BT, false, false, GEN>), dim3(n_bricks), dim3(BRICK_THREADS), 0, s, a, b, e);
}

}  // namespace

int launch_op_bricks(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
                     const double *src_p, double *partials, double *slots_p, const EpiArgs &eu, const EpiArgs *ep) {
  const BrickTables &bk = L.bk;
  if (!bk.built || L.dim != 3) { set_error("launch_op_bricks: level has no brick tables"); return -1; }
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (eu.epi != EPI_STORE && eu.epi != EPI_RESID && eu.epi != EPI_CHEB) { set_error("launch_op_bricks: epilogue not supported"); return -1; }
  if (ep && which != AGMG_OP_STOKES) { set_error("launch_op_bricks: pressure epilogue misuse"); return -1; }
  const CellArgs a = make_cell_args(L, cd, src_u, src_p, nullptr, slots_p);
  static unsigned long long *dbg_buf = nullptr;
#if AGMG_ABLATE & 2048
  if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, (size_t)1 << 26);
#endif
  const BrickArgs b{bk.gnode, bk.bcell, bk.pidx, partials, bk.Z, bk.latn, bk.npb, dbg_buf};
  const bool perq = cd.nqv > 1;
  const bool comp = (cd.is_compressible != 0 || cd.dilation) && which != AGMG_OP_BT;
  static const bool use_q = !(getenv("AGMG_BRICKQ") && atoi(getenv("AGMG_BRICKQ")) == 0);
  if (use_q && brickq_supported(L, cd, which)) {
    const int rc = launch_brickq(s, L, cd, which, src_u, src_p, partials, slots_p, eu);
    if (rc) return rc;
  } else if (L.geom == AGMG_GEOM_GENERAL) launch_brick_op<true>(s, a, b, eu, bk.n_bricks, which, comp, perq);
  else launch_brick_op<false>(s, a, b, eu, bk.n_bricks, which, comp, perq);
  AGMG_HIP_CHECK(hipGetLastError());
#if AGMG_ABLATE & 2048
  {
    (void)hipDeviceSynchronize();
    const size_t nw = (size_t)bk.n_bricks * 3;
    std::vector<unsigned long long> h(nw * 8);
    (void)hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost);
    double d[8] = {0};
    for (size_t w = 0; w < nw; ++w)
      for (int q = 0; q < 8; ++q) d[q] += (double)h[w * 8 + q];
    fprintf(stderr, "brick stamps (mean cycles per wave and layer, Z=%d): loop head %.0f | u from tiles %.0f | cell_compute %.0f | hand-over %.0f | next loads issue %.0f | stores %.0f | barrier %.0f | tile writes+barrier %.0f\n",
            bk.Z, d[0] / nw / bk.Z, d[1] / nw / bk.Z, d[2] / nw / bk.Z, d[3] / nw / bk.Z, d[4] / nw / bk.Z, d[5] / nw / bk.Z, d[6] / nw / bk.Z, d[7] / nw / bk.Z);
  }
#endif
  const BrickFixArgs f{bk.fx_nodes, bk.fx_ptr, bk.fx_list, partials, bk.n_fix, eu};
  const unsigned nf = (unsigned)((bk.n_fix + 255) / 256);
  if (!ep) {
    if (nf) hipLaunchKernelGGL(k_brick_fix, dim3(nf), dim3(256), 0, s, f);
  } else {
    const PullArgs p{L.pull_ptr_p, L.pull_list_p, slots_p, L.cmask_p, 0, L.n_nodes_p, 0xFFFFFFFFu, nullptr, nullptr, *ep};
    const unsigned np = (unsigned)((L.n_nodes_p + 255) / 256);
    hipLaunchKernelGGL(k_brick_fix_pull, dim3(nf + np), dim3(256), 0, s, f, p, nf);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace agmg
