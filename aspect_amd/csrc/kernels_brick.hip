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

namespace agmg {

namespace {

struct BrickArgs {
  const uint32_t *gnode, *bcell;
  const uint16_t *pidx;
  double *partials;
  int Z, latn, npb;
};

constexpr int BRICK_THREADS = 192;
constexpr int NB_TILE = 9 * 64;   // doubles of the neighbour tile of one wave

// row `t`-th component of node word w: final (epilogue + store) or partial (this brick's share)
__device__ __forceinline__ void brick_store(const uint32_t w, const int lat, const double sum, const int c,
                                            const int brick, const BrickArgs &B, const EpiArgs &E) {
  if (w & BRICK_PARTIAL) {
    const size_t slot = (size_t)brick * B.npb + B.pidx[lat];
    B.partials[slot * 3 + c] = sum;
    return;
  }
  const size_t t = (size_t)(w & BRICK_NODE_MASK) * 3 + c;
  const bool con = (w >> (29 + c)) & 1u;
  double out;
  if (E.epi == EPI_STORE) {
    out = con ? (E.x ? E.x[t] : 0.0) : sum;
  } else if (E.epi == EPI_RESID) {
    out = E.b[t] - (con ? (E.x ? E.x[t] : 0.0) : sum);
  } else {  // EPI_CHEB
    const double x = E.x[t];
    double v = fma(E.f2 * E.dinv[t], E.b[t] - (con ? x : sum), (1.0 + E.f1) * x);
    if (E.xold) v = fma(-E.f1, E.xold[t], v);
    out = v;
  }
  E.dst[t] = out;
}

// the 3 x 3 nodes (a_x, a_y) of one lattice plane held by this lane: the lane owns a_x, a_y < 2, the
// lanes at i = 7 / j = 7 also the brick's upper ring
__device__ __forceinline__ void brick_store_plane(const uint32_t *gn, const int lat0, const double *v /* [9], stride 1 */,
                                                  const int i, const int j, const int c, const int brick,
                                                  const BrickArgs &B, const EpiArgs &E) {
  uint32_t w[9];
#pragma unroll
  for (int ay = 0; ay < 3; ++ay)
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const bool own = (ax < 2 || i == 7) && (ay < 2 || j == 7);
      w[ax + 3 * ay] = own ? gn[lat0 + ax + BRICK_LAT * ay] : 0u;
    }
#pragma unroll
  for (int ay = 0; ay < 3; ++ay)
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const bool own = (ax < 2 || i == 7) && (ay < 2 || j == 7);
      if (own) brick_store(w[ax + 3 * ay], lat0 + ax + BRICK_LAT * ay, v[ax + 3 * ay], c, brick, B, E);
    }
}

template <int OP, bool COMP, bool PERQ, bool GEN>
__global__ void __launch_bounds__(BRICK_THREADS, 2)
k_brick(const CellArgs A, const BrickArgs B, const EpiArgs E) {
  constexpr int NV = 27, NP = 8;
  constexpr bool HAS_U = (OP != KOP_BT);
  __shared__ double xch[2 * BrickXch::TILE];   // gradient exchange between the three component waves
  __shared__ double nbt[3 * NB_TILE];          // neighbour hand-over, one tile per wave
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 7, j = lane >> 3;
  const int brick = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  const uint32_t *const gn = B.gnode + (size_t)brick * B.latn;
  const int l0 = 2 * i + BRICK_LAT * 2 * j;    // lattice index of the lane's lower corner in plane 0
  double *const T = nbt + c * NB_TILE;
  const size_t cs = (size_t)A.cstride;
  BrickXch ex{xch, lane, 0};
  double carry[9], ub[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) { carry[m] = 0.0; ub[m] = 0.0; }

#pragma unroll 1
  for (int k = 0; k < B.Z; ++k) {
    const int cell = (int)B.bcell[((size_t)brick * B.Z + k) * 64 + lane];
    const int lk = l0 + BRICK_PLANE * 2 * k;
    double u[NV], G[NV], pout[NP];
    if (HAS_U) {
      // gather (constrained DoFs read as zero); the lower plane is the previous layer's upper plane
      uint32_t w[NV];
#pragma unroll
      for (int a = 0; a < NV; ++a) {
        const int ax = a % 3, ay = (a / 3) % 3, az = a / 9;
        w[a] = (az == 0 && k > 0) ? 0u : gn[lk + ax + BRICK_LAT * ay + BRICK_PLANE * az];
      }
#pragma unroll
      for (int a = 0; a < NV; ++a) {
        if (a < 9 && k > 0) {
          u[a] = ub[a];
        } else {
          const double v = A.src_u[(size_t)(w[a] & BRICK_NODE_MASK) * 3 + c];
          u[a] = ((w[a] >> (29 + c)) & 1u) ? 0.0 : v;
        }
      }
#pragma unroll
      for (int m = 0; m < 9; ++m) ub[m] = u[18 + m];
    }
    cell_compute<3, OP, COMP, PERQ, false, false, GEN, false, BrickXch>(A, cell, c, lane, lane, ex, G, pout, nullptr, u);

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
    brick_store_plane(gn, lk, G, i, j, c, brick, B, E);
    brick_store_plane(gn, lk + BRICK_PLANE, G + 9, i, j, c, brick, B, E);

    if (OP == KOP_STOKES) {
      // pressure rows: the three component waves' shares summed (c = 0, 1, 2) into the cell slots;
      // the tile the next line would write is free (its last readers passed this line's barrier)
      double *const PP = xch + ex.parity * BrickXch::TILE;
#pragma unroll
      for (int b = 0; b < NP; ++b) PP[(c * NP + b) * 64 + lane] = pout[b];
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NP; ++b)
        if (b % 3 == c) A.slots_p[(size_t)b * cs + cell] = (PP[b * 64 + lane] + PP[(NP + b) * 64 + lane]) + PP[(2 * NP + b) * 64 + lane];
      __syncthreads();
    }
  }
  brick_store_plane(gn, l0 + BRICK_PLANE * 2 * B.Z, carry, i, j, c, brick, B, E);
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
  const BrickArgs b{bk.gnode, bk.bcell, bk.pidx, partials, bk.Z, bk.latn, bk.npb};
  const bool perq = cd.nqv > 1;
  const bool comp = (cd.is_compressible != 0 || cd.dilation) && which != AGMG_OP_BT;
  if (L.geom == AGMG_GEOM_GENERAL) launch_brick_op<true>(s, a, b, eu, bk.n_bricks, which, comp, perq);
  else launch_brick_op<false>(s, a, b, eu, bk.n_bricks, which, comp, perq);
  AGMG_HIP_CHECK(hipGetLastError());
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
