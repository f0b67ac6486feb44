// kernels_transfer.hip -- matrix-free multigrid transfer for Q2^dim / Q1 on uniformly refined
// parents (gfx950).  Replaces deal.II MGTransferMF::prolongate_and_add / restrict_and_add as
// wired by stokes_matrix_free_local_smoothing.cc:1871-1877: 1-D embedding
// rows at x = 0, 1/4, 1/2, 3/4, 1 (Q2) or 0, 1/2, 1 (Q1), tensor product per component,
// coarse Dirichlet entries zeroed before prolongation and dropped after restriction.
//
// Prolongation is node-owner ("pull"): one thread per fine DoF interpolates from ONE parent
// cell that contains the node (the FE space is conforming, so every parent gives the same
// value).  Restriction is its exact transpose, organised per (parent cell, coarse local DoF);
// each fine value is weighted by 1/multiplicity over the parent patches so that the sum over
// parents counts it once; contributions go to the coarse level's slot array and are summed by
// the same deterministic pull kernel as the operators (EPI_ADD).
#include "agmg_internal.h"

namespace agmg {
namespace {

// Q2 embedding rows P2[r][i] = phi_i(r/4); Q1 rows P1[r][i] = psi_i(r/2)
__host__ __device__ constexpr double p2row(int r, int i) {
  return r == 0 ? (i == 0 ? 1.0 : 0.0)
       : r == 1 ? (i == 0 ? 0.375 : (i == 1 ? 0.75 : -0.125))
       : r == 2 ? (i == 1 ? 1.0 : 0.0)
       : r == 3 ? (i == 0 ? -0.125 : (i == 1 ? 0.75 : 0.375))
                : (i == 2 ? 1.0 : 0.0);
}
__host__ __device__ constexpr double p1row(int r, int i) {
  return r == 0 ? (i == 0 ? 1.0 : 0.0) : r == 1 ? 0.5 : (i == 1 ? 1.0 : 0.0);
}
template <int K> __device__ __forceinline__ double prow(int r, int i) {
  if (K == 2) {
    const double t[5][3] = {{1, 0, 0}, {0.375, 0.75, -0.125}, {0, 1, 0}, {-0.125, 0.75, 0.375}, {0, 0, 1}};
    return t[r][i];
  } else {
    const double t[3][2] = {{1, 0}, {0.5, 0.5}, {0, 1}};
    return t[r][i];
  }
}

template <int K> __host__ __device__ constexpr double pw(int r, int i) { return K == 2 ? p2row(r, i) : p1row(r, i); }

struct ProlArgs {
  int n_fine_nodes;
  const uint32_t *parent;   // [n_fine_nodes]
  const uint8_t *pos;       // [n_fine_nodes] position in the parent's (2K+1)^DIM patch
  const uint32_t *cnT;      // coarse [ (K+1)^DIM ][cstride_c]
  int cstride_c;
  const double *src;
  double *dst;
};

// K: polynomial degree (2 velocity, 1 pressure); NC: components per node.  One thread per fine
// node: the parent's node indices and the weights serve all NC components.
template <int DIM, int K, int NC>
__global__ void __launch_bounds__(256) k_prolongate_add(const ProlArgs A) {
  constexpr int N1 = K + 1, NF = 2 * K + 1, NPL = N1 * N1;
  const int node = blockIdx.x * 256 + threadIdx.x;
  if (node >= A.n_fine_nodes) return;
  const uint32_t par = A.parent[node];
  int pos = A.pos[node];
  int r[3] = {0, 0, 0};
  for (int d = 0; d < DIM; ++d) { r[d] = pos % NF; pos /= NF; }
  // weights per direction (1 or K+1 of them are non-zero)
  double wd[3][N1];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int i = 0; i < N1; ++i) wd[d][i] = d < DIM ? prow<K>(r[d], i) : (i == 0 ? 1.0 : 0.0);
  double val[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) val[c] = 0.0;
  // plane by plane of the parent's nodes: all index loads of a plane are issued together, then all value loads
  // (the first version walked the up to 27 parents one after the other: 2 x 27 dependent loads, 16 us on a
  // level of one workgroup), summation order unchanged
#pragma unroll
  for (int i2 = 0; i2 < (DIM == 3 ? N1 : 1); ++i2) {
    const double w2 = wd[2][i2];
    if (w2 == 0.0) continue;                         // (uniform per plane of fine nodes in most wavefronts)
    double w[NPL];
    uint32_t idx[NPL];
#pragma unroll
    for (int a = 0; a < NPL; ++a) {
      w[a] = w2 * wd[1][a / N1] * wd[0][a % N1];
      idx[a] = w[a] != 0.0 ? A.cnT[(size_t)(a + NPL * i2) * A.cstride_c + par] : 0u;
    }
    double v[NPL][NC];
#pragma unroll
    for (int a = 0; a < NPL; ++a) {
      const double *sp = A.src + (size_t)(idx[a] & NODE_MASK) * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c)   // coarse Dirichlet values read as zero
        v[a][c] = (w[a] != 0.0 && !((idx[a] >> (29 + c)) & 1u)) ? sp[c] : 0.0;
    }
    // same order as the nested loops over i1, i0 with their zero weights skipped
#pragma unroll
    for (int a = 0; a < NPL; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (w[a] != 0.0 && !((idx[a] >> (29 + c)) & 1u)) val[c] = fma(w[a], v[a][c], val[c]);
  }
  double *dp = A.dst + (size_t)node * NC;
#pragma unroll
  for (int c = 0; c < NC; ++c) dp[c] += val[c];
}

// Velocity prolongation on levels with many cells, the transpose of k_restrict_cells: one lane per
// (parent cell, component) expands the parent's (K+1)^DIM values to the (2K+1)^DIM patch by sum
// factorisation, plane by plane, and adds them to the fine nodes this parent is the designated
// parent of (`parent[fine node] == cell` at the designated patch position: every fine node is
// written exactly once, no atomics).
// Uniform control flow, where the node-owner form diverges on the 1..27 weights per node.
struct ProlCellArgs {
  int n_coarse_cells, cstride_c, cstride_f;
  const uint32_t *childrenT;  // [2^DIM][cstride_c]
  const uint32_t *cnfT;       // fine  [(K+1)^DIM][cstride_f]
  const uint32_t *cncT;       // coarse [(K+1)^DIM][cstride_c]
  const uint32_t *parent;     // [n_fine_nodes] designated parent cell of each fine node
  const uint8_t *pos;         // [n_fine_nodes] and its position in that parent's patch (a periodic
                              // direction one or two cells wide puts a node into a patch twice)
  const uint32_t *mask;       // [4][cstride_c]: the same designation per parent, one bit per patch position (no load
                              // of parent[] / pos[] / the fine index for the positions another parent writes)
  const double *src;          // coarse vector
  double *dst;                // fine vector
};
template <int DIM, int K, int NC>
__global__ void __launch_bounds__(256) k_prolongate_cells(const ProlCellArgs A) {
  constexpr int N1 = K + 1, NF = 2 * K + 1, NLOC = (DIM == 3) ? N1 * N1 * N1 : N1 * N1, NCH = 1 << DIM;
  constexpr int NZ = (DIM == 3) ? NF : 1, NZC = (DIM == 3) ? N1 : 1;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)A.n_coarse_cells * NC) return;
  const int cell = (int)(t / NC), c = (int)(t - (long)cell * NC);
  uint32_t ch[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) ch[k] = A.childrenT[(size_t)k * A.cstride_c + cell];
  uint32_t own[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) own[k] = A.mask[(size_t)k * A.cstride_c + cell];
  double cv[NLOC];
#pragma unroll
  for (int a = 0; a < NLOC; ++a) {
    const uint32_t idx = A.cncT[(size_t)a * A.cstride_c + cell];
    const double v = A.src[(size_t)(idx & NODE_MASK) * NC + c];
    cv[a] = ((idx >> (29 + c)) & 1u) ? 0.0 : v;   // coarse Dirichlet values read as zero
  }
#pragma unroll
  for (int r2 = 0; r2 < NZ; ++r2) {
    const int up2 = (DIM == 3 && r2 > K) ? 1 : 0;
    double pl[N1][N1];   // the coarse values interpolated to fine plane r2
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
      for (int i0 = 0; i0 < N1; ++i0) {
        double s = 0.0;
#pragma unroll
        for (int i2 = 0; i2 < NZC; ++i2) {
          const double w = (DIM == 3) ? pw<K>(r2, i2) : 1.0;
          if (w != 0.0) s = fma(w, cv[i0 + N1 * (i1 + N1 * i2)], s);
        }
        pl[i1][i0] = s;
      }
    double tx[N1][NF];
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
      for (int r0 = 0; r0 < NF; ++r0) {
        double s = 0.0;
#pragma unroll
        for (int i0 = 0; i0 < N1; ++i0)
          if (pw<K>(r0, i0) != 0.0) s = fma(pw<K>(r0, i0), pl[i1][i0], s);
        tx[i1][r0] = s;
      }
#pragma unroll
    for (int r1 = 0; r1 < NF; ++r1)
#pragma unroll
      for (int r0 = 0; r0 < NF; ++r0) {
        double s = 0.0;
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
          if (pw<K>(r1, i1) != 0.0) s = fma(pw<K>(r1, i1), tx[i1][r0], s);
        const int up0 = r0 > K ? 1 : 0, up1 = r1 > K ? 1 : 0;
        const int child = up0 | (up1 << 1) | (up2 << 2);
        const int loc = (r0 - K * up0) + N1 * (r1 - K * up1) + (DIM == 3 ? N1 * N1 * (r2 - K * up2) : 0);
        const int p = r0 + NF * (r1 + NF * r2);
        if ((own[p >> 5] >> (p & 31)) & 1u) {
          const uint32_t fn = A.cnfT[(size_t)loc * A.cstride_f + ch[child]] & NODE_MASK;
          A.dst[(size_t)fn * NC + c] += s;
        }
      }
  }
}

struct RestrArgs {
  int n_coarse_cells, cstride_c, cstride_f;
  const uint32_t *childrenT;  // [2^DIM][cstride_c]
  const uint32_t *cnfT;       // fine [(K+1)^DIM][cstride_f]
  const double *inv_mult;     // [n_fine_nodes]
  const double *src;          // fine vector
  double *slots;              // coarse slots [a][cstride_c][NC]
};

// One lane per (parent cell, component): the (2K+1)^DIM patch values of the children are gathered
// once (weighted by 1/multiplicity), then restricted by sum factorisation, (2K+1) -> (K+1) per
// direction, plane by plane along the last axis.  (The first version looped over the patch once
// per coarse local DoF: 27 x 27 dependent gathers per cell and component instead of 125.)
template <int DIM, int K, int NC>
__global__ void __launch_bounds__(256) k_restrict_cells(const RestrArgs A) {
  constexpr int N1 = K + 1, NF = 2 * K + 1, NLOC = (DIM == 3) ? N1 * N1 * N1 : N1 * N1, NCH = 1 << DIM;
  constexpr int NZ = (DIM == 3) ? NF : 1, NZC = (DIM == 3) ? N1 : 1;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)A.n_coarse_cells * NC) return;
  const int cell = (int)(t / NC), c = (int)(t - (long)cell * NC);
  uint32_t ch[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) ch[k] = A.childrenT[(size_t)k * A.cstride_c + cell];
  double acc[NLOC];
#pragma unroll
  for (int a = 0; a < NLOC; ++a) acc[a] = 0.0;
#pragma unroll
  for (int r2 = 0; r2 < NZ; ++r2) {
    const int up2 = (DIM == 3 && r2 > K) ? 1 : 0;
    // the plane's values: fine node at patch position (r0, r1, r2) lives in the child that is the
    // upper half wherever r > K
    double v[NF][NF];
#pragma unroll
    for (int r1 = 0; r1 < NF; ++r1)
#pragma unroll
      for (int r0 = 0; r0 < NF; ++r0) {
        const int up0 = r0 > K ? 1 : 0, up1 = r1 > K ? 1 : 0;
        const int child = up0 | (up1 << 1) | (up2 << 2);
        const int loc = (r0 - K * up0) + N1 * (r1 - K * up1) + (DIM == 3 ? N1 * N1 * (r2 - K * up2) : 0);
        const uint32_t fn = A.cnfT[(size_t)loc * A.cstride_f + ch[child]] & NODE_MASK;
        v[r1][r0] = A.inv_mult[fn] * A.src[(size_t)fn * NC + c];
      }
    double tx[NF][N1];
#pragma unroll
    for (int r1 = 0; r1 < NF; ++r1)
#pragma unroll
      for (int i0 = 0; i0 < N1; ++i0) {
        double s = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < NF; ++r0)
          if (pw<K>(r0, i0) != 0.0) s = fma(pw<K>(r0, i0), v[r1][r0], s);
        tx[r1][i0] = s;
      }
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
      for (int i0 = 0; i0 < N1; ++i0) {
        double s = 0.0;
#pragma unroll
        for (int r1 = 0; r1 < NF; ++r1)
          if (pw<K>(r1, i1) != 0.0) s = fma(pw<K>(r1, i1), tx[r1][i0], s);
#pragma unroll
        for (int i2 = 0; i2 < NZC; ++i2) {
          const double w = (DIM == 3) ? pw<K>(r2, i2) : 1.0;
          if (w != 0.0) acc[i0 + N1 * (i1 + N1 * i2)] = fma(w, s, acc[i0 + N1 * (i1 + N1 * i2)]);
        }
      }
  }
#pragma unroll
  for (int a = 0; a < NLOC; ++a) A.slots[((size_t)a * A.cstride_c + cell) * NC + c] = acc[a];
}

// The same restriction with the (2K+1) planes of a parent's patch on separate lanes: lane = (parent cell, component, plane
// r2).  A lane gathers its plane (one index load and one value pair per fine node, all independent) and reduces it to the
// (K+1)^2 in-plane sums; the (K+1) lanes with r2 <= K then combine the planes through LDS, in the plane order of
// k_restrict_cells (bit-identical).  The one-lane form walks the planes one after the other: 2 x (2K+1) dependent gathers,
// 21 us on levels of 6 to 48 workgroups.  3-D only (the 2-D patch is one plane).
template <int K, int NC>
__global__ void __launch_bounds__(256) k_restrict_planes(const RestrArgs A) {
  constexpr int N1 = K + 1, NF = 2 * K + 1, NPL = N1 * N1;
  constexpr int GROUPS = 256 / NF;                       // (cell, component) pairs per workgroup
  __shared__ double sh[GROUPS][NF][NPL];
  const int grp = threadIdx.x / NF, r2 = threadIdx.x - grp * NF;
  const long t = (long)blockIdx.x * GROUPS + grp;
  const bool live = grp < GROUPS && t < (long)A.n_coarse_cells * NC;
  const int cell = live ? (int)(t / NC) : 0, c = live ? (int)(t - (long)cell * NC) : 0;
  if (live) {
    const int up2 = r2 > K ? 1 : 0;
    uint32_t ch[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ch[k] = A.childrenT[(size_t)(k + 4 * up2) * A.cstride_c + cell];
    uint32_t fn[NF][NF];
#pragma unroll
    for (int r1 = 0; r1 < NF; ++r1)
#pragma unroll
      for (int r0 = 0; r0 < NF; ++r0) {
        const int up0 = r0 > K ? 1 : 0, up1 = r1 > K ? 1 : 0;
        const int loc = (r0 - K * up0) + N1 * (r1 - K * up1) + N1 * N1 * (r2 - K * up2);
        fn[r1][r0] = A.cnfT[(size_t)loc * A.cstride_f + ch[up0 | (up1 << 1)]] & NODE_MASK;
      }
    double v[NF][NF];
#pragma unroll
    for (int r1 = 0; r1 < NF; ++r1)
#pragma unroll
      for (int r0 = 0; r0 < NF; ++r0) v[r1][r0] = A.inv_mult[fn[r1][r0]] * A.src[(size_t)fn[r1][r0] * NC + c];
    double tx[NF][N1];
#pragma unroll
    for (int r1 = 0; r1 < NF; ++r1)
#pragma unroll
      for (int i0 = 0; i0 < N1; ++i0) {
        double s = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < NF; ++r0)
          if (pw<K>(r0, i0) != 0.0) s = fma(pw<K>(r0, i0), v[r1][r0], s);
        tx[r1][i0] = s;
      }
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1)
#pragma unroll
      for (int i0 = 0; i0 < N1; ++i0) {
        double s = 0.0;
#pragma unroll
        for (int r1 = 0; r1 < NF; ++r1)
          if (pw<K>(r1, i1) != 0.0) s = fma(pw<K>(r1, i1), tx[r1][i0], s);
        sh[grp][r2][i0 + N1 * i1] = s;
      }
  }
  __syncthreads();
  if (live && r2 < N1) {
    const int i2 = r2;
#pragma unroll
    for (int a = 0; a < NPL; ++a) {
      double acc = 0.0;
      for (int q = 0; q < NF; ++q) {
        const double w = prow<K>(q, i2);
        if (w != 0.0) acc = fma(w, sh[grp][q][a], acc);
      }
      A.slots[((size_t)(a + NPL * i2) * A.cstride_c + cell) * NC + c] = acc;
    }
  }
}

}  // namespace

int launch_prolongate_add(hipStream_t s, const DeviceLevel &fine, const DeviceLevel &coarse, bool velocity,
                          double *dst_fine, const double *src_coarse) {
  if (!fine.has_transfer) { set_error("transfer tables not built for this level"); return -5; }
  ProlArgs a;
  a.n_fine_nodes = velocity ? fine.n_nodes_v : fine.n_nodes_p;
  a.parent = velocity ? fine.prol_parent_v : fine.prol_parent_p;
  a.pos = velocity ? fine.prol_pos_v : fine.prol_pos_p;
  a.cnT = velocity ? coarse.cnvT : coarse.cnpT;
  a.cstride_c = coarse.cstride;
  a.src = src_coarse;
  a.dst = dst_fine;
  const long n = a.n_fine_nodes;
  if (n <= 0) return 0;
  // many cells: the per-parent sum-factorised form (the summation order of a fine value differs from
  // the node-owner form only in rounding)
  // (AGMG_PROL_CELLS_MIN: developer/test knob for the switch-over size, read per call)
  const char *env = getenv("AGMG_PROL_CELLS_MIN");
  const long min_cells = env ? atol(env) : 131072;   // below: too few lanes, the node-owner form wins (r5: 19 vs 100 us)
  if (velocity && coarse.n_cells >= min_cells && fine.prol_mask_v) {
    ProlCellArgs pc;
    pc.n_coarse_cells = coarse.n_cells;
    pc.cstride_c = coarse.cstride;
    pc.cstride_f = fine.cstride;
    pc.childrenT = fine.childrenT;
    pc.cnfT = fine.cnvT;
    pc.cncT = coarse.cnvT;
    pc.parent = fine.prol_parent_v;
    pc.pos = fine.prol_pos_v;
    pc.mask = fine.prol_mask_v;
    pc.src = src_coarse;
    pc.dst = dst_fine;
    const long nl = (long)coarse.n_cells * fine.dim;
    const dim3 g((unsigned)((nl + 255) / 256)), b(256);
    if (fine.dim == 3) hipLaunchKernelGGL((k_prolongate_cells<3, 2, 3>), g, b, 0, s, pc);
    else hipLaunchKernelGGL((k_prolongate_cells<2, 2, 2>), g, b, 0, s, pc);
    AGMG_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (fine.dim == 3) {
    if (velocity) hipLaunchKernelGGL((k_prolongate_add<3, 2, 3>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_prolongate_add<3, 1, 1>), grid, block, 0, s, a);
  } else {
    if (velocity) hipLaunchKernelGGL((k_prolongate_add<2, 2, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_prolongate_add<2, 1, 1>), grid, block, 0, s, a);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_restrict_cells(hipStream_t s, const DeviceLevel &fine, const DeviceLevel &coarse, bool velocity,
                          const double *src_fine, double *slots) {
  if (!fine.has_transfer) { set_error("transfer tables not built for this level"); return -5; }
  RestrArgs a;
  a.n_coarse_cells = coarse.n_cells;
  a.cstride_c = coarse.cstride;
  a.cstride_f = fine.cstride;
  a.childrenT = fine.childrenT;
  a.cnfT = velocity ? fine.cnvT : fine.cnpT;
  a.inv_mult = velocity ? fine.inv_mult_v : fine.inv_mult_p;
  a.src = src_fine;
  a.slots = slots;
  const long n = (long)coarse.n_cells * (velocity ? fine.dim : 1);
  if (n <= 0) return 0;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  // velocity patches in 3-D on levels that do not fill the chip: one lane per plane of the patch
  // (AGMG_RESTRICT_PLANES_MAX: developer/test knob for the switch-over size, read per call)
  const char *env = getenv("AGMG_RESTRICT_PLANES_MAX");
  const long planes_max = env ? atol(env) : 32768;
  if (fine.dim == 3 && velocity && coarse.n_cells <= planes_max) {
    constexpr int GROUPS = 256 / 5;
    const dim3 g2((unsigned)((n + GROUPS - 1) / GROUPS));
    hipLaunchKernelGGL((k_restrict_planes<2, 3>), g2, block, 0, s, a);
    AGMG_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (fine.dim == 3) {
    if (velocity) hipLaunchKernelGGL((k_restrict_cells<3, 2, 3>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_restrict_cells<3, 1, 1>), grid, block, 0, s, a);
  } else {
    if (velocity) hipLaunchKernelGGL((k_restrict_cells<2, 2, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_restrict_cells<2, 1, 1>), grid, block, 0, s, a);
  }
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace agmg
