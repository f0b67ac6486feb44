// kernels_brick.hip -- launcher and fix-up pass of the brick path (gfx950).
//
// What this replaces in the reference (tjhei/aspect 3.1.0-pre), as kernels_ops.hip does:
//   ABlockOperator ::local_apply + the deal.II cell_loop and MatrixFreeOperators::Base::vmult around it
//   (matrix_free_operators.cc:705-795)
//
// A level whose cells tile into bricks of 8 x 8 x Z cells (BrickTables, agmg_internal.h) is applied by the
// quad-per-cell kernel of kernels_brickq.hip, one workgroup per brick.  Rows a brick owns are stored to dst with
// the vector epilogue fused; rows owned by another brick go to a compact partial-sum array (the brick's
// surface), and k_brick_fix adds those shares to the owners' rows afterwards (all epilogues are affine in the
// sum, so the correction is  dst += coefficient * partial sums).
#include <cstdlib>

#include "agmg_internal.h"
#include "kernels_cell.h"

namespace agmg {

namespace {

// ---- fix-up: dst[owner row] += coefficient * (partial sums of the other bricks), fixed order
struct BrickFixArgs {
  const uint32_t *fx_nodes, *fx_ptr, *fx_list;
  const double *partials;
  int n_fix;
  EpiArgs e;
  // distributed levels: list entries >= recv_base address the receive buffer of the ghost exchange
  uint32_t recv_base;
  const double *recv;
};
__device__ __forceinline__ const double *brick_partial(const BrickFixArgs &F, const uint32_t ent) {
  return ent >= F.recv_base ? F.recv + (size_t)(ent - F.recv_base) * 3 : F.partials + (size_t)ent * 3;
}
// One thread per (node, component) that receives partial sums (one thread per node with its three components: 102 us
// at r7 against 77 us: three times as many loads in flight, and the three lanes of a node read its 24 bytes -- row,
// coefficient, partial slots -- side by side).  The loads go out in three dependent rounds instead of one per list
// entry: (row bounds, node word) -> (the first FX_PRE list entries, clamped) + the row itself and the epilogue's
// coefficient -> the partial sums; longer rows finish in a loop.  Same order of additions as the list.
__global__ void __launch_bounds__(256) k_brick_fix(const BrickFixArgs F) {
  constexpr int FX_PRE = 3;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n = (int)(t / 3), c = (int)(t - (size_t)n * 3);
  if (n >= F.n_fix) return;
  const uint32_t nd = F.fx_nodes[n];
  const uint32_t s = F.fx_ptr[n], en = F.fx_ptr[n + 1];
  const size_t g = (size_t)(nd & NODE_MASK) * 3 + c;
  uint32_t ent[FX_PRE];
#pragma unroll
  for (int k = 0; k < FX_PRE; ++k) ent[k] = F.fx_list[min(s + k, en - 1)];
  const double d0 = F.e.dst[g];
  const double coef = F.e.epi == EPI_CHEB ? -F.e.f2 * F.e.dinv[g] : (F.e.epi == EPI_RESID ? -1.0 : 1.0);
  double part[FX_PRE];
#pragma unroll
  for (int k = 0; k < FX_PRE; ++k) part[k] = brick_partial(F, ent[k])[c];
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < FX_PRE; ++k)
    if (s + k < en) sum += part[k];
  for (uint32_t k = s + FX_PRE; k < en; ++k) sum += brick_partial(F, F.fx_list[k])[c];
  if ((nd >> (29 + c)) & 1u) return;   // identity row: independent of the sum
  F.e.dst[g] = fma(coef, sum, d0);
}
// distributed levels: the sum of the partial slots of every ghost node, brick order, into the send buffer of the compress
// step (ghost_out[(node - n_owned) * 3 + c]; the owner adds it to its row in its own fix-up)
__global__ void __launch_bounds__(256) k_brick_ghost(const uint32_t *__restrict__ gh_nodes, const uint32_t *__restrict__ gh_ptr,
                                                     const uint32_t *__restrict__ gh_list, const double *__restrict__ partials,
                                                     double *__restrict__ ghost_out, const int n_gh, const int n_owned) {
  const int n = (int)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_gh) return;
  double sum[3] = {0.0, 0.0, 0.0};
  for (uint32_t k = gh_ptr[n]; k < gh_ptr[n + 1]; ++k) {
    const double *p = partials + (size_t)gh_list[k] * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) sum[c] += p[c];
  }
  double *o = ghost_out + (size_t)(gh_nodes[n] - (uint32_t)n_owned) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = sum[c];
}

}  // namespace

int launch_op_bricks(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
                     const double *src_p, double *partials, double *slots_p, const EpiArgs &eu, const EpiArgs *ep, int phase) {
  const BrickTables &bk = L.bk;
  if (!bk.built || L.dim != 3) { set_error("launch_op_bricks: level has no brick tables"); return -1; }
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (eu.epi != EPI_STORE && eu.epi != EPI_RESID && eu.epi != EPI_CHEB) { set_error("launch_op_bricks: epilogue not supported"); return -1; }
  if (ep) { set_error("launch_op_bricks: the brick path has no pressure rows yet"); return -1; }
  if (!brickq_supported(L, cd, which)) { set_error("launch_op_bricks: operator not supported on the brick path"); return -1; }
  if (phase != PH_OWNED) {
    const int b0 = phase == PH_INTERFACE ? bk.n_interior : 0;
    const int nb = phase == PH_INTERIOR ? bk.n_interior : bk.n_bricks - b0;
    const int rc = launch_brickq(s, L, cd, which, src_u, src_p, partials, slots_p, eu, b0, nb);
    if (rc) return rc;
  }
  if (phase == PH_INTERIOR) return 0;
  if (phase == PH_GHOST || phase == PH_INTERFACE) {
    if (bk.n_gh > 0) {
      if (!L.xv.ghostbuf || L.xv.n_ghost <= 0) { set_error("launch_op_bricks: ghost nodes without an exchange buffer"); return -1; }
      hipLaunchKernelGGL(k_brick_ghost, dim3((unsigned)((bk.n_gh + 255) / 256)), dim3(256), 0, s, bk.gh_nodes, bk.gh_ptr, bk.gh_list,
                         partials, L.xv.ghostbuf, bk.n_gh, L.n_owned_v);
      AGMG_HIP_CHECK(hipGetLastError());
    }
    return 0;
  }
  BrickFixArgs f{bk.fx_nodes, bk.fx_ptr, bk.fx_list, partials, bk.n_fix, eu, 0xFFFFFFFFu, nullptr};
  if (L.xv.n_export > 0) { f.recv_base = (uint32_t)bk.n_partial_slots; f.recv = L.xv.expbuf; }
  if (bk.n_fix) hipLaunchKernelGGL(k_brick_fix, dim3((unsigned)(((size_t)bk.n_fix * 3 + 255) / 256)), dim3(256), 0, s, f);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace agmg
