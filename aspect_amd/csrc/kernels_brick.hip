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
};
// One thread per node that receives partial sums.  The loads go out in three dependent rounds instead of one per
// list entry: (row bounds, node word) -> (the first FX_PRE list entries, clamped) + the row itself and the
// epilogue's coefficient -> the partial sums; longer rows finish in a loop.  Same order of additions as the list.
__device__ __forceinline__ void brick_fix_body(const BrickFixArgs &F, const unsigned block) {
  constexpr int FX_PRE = 3;
  const int n = (int)block * 256 + threadIdx.x;
  if (n >= F.n_fix) return;
  const uint32_t nd = F.fx_nodes[n];
  const uint32_t s = F.fx_ptr[n], en = F.fx_ptr[n + 1];
  const size_t g0 = (size_t)(nd & NODE_MASK) * 3;
  uint32_t ent[FX_PRE];
#pragma unroll
  for (int k = 0; k < FX_PRE; ++k) ent[k] = F.fx_list[min(s + k, en - 1)];
  double d0[3], coef[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    d0[c] = F.e.dst[g0 + c];
    coef[c] = F.e.epi == EPI_CHEB ? -F.e.f2 * F.e.dinv[g0 + c] : (F.e.epi == EPI_RESID ? -1.0 : 1.0);
  }
  double part[FX_PRE][3];
#pragma unroll
  for (int k = 0; k < FX_PRE; ++k) {
    const double *p = F.partials + (size_t)ent[k] * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) part[k][c] = p[c];
  }
  double sum[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < FX_PRE; ++k)
    if (s + k < en) {
#pragma unroll
      for (int c = 0; c < 3; ++c) sum[c] += part[k][c];
    }
  for (uint32_t k = s + FX_PRE; k < en; ++k) {
    const double *p = F.partials + (size_t)F.fx_list[k] * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) sum[c] += p[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if ((nd >> (29 + c)) & 1u) continue;   // identity row: independent of the sum
    F.e.dst[g0 + c] = fma(coef[c], sum[c], d0[c]);
  }
}
__global__ void __launch_bounds__(256) k_brick_fix(const BrickFixArgs F) { brick_fix_body(F, blockIdx.x); }
__global__ void __launch_bounds__(256) k_brick_fix_pull(const BrickFixArgs F, const PullArgs P, const unsigned n_fix_blocks) {
  if (blockIdx.x < n_fix_blocks) brick_fix_body(F, blockIdx.x);
  else pull_body<1, 1>(P, blockIdx.x - n_fix_blocks);
}

}  // namespace

int launch_op_bricks(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
                     const double *src_p, double *partials, double *slots_p, const EpiArgs &eu, const EpiArgs *ep) {
  const BrickTables &bk = L.bk;
  if (!bk.built || L.dim != 3) { set_error("launch_op_bricks: level has no brick tables"); return -1; }
  if (!cd.set) { set_error("coefficients not set for this level"); return -4; }
  if (eu.epi != EPI_STORE && eu.epi != EPI_RESID && eu.epi != EPI_CHEB) { set_error("launch_op_bricks: epilogue not supported"); return -1; }
  if (ep && which != AGMG_OP_STOKES) { set_error("launch_op_bricks: pressure epilogue misuse"); return -1; }
  if (!brickq_supported(L, cd, which)) { set_error("launch_op_bricks: operator not supported on the brick path"); return -1; }
  const int rc = launch_brickq(s, L, cd, which, src_u, src_p, partials, slots_p, eu);
  if (rc) return rc;
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
