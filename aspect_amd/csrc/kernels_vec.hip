// kernels_vec.hip -- streaming vector kernels of the Krylov / smoother layer (gfx950).
// Replace the deal.II LinearAlgebra::distributed::Vector operations used by SolverGMRES /
// SolverFGMRES / SolverIDR / SolverCG / PreconditionChebyshev on this path.
// All are HBM-bound: grid-stride, 16 B per lane per access where alignment allows, reductions
// in a fixed order (per-block tree, then one block over the block partials) so results are
// bitwise reproducible run to run.
#include "agmg_internal.h"

namespace agmg {
namespace {

constexpr int VB = 256;
inline unsigned grid_for(size_t n) {
  size_t g = (n + VB - 1) / VB;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

__global__ void __launch_bounds__(VB) k_set(double *x, double v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) x[i] = v;
}
__global__ void __launch_bounds__(VB) k_copy(double *d, const double *s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) d[i] = s[i];
}
__global__ void __launch_bounds__(VB) k_scale(double *x, double a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) x[i] *= a;
}
__global__ void __launch_bounds__(VB) k_axpy(double *y, double a, const double *x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) y[i] = fma(a, x[i], y[i]);
}
__global__ void __launch_bounds__(VB) k_sadd(double *y, double sy, double a, const double *x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) y[i] = fma(a, x[i], sy * y[i]);
}
__global__ void __launch_bounds__(VB) k_lincomb(double *d, double a, const double *x, double b, const double *y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) d[i] = fma(a, x[i], b * y[i]);
}
__global__ void __launch_bounds__(VB) k_pointwise(double *d, double f, const double *dg, const double *b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) d[i] = f * dg[i] * b[i];
}
__global__ void __launch_bounds__(VB) k_zero_constrained(double *x, const uint8_t *cmask, size_t n, int ncomp) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t node = i / ncomp;
    const int c = (int)(i - node * ncomp);
    if ((cmask[node] >> c) & 1u) x[i] = 0.0;
  }
}
__global__ void __launch_bounds__(VB) k_shift(double *x, double a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) x[i] += a;
}
__global__ void __launch_bounds__(VB) k_fill_mod(double *x, size_t n, int mod) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) x[i] = (double)(i % mod);
}

template <int NV>
__global__ void __launch_bounds__(VB) k_multi_dot(const double *x, VecPtrs ys, size_t n, size_t seg1, size_t gap,
                                                  double *partials) {
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0;
  for (size_t il = (size_t)blockIdx.x * VB + threadIdx.x; il < n; il += (size_t)gridDim.x * VB) {
    const size_t i = il < seg1 ? il : il + gap;
    const double xv = x[i];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = fma(xv, ys.p[k][i], acc[k]);
  }
  __shared__ double red[NV][VB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double v = 0.0;
    for (int w = 0; w < VB / 64; ++w) v += red[threadIdx.x][w];
    partials[(size_t)threadIdx.x * DOT_BLOCKS + blockIdx.x] = v;
  }
}
__global__ void __launch_bounds__(DOT_BLOCKS) k_dot_final(const double *partials, int nblocks, double *out) {
  // one block; blockIdx.x = vector index
  const int k = blockIdx.x;
  double v = threadIdx.x < nblocks ? partials[(size_t)k * DOT_BLOCKS + threadIdx.x] : 0.0;
  __shared__ double red[DOT_BLOCKS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < DOT_BLOCKS / 64; ++w) s += red[w];
    out[k] = s;
  }
}

__global__ void __launch_bounds__(VB) k_fill_gid_mod(double *x, const uint32_t *gid, size_t n, int nc, int mod) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t node = i / nc;
    const size_t c = i - node * nc;
    x[i] = (double)(((size_t)gid[node] * nc + c) % mod);
  }
}
__global__ void __launch_bounds__(VB) k_gather_nodes(double *out, const double *x, const uint32_t *nodes, size_t n, int nc) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t k = i / nc;
    const size_t c = i - k * nc;
    out[i] = x[(size_t)nodes[k] * nc + c];
  }
}


// ---- hanging-node constraints of the active mesh of an adaptive context (MatrixFree's read_dof_values /
// distribute_local_to_global around the cell loop) and the level copies of MGLevelGlobalTransfer, as index kernels.
// Masters are never hanging themselves (checked at agmg_set_level), so readers and writers never meet.
// x[h] <- sum_k w_k x[m_k] over the unconstrained masters; the entry as it came in is kept in `saved`
__global__ void __launch_bounds__(VB) k_hang_resolve(double *x, const uint32_t *hn, const int32_t *hptr, const uint32_t *hm,
                                                     const double *hw, const uint8_t *cmask, size_t n, int nc, double *saved) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t k = i / nc;
    const int c = (int)(i - k * nc);
    double v = 0.0;
    for (int q = hptr[k]; q < hptr[k + 1]; ++q) {
      const uint32_t m = hm[q];
      if (!((cmask[m] >> c) & 1u)) v += hw[q] * x[(size_t)m * nc + c];
    }
    const size_t at = (size_t)hn[k] * nc + c;
    saved[i] = x[at];
    x[at] = v;
  }
}
// transpose, one thread per (master, component), the hanging nodes of a master in the order of the constraint table:
// y[m] += sum w y[h] over its unconstrained hanging nodes
__global__ void __launch_bounds__(VB) k_hang_distribute(double *y, const uint32_t *tm, const int32_t *tptr, const uint32_t *th,
                                                        const double *tw, const uint8_t *cmask, size_t n, int nc) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t k = i / nc;
    const int c = (int)(i - k * nc);
    const uint32_t m = tm[k];
    if ((cmask[m] >> c) & 1u) continue;
    double v = y[(size_t)m * nc + c];
    for (int q = tptr[k]; q < tptr[k + 1]; ++q) {
      const uint32_t h = th[q];
      if (!((cmask[h] >> c) & 1u)) v += tw[q] * y[(size_t)h * nc + c];
    }
    y[(size_t)m * nc + c] = v;
  }
}
// identity rows of the hanging nodes
__global__ void __launch_bounds__(VB) k_hang_restore(double *y, const uint32_t *hn, const double *saved, size_t n, int nc) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t k = i / nc;
    y[(size_t)hn[k] * nc + (i - k * nc)] = saved[i];
  }
}
__global__ void __launch_bounds__(VB) k_copy_nodes(double *dst, const uint32_t *dn, const double *src, const uint32_t *sn, size_t n, int nc) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    const size_t k = i / nc, c = i - k * nc;
    dst[(size_t)dn[k] * nc + c] = src[(size_t)sn[k] * nc + c];
  }
}

template <int NV>
__global__ void __launch_bounds__(VB) k_multi_axpy(double *y, VecPtrs xs, Coeffs a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    double v = y[i];
#pragma unroll
    for (int k = 0; k < NV; ++k) v = fma(a.a[k], xs.p[k][i], v);
    y[i] = v;
  }
}

}  // namespace

#define LAUNCH_VEC(kern, n, ...)                                                        \
  do {                                                                                  \
    if ((n) == 0) return 0;                                                             \
    hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(VB), 0, s, __VA_ARGS__);           \
    AGMG_HIP_CHECK(hipGetLastError());                                                  \
    return 0;                                                                           \
  } while (0)

int vec_set(hipStream_t s, double *x, double v, size_t n) { LAUNCH_VEC(k_set, n, x, v, n); }
int vec_copy(hipStream_t s, double *d, const double *src, size_t n) { LAUNCH_VEC(k_copy, n, d, src, n); }
int vec_scale(hipStream_t s, double *x, double a, size_t n) { LAUNCH_VEC(k_scale, n, x, a, n); }
int vec_axpy(hipStream_t s, double *y, double a, const double *x, size_t n) { LAUNCH_VEC(k_axpy, n, y, a, x, n); }
int vec_sadd(hipStream_t s, double *y, double sy, double a, const double *x, size_t n) {
  LAUNCH_VEC(k_sadd, n, y, sy, a, x, n);
}
int vec_lincomb(hipStream_t s, double *d, double a, const double *x, double b, const double *y, size_t n) {
  LAUNCH_VEC(k_lincomb, n, d, a, x, b, y, n);
}
int vec_pointwise_scale(hipStream_t s, double *d, double f, const double *dg, const double *b, size_t n) {
  LAUNCH_VEC(k_pointwise, n, d, f, dg, b, n);
}
int vec_zero_constrained(hipStream_t s, double *x, const uint8_t *cmask, int n_nodes, int ncomp) {
  const size_t n = (size_t)n_nodes * ncomp;
  LAUNCH_VEC(k_zero_constrained, n, x, cmask, n, ncomp);
}
int vec_shift(hipStream_t s, double *x, double a, size_t n) { LAUNCH_VEC(k_shift, n, x, a, n); }
// coefficient tables arrive [cell][q] (the caller's flattened Table) and live [q][cstride]: a tile
// through LDS keeps both sides coalesced
__global__ void __launch_bounds__(256) k_transpose_table(double *dst, const double *src, size_t n_cells, int nq, size_t cs, int *bad) {
  __shared__ double tile[64][33];
  const size_t cell0 = (size_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  bool any_bad = false;
  for (int q0 = 0; q0 < nq; q0 += 32) {
    // read: rows = cells, contiguous in q
    for (int r = ty; r < 64; r += 8) {
      const size_t cell = cell0 + r;
      const int q = q0 + tx;
      double v = 1.0;
      if (cell < n_cells && q < nq) {
        v = src[cell * nq + q];
        if (!(v > 0)) any_bad = true;
      }
      tile[r][tx] = v;
    }
    __syncthreads();
    // write: rows = q, contiguous in cells
    for (int qq = ty; qq < 32; qq += 8) {
      const int q = q0 + qq;
      if (q < nq) {
        for (int half = 0; half < 2; ++half) {
          const size_t cell = cell0 + tx + 32 * half;
          if (cell < cs) dst[(size_t)q * cs + cell] = tile[tx + 32 * half][qq];
        }
      }
    }
    __syncthreads();
  }
  if (any_bad) atomicOr(bad, 1);
}
int vec_transpose_table(hipStream_t s, double *dst, const double *src, size_t n_cells, int nq, size_t cstride, int *bad) {
  if (cstride == 0) return 0;
  hipLaunchKernelGGL(k_transpose_table, dim3((unsigned)((cstride + 63) / 64)), dim3(256), 0, s, dst, src, n_cells, nq, cstride, bad);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int vec_fill_index_mod(hipStream_t s, double *x, size_t n, int mod) { LAUNCH_VEC(k_fill_mod, n, x, n, mod); }

int vec_fill_gid_mod(hipStream_t s, double *x, const uint32_t *gid, int n_nodes, int nc, int mod) {
  const size_t n = (size_t)n_nodes * nc;
  LAUNCH_VEC(k_fill_gid_mod, n, x, gid, n, nc, mod);
}
int vec_gather_nodes(hipStream_t s, double *out, const double *x, const uint32_t *nodes, int n_nodes, int nc) {
  const size_t n = (size_t)n_nodes * nc;
  LAUNCH_VEC(k_gather_nodes, n, out, x, nodes, n, nc);
}

int vec_hang_resolve(hipStream_t s, double *x, const HangTables &t, const uint8_t *cmask, int nc) {
  const size_t n = (size_t)t.n_hang * nc;
  LAUNCH_VEC(k_hang_resolve, n, x, t.hn, t.hptr, t.hm, t.hw, cmask, n, nc, t.saved);
}
int vec_hang_distribute(hipStream_t s, double *y, const HangTables &t, const uint8_t *cmask, int nc) {
  const size_t n = (size_t)t.n_masters * nc;
  LAUNCH_VEC(k_hang_distribute, n, y, t.tm, t.tptr, t.th, t.tw, cmask, n, nc);
}
int vec_hang_restore(hipStream_t s, double *y, const HangTables &t, int nc) {
  const size_t n = (size_t)t.n_hang * nc;
  LAUNCH_VEC(k_hang_restore, n, y, t.hn, t.saved, n, nc);
}
int vec_copy_nodes(hipStream_t s, double *dst, const uint32_t *dst_nodes, const double *src, const uint32_t *src_nodes, int n_nodes, int nc) {
  const size_t n = (size_t)n_nodes * nc;
  LAUNCH_VEC(k_copy_nodes, n, dst, dst_nodes, src, src_nodes, n, nc);
}

int vec_multi_dot(hipStream_t s, const double *x, const VecPtrs &ys, int nvec, size_t n, size_t seg1, size_t gap,
                  double *partials, double *out) {
  if (nvec < 1 || nvec > DOT_MAX_VECS) { set_error("vec_multi_dot: bad nvec"); return -1; }
  size_t g = (n + VB - 1) / VB;
  if (g > DOT_BLOCKS) g = DOT_BLOCKS;
  if (g < 1) g = 1;
  const dim3 grid((unsigned)g), block(VB);
#define AGMG_DOT_CASE(N) case N: hipLaunchKernelGGL((k_multi_dot<N>), grid, block, 0, s, x, ys, n, seg1, gap, partials); break
  switch (nvec) {
    AGMG_DOT_CASE(1); AGMG_DOT_CASE(2); AGMG_DOT_CASE(3); AGMG_DOT_CASE(4); AGMG_DOT_CASE(5); AGMG_DOT_CASE(6); AGMG_DOT_CASE(7); AGMG_DOT_CASE(8);
    AGMG_DOT_CASE(9); AGMG_DOT_CASE(10); AGMG_DOT_CASE(11); AGMG_DOT_CASE(12); AGMG_DOT_CASE(13); AGMG_DOT_CASE(14); AGMG_DOT_CASE(15);
    default: hipLaunchKernelGGL((k_multi_dot<16>), grid, block, 0, s, x, ys, n, seg1, gap, partials); break;
  }
#undef AGMG_DOT_CASE
  hipLaunchKernelGGL(k_dot_final, dim3(nvec), dim3(DOT_BLOCKS), 0, s, partials, (int)g, out);
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

int vec_multi_axpy(hipStream_t s, double *y, const VecPtrs &xs, const Coeffs &a, int nvec, size_t n) {
  if (nvec < 1 || nvec > DOT_MAX_VECS) { set_error("vec_multi_axpy: bad nvec"); return -1; }
  if (n == 0) return 0;
  const dim3 grid(grid_for(n)), block(VB);
#define AGMG_AXPY_CASE(N) case N: hipLaunchKernelGGL((k_multi_axpy<N>), grid, block, 0, s, y, xs, a, n); break
  switch (nvec) {
    AGMG_AXPY_CASE(1); AGMG_AXPY_CASE(2); AGMG_AXPY_CASE(3); AGMG_AXPY_CASE(4); AGMG_AXPY_CASE(5); AGMG_AXPY_CASE(6); AGMG_AXPY_CASE(7); AGMG_AXPY_CASE(8);
    AGMG_AXPY_CASE(9); AGMG_AXPY_CASE(10); AGMG_AXPY_CASE(11); AGMG_AXPY_CASE(12); AGMG_AXPY_CASE(13); AGMG_AXPY_CASE(14); AGMG_AXPY_CASE(15);
    default: hipLaunchKernelGGL((k_multi_axpy<16>), grid, block, 0, s, y, xs, a, n); break;
  }
#undef AGMG_AXPY_CASE
  AGMG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace agmg
