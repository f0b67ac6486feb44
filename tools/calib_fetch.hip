// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths of this
// library's kernels (MI355X_MICROARCH.md: only 16-B-per-lane streaming is documented).
// Each kernel moves a known byte count (N doubles = 1 GiB, far beyond the 256 MiB L3).
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) read8(const double *x, double *out, size_t n) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void __launch_bounds__(256) read16(const double2 *x, double *out, size_t n2) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = x[i]; acc += v.x + v.y; }
  if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void __launch_bounds__(256) read2(const uint16_t *x, double *out, size_t n) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}
// 24-byte records gathered by a permuted node index (the velocity gather / fix-up pattern)
__global__ void __launch_bounds__(256) gather24(const double *x, const uint32_t *idx, double *out, size_t nn) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nn; i += (size_t)gridDim.x * 256) {
    const double *p = x + (size_t)idx[i] * 3;
    acc += p[0] + p[1] + p[2];
  }
  if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void __launch_bounds__(256) write8(double *x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = (double)i;
}
__global__ void __launch_bounds__(256) write24(double *x, size_t nn) {   // one thread per 24-byte node record
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nn; i += (size_t)gridDim.x * 256) {
    x[3 * i] = 1.0; x[3 * i + 1] = 2.0; x[3 * i + 2] = 3.0;
  }
}
int main() {
  const size_t n = (size_t)1 << 27;   // 1 GiB of doubles
  double *x, *out; uint32_t *idx;
  CK(hipMalloc((void **)&x, n * 8)); CK(hipMalloc((void **)&out, 64)); CK(hipMalloc((void **)&idx, (n / 3) * 4));
  CK(hipMemset(x, 0, n * 8));
  const size_t nn = n / 3;
  {  // node permutation with block locality (consecutive threads -> nodes within a 4096-node window)
    uint32_t *h = (uint32_t *)malloc(nn * 4);
    for (size_t i = 0; i < nn; ++i) { const size_t w = i / 4096, j = i % 4096; h[i] = (uint32_t)(w * 4096 + (j * 1237) % 4096); }
    for (size_t i = nn / 4096 * 4096; i < nn; ++i) h[i] = (uint32_t)i;
    CK(hipMemcpy(idx, h, nn * 4, hipMemcpyHostToDevice)); free(h);
  }
  const dim3 g(4096), b(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read8, g, b, 0, 0, x, out, n);
    hipLaunchKernelGGL(read16, g, b, 0, 0, (const double2 *)x, out, n / 2);
    hipLaunchKernelGGL(read2, g, b, 0, 0, (const uint16_t *)x, out, n * 4);
    hipLaunchKernelGGL(gather24, g, b, 0, 0, x, idx, out, nn);
    hipLaunchKernelGGL(write8, g, b, 0, 0, x, n);
    hipLaunchKernelGGL(write24, g, b, 0, 0, x, nn);
  }
  CK(hipDeviceSynchronize());
  printf("bytes per launch: read8/read16/read2/write8 %zu, gather24 %zu (+%zu index), write24 %zu\n", n * 8, nn * 24, nn * 4, nn * 24);
  return 0;
}
