// Probe kernel for tests/tools/overlap_probe.py: z = x + y as a streaming pass with a chosen grid and U independent 16-byte
// loads in flight per thread (not part of the product library).
#include <hip/hip_runtime.h>
template <int U>
__global__ __launch_bounds__(256) void probe_kernel(const float4* __restrict__ x, const float4* __restrict__ y, float4* __restrict__ z, long n4) {
  const long stride = (long)gridDim.x * 256, t = (long)blockIdx.x * 256 + threadIdx.x;
  for (long i0 = t; i0 < n4; i0 += U * stride) {
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; const long j = i < n4 ? i : t; a[u] = x[j]; b[u] = y[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) z[i] = make_float4(a[u].x + b[u].x, a[u].y + b[u].y, a[u].z + b[u].z, a[u].w + b[u].w);
    }
  }
}
extern "C" int probe_stream(const float* x, const float* y, float* z, long n4, int grid, int unroll, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const float4 *x4 = (const float4*)x, *y4 = (const float4*)y; float4* z4 = (float4*)z;
  if (unroll == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(grid), dim3(256), 0, s, x4, y4, z4, n4);
  else if (unroll == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(grid), dim3(256), 0, s, x4, y4, z4, n4);
  else if (unroll == 4) hipLaunchKernelGGL(probe_kernel<4>, dim3(grid), dim3(256), 0, s, x4, y4, z4, n4);
  else hipLaunchKernelGGL(probe_kernel<8>, dim3(grid), dim3(256), 0, s, x4, y4, z4, n4);
  return (int)hipGetLastError();
}
