// Shared helpers for the STiL gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define STIL_OK 0
#define STIL_EINVAL -1
#define STIL_EUNSUPPORTED -2
#define STIL_EHIP -3

extern "C" const char* stil_last_error(void);
void stil_set_error(const char* fmt, ...);

#define STIL_REQUIRE(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      stil_set_error(__VA_ARGS__);             \
      return STIL_EINVAL;                      \
    }                                          \
  } while (0)

#define STIL_LAUNCH_CHECK()                                             \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      stil_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,  \
                     hipGetErrorString(e_));                            \
      return STIL_EHIP;                                                 \
    }                                                                   \
  } while (0)

#define WAVE 64

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `sh` must hold >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];  // fixed order: deterministic
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}

// Fixed-order reduction of per-chunk partials: a block of 32*LANES threads owns 32 columns; lane l sums chunks
// l, l+LANES, ... and the LANES lane-sums are added in order (deterministic).  part: [nplanes][nch][C].
// Returns true in the threads (lane 0) that hold the totals of column c.
template <int LANES, int NPL>
__device__ __forceinline__ bool chunk_reduce(const float* __restrict__ part, int nch, int C, float* sh, int& c,
                                             float (&tot)[NPL]) {
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  c = blockIdx.x * 32 + cl;
  float s[NPL];
#pragma unroll
  for (int q = 0; q < NPL; ++q) s[q] = 0.f;
  if (c < C)
    for (int i = lane; i < nch; i += LANES)
#pragma unroll
      for (int q = 0; q < NPL; ++q) s[q] += part[((long)q * nch + i) * C + c];
#pragma unroll
  for (int q = 0; q < NPL; ++q) sh[(q * LANES + lane) * 32 + cl] = s[q];
  __syncthreads();
  if (lane != 0 || c >= C) return false;
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    float t = 0.f;
    for (int l = 0; l < LANES; ++l) t += sh[(q * LANES + l) * 32 + cl];
    tot[q] = t;
  }
  return true;
}

// exact (erf) GELU and its derivative: nn.GELU() default.
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
