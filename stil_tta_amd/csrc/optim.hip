// Flat-buffer optimizer kernels: the student's parameters, gradients and Adam moments each live in ONE
// contiguous HBM slab (every tensor padded to a 1024-element boundary), and the EMA teacher mirrors the
// same layout, so that
//   * the EMA update (STiLModel.py:154-168) is a single streaming axpby over 43 M floats,
//   * Adam (STiLModel.py:563-570, torch.optim.Adam semantics incl. per-parameter step counts and
//     "grad is None -> skipped") is a single pass, and
//   * the data-parallel gradient exchange is one RCCL all-reduce of the slab.
// HBM-bound: 16-byte loads/stores, grid-stride.
#include "common.h"

// e = m*e + (1-m)*v with the reference's rounding: fl(fl(m*e) + fl(c*v)), c = float(1-m)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ v, long n4, float m,
                                                   float c) {
#pragma clang fp contract(off)  // HIP's __fmul_rn/__fadd_rn are plain operators: forbid FMA contraction here
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(e)[i];
    const float4 b = reinterpret_cast<const float4*>(v)[i];
    float t0, t1;
    t0 = a.x * m; t1 = c * b.x; a.x = t0 + t1;
    t0 = a.y * m; t1 = c * b.y; a.y = t0 + t1;
    t0 = a.z * m; t1 = c * b.z; a.z = t0 + t1;
    t0 = a.w * m; t1 = c * b.w; a.w = t0 + t1;
    reinterpret_cast<float4*>(e)[i] = a;
  }
}

__global__ void adam_tick_kernel(int* __restrict__ steps, const unsigned char* __restrict__ active, int nt) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nt && active[t]) steps[t] += 1;
}

// one block per 1024-element chunk; chunk2tensor gives the owning tensor (step count / active flag)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const int* __restrict__ chunk2tensor,
                                                    const int* __restrict__ steps,
                                                    const unsigned char* __restrict__ active, long nchunks, float lr,
                                                    float b1, float b2, float eps, float wd, float gscale) {
  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int t = chunk2tensor[ch];
    if (t < 0 || !active[t]) continue;
    const int step = steps[t];
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    const long i = ch * 256 + threadIdx.x;
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(P, G, M, V)                                   \
    {                                                       \
      float gr = G * gscale;                                \
      if (wd != 0.f) gr += wd * P;                          \
      M = M * b1 + (1.f - b1) * gr;                         \
      V = V * b2 + (1.f - b2) * gr * gr;                    \
      float den = sqrtf(V) / bc2s + eps;                    \
      P = P - step_size * (M / den);                        \
    }
    ADAM1(pp.x, gg.x, mm.x, vv.x) ADAM1(pp.y, gg.y, mm.y, vv.y) ADAM1(pp.z, gg.z, mm.z, vv.z) ADAM1(pp.w, gg.w, mm.w, vv.w)
#undef ADAM1
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

extern "C" int stil_ema_update(float* ema, const float* model, long n, double momentum, void* stream) {
  STIL_REQUIRE(ema && model && n >= 0 && n % 4 == 0, "stil_ema_update: n=%ld must be a multiple of 4", n);
  STIL_REQUIRE(((uintptr_t)ema % 16 == 0) && ((uintptr_t)model % 16 == 0), "stil_ema_update: slabs must be 16-byte aligned");
  if (n == 0) return STIL_OK;
  long n4 = n / 4;
  long g = (n4 + 255) / 256;
  int grid = (int)(g < 8192 ? g : 8192);
  // python: v_ema.mul_(momentum).add_((1. - momentum) * v_main): both scalars are rounded to fp32 from DOUBLE
  const float m = (float)momentum, c = (float)(1.0 - momentum);
  hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ema, model, n4, m, c);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// CoTraining_SAINT.momentum_update_ema (models/SemiMultimodal/CoTraining_SAINT.py:102-105) applies
// `v_ema.copy_(v_ema * m + (1 - m) * v_main)` to SAINT's int64 *_offset buffers as well: float32 arithmetic, then the copy
// truncates toward zero -- so 29 becomes 28 at m = 0.996.  Same rounding sequence here; parity is with the reference as shipped.
__global__ void ema_int_trunc_kernel(long long* __restrict__ e, const long long* __restrict__ v, int n, float m, float c) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t0 = (float)e[i] * m, t1 = c * (float)v[i];
  e[i] = (long long)(t0 + t1);
}

extern "C" int stil_ema_int_trunc(long long* ema, const long long* model, int n, double momentum, void* stream) {
  STIL_REQUIRE(ema && model && n >= 0, "stil_ema_int_trunc: bad arguments");
  if (n == 0) return STIL_OK;
  hipLaunchKernelGGL(ema_int_trunc_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ema, model, n, (float)momentum,
                     (float)(1.0 - momentum));
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// n must be a multiple of 1024 (tensor slots are 1024-aligned); chunk2tensor has n/1024 entries (-1 = padding only).
// grad_scale multiplies every gradient first (1/world_size after a SUM all-reduce).
extern "C" int stil_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              const int* chunk2tensor, int* steps, const unsigned char* active, int n_tensors, long n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                              void* stream) {
  STIL_REQUIRE(params && grads && exp_avg && exp_avg_sq && chunk2tensor && steps && active, "stil_adam_step: null pointer");
  STIL_REQUIRE(n % 1024 == 0, "stil_adam_step: n=%ld must be a multiple of 1024", n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(cdiv(n_tensors, 256)), dim3(256), 0, s, steps, active, n_tensors);
  STIL_LAUNCH_CHECK();
  long nch = n / 1024;
  int grid = (int)(nch < 16384 ? nch : 16384);
  if (grid > 0) {
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, chunk2tensor, steps,
                       active, nch, lr, beta1, beta2, eps, weight_decay, grad_scale);
    STIL_LAUNCH_CHECK();
  }
  return STIL_OK;
}
