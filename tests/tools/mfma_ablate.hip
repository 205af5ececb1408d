// Ablation of the NT GEMM main loop on gfx950: which stage costs what (MFMA only / + LDS operand reads / + barrier / + LDS writes).
// build: hipcc --offload-arch=gfx950 -O3 -o tests/tools/mfma_ablate tests/tools/mfma_ablate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ A = nullptr, const float* __restrict__ B = nullptr, int K = 0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * 256 * 20; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra = make_float4(1.f, 2.f, 3.f, 4.f);
  float4 a[2], b[2];
  a[0] = a[1] = b[0] = b[1] = ra;
  float4 g0 = ra, g1 = ra, g2 = ra, g3 = ra;
  const int kq_ = tid & 3, r0_ = tid >> 2;
  const float* pa0 = A ? A + ((long)blockIdx.x * 128 + r0_) * K + kq_ * 4 : nullptr;
  const float* pa1 = A ? pa0 + 64L * K : nullptr;
  const float* pb0 = B ? B + (long)r0_ * K + kq_ * 4 : nullptr;
  const float* pb1 = B ? pb0 + 64L * K : nullptr;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE >= 4) {
      const int k0 = ((it + 1) * 16) % K;
      g0 = *reinterpret_cast<const float4*>(pa0 + k0);
      g1 = *reinterpret_cast<const float4*>(pa1 + k0);
      g2 = *reinterpret_cast<const float4*>(pb0 + k0);
      g3 = *reinterpret_cast<const float4*>(pb1 + k0);
    }
    const float* Ab = lds + buf * 256 * 20 + (wm * 64 + li) * 20 + lh * 4;
    const float* Bb = lds + buf * 256 * 20 + (128 + wn * 64 + li) * 20 + lh * 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (MODE >= 1) {
        a[0] = *reinterpret_cast<const float4*>(Ab + t * 8);
        a[1] = *reinterpret_cast<const float4*>(Ab + 32 * 20 + t * 8);
        b[0] = *reinterpret_cast<const float4*>(Bb + t * 8);
        b[1] = *reinterpret_cast<const float4*>(Bb + 32 * 20 + t * 8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (MODE >= 3) {
      const int kq = tid & 3, r0 = tid >> 2;
      float* dst = lds + (buf ^ 1) * 256 * 20;
      *reinterpret_cast<float4*>(dst + (r0) * 20 + kq * 4) = g0;
      *reinterpret_cast<float4*>(dst + (r0 + 64) * 20 + kq * 4) = g1;
      *reinterpret_cast<float4*>(dst + (r0 + 128) * 20 + kq * 4) = g2;
      *reinterpret_cast<float4*>(dst + (r0 + 192) * 20 + kq * 4) = g3;
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, float* out, const float* A = nullptr, const float* B = nullptr, int K = 0) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  size_t lds = 2 * 256 * 20 * sizeof(float);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters, A, B, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters, A, B, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("%-34s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
  const int K = 2048;
  float *A, *B;
  hipMalloc(&A, (size_t)1536 * 128 * K * sizeof(float)); hipMalloc(&B, (size_t)128 * K * sizeof(float));
  hipMemset(A, 0, (size_t)1536 * 128 * K * sizeof(float)); hipMemset(B, 0, (size_t)128 * K * sizeof(float));
  for (int blocks : {768, 1536}) {
    run<4>("+ 4 global_load_dwordx4 (K=2048)", blocks, 2000, out, A, B, K);
    run<4>("+ 4 global_load_dwordx4 (K=256, L2-resident)", blocks, 2000, out, A, B, 256);
  }
  for (int blocks : {768, 1536}) {
    run<0>("mfma only", blocks, 2000, out);
    run<1>("+ ds_read_b128 operands", blocks, 2000, out);
    run<2>("+ barrier / iteration", blocks, 2000, out);
    run<3>("+ 4 ds_write_b128 / iteration", blocks, 2000, out);
  }
  return 0;
}
