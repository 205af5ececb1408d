// Ablation of the 64x64-tile NT main loop (one 32x32 accumulator per wave, BK = 16, 8 blocks per CU): which stage costs what.
// build: hipcc --offload-arch=gfx950 -O3 -o tests/tools/mfma_ablate11 tests/tools/mfma_ablate11.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only; 1: + ds_read_b128 operands; 2: + barrier; 3: + 2 ds_write_b128; 4: + 2 global_load_dwordx4
// MODE 5: like 4 but LDS stores issued right after the loads land at the TOP of the next iteration (software pipelining variant)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ A, const float* __restrict__ B, int K) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * 128 * 20; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b = a, g0 = a, g1 = a;
  const int kq = tid & 3, r0 = tid >> 2;
  const float* pa = A + ((long)blockIdx.x * 64 + r0) * K + kq * 4;
  const float* pb = B + (long)r0 * K + kq * 4;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE >= 4) {
      const int k0 = ((it + 1) * 16) % K;
      g0 = *reinterpret_cast<const float4*>(pa + k0);
      g1 = *reinterpret_cast<const float4*>(pb + k0);
    }
    const float* Ab = lds + buf * 128 * 20 + (wm * 32 + li) * 20 + lh * 4;
    const float* Bb = lds + buf * 128 * 20 + (64 + wn * 32 + li) * 20 + lh * 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (MODE >= 1) {
        a = *reinterpret_cast<const float4*>(Ab + t * 8);
        b = *reinterpret_cast<const float4*>(Bb + t * 8);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    if (MODE >= 3) {
      float* dst = lds + (buf ^ 1) * 128 * 20;
      *reinterpret_cast<float4*>(dst + r0 * 20 + kq * 4) = g0;
      *reinterpret_cast<float4*>(dst + (64 + r0) * 20 + kq * 4) = g1;
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, float* out, const float* A, const float* B, int K) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  size_t lds = 2 * 128 * 20 * sizeof(float);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters, A, B, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters, A, B, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 8 * 4096.0;
  printf("%-40s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  const int maxb = 4096;
  float* out; hipMalloc(&out, (size_t)maxb * 256 * sizeof(float));
  const int K = 2048;
  float *A, *B;
  hipMalloc(&A, (size_t)maxb * 64 * K * sizeof(float)); hipMalloc(&B, (size_t)64 * K * sizeof(float));
  hipMemset(A, 0, (size_t)maxb * 64 * K * sizeof(float)); hipMemset(B, 0, (size_t)64 * K * sizeof(float));
  for (int blocks : {1024, 2048, 4096}) {
    run<0>("mfma only", blocks, 1000, out, A, B, K);
    run<1>("+ ds_read_b128 operands", blocks, 1000, out, A, B, K);
    run<2>("+ barrier / iteration", blocks, 1000, out, A, B, K);
    run<3>("+ 2 ds_write_b128 / iteration", blocks, 1000, out, A, B, K);
    run<4>("+ 2 global_load_dwordx4 (K=2048)", blocks, 1000, out, A, B, K);
    run<4>("+ 2 global_load_dwordx4 (K=256, L2)", blocks, 1000, out, A, B, 256);
  }
  return 0;
}
