// Pricing of the split-precision GEMM (SURVEY section 7's opt-in perf mode; round-4 verdict item 4) on gfx950 -- measurement tool.
// An fp32 operand is split into three bf16 terms a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 24
// mantissa bits in all) and the product is formed from the six term pairs of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate): 6 x 32 cycles per 32x32x16 block against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 -- a 2.67x higher ceiling.
//   P: operands pre-split into bf16 planes in HBM (6 bytes per element instead of 4), 128x128 tile, 4 waves of 2x2 MFMA tiles
//   F: fp32 operands split on the fly while they are staged into LDS (what a drop-in for stil_gemm_nt would do)
// against the fp32-exact 64x64 kernel of tests/tools/gemm_lab.hip, on the step's shapes, with the relative L2 error of each against
// a float64 product.        build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/tools/bf16x3_lab tests/tools/bf16x3_lab.hip
#define GEMM_LAB_NO_MAIN
#include "gemm_lab.hip"
#include <math.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct Args3 {
  const u16* A3; const u16* W3;   // P: [3][M][K], [3][N][K] bf16 planes
  const float* A; const float* W; // F: fp32 operands
  float* C;
  int M, N, K;
};

__device__ __forceinline__ void split3(float a, __bf16& a1, __bf16& a2, __bf16& a3) {
  a1 = (__bf16)a;
  const float r1 = a - (float)a1;      // exact
  a2 = (__bf16)r1;
  const float r2 = r1 - (float)a2;     // exact
  a3 = (__bf16)r2;
}

__global__ void split_planes_kernel(const float* __restrict__ x, u16* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    __bf16 a1, a2, a3;
    split3(x[i], a1, a2, a3);
    p[i] = *reinterpret_cast<u16*>(&a1); p[n + i] = *reinterpret_cast<u16*>(&a2); p[2 * n + i] = *reinterpret_cast<u16*>(&a3);
  }
}

// (64 TM) x (64 TN) block tile, 256 threads = 2x2 waves x (TM x TN tiles of 32x32), BK = 32, one LDS buffer:
// [A planes 3][BM rows][40 bf16] then [W planes 3][BN rows][40 bf16]
// ACC2: two-level accumulation as in the fp32 kernels (chains of 64 k into a partial accumulator, then into the master)
template <int TM, int TN, bool FLY, bool ACC2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TM * TN >= 4 ? 2 : (TM * TN == 2 ? 3 : 4)))) void bf16x3_kernel(Args3 p) {
  constexpr int LS = 40;                       // bf16 per LDS row: 64 B of data + 16 B pad (conflict-free ds_read_b128)
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int PLA = BM * LS, PLB = BN * LS;  // one plane of A / of W
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  __bf16* lds = reinterpret_cast<__bf16*>(lds_f);
  __bf16* ldsB = lds + 3 * PLA;
  const int nbn = p.N / BN, nbm = p.M / BM;
  const int wg = xcd_remap(blockIdx.x, nbm * nbn);
  const int tm = wg / nbn, tn = wg - tm * nbn;
  const int tid = threadIdx.x, w = tid >> 6, wm = w >> 1, wn = w & 1, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int FA = BM / 32, FB = BN / 32;    // FLY: float4 per thread and k-tile
  constexpr int PA = BM / 64, PB = BN / 64;    // planes: 16-byte chunks per thread, plane and k-tile
  f32x4 st[FLY ? (FA + FB) : 3 * (PA + PB)];
  const long Mk = (long)p.M * p.K, Nk = (long)p.N * p.K;
  int k0 = 0;
  auto load = [&]() __attribute__((always_inline)) {
    if constexpr (FLY) {
#pragma unroll
      for (int i = 0; i < FA; ++i) { const int c = tid + 256 * i, row = c >> 3, q = c & 7; st[i] = *reinterpret_cast<const f32x4*>(p.A + (long)(tm * BM + row) * p.K + k0 + q * 4); }
#pragma unroll
      for (int i = 0; i < FB; ++i) { const int c = tid + 256 * i, row = c >> 3, q = c & 7; st[FA + i] = *reinterpret_cast<const f32x4*>(p.W + (long)(tn * BN + row) * p.K + k0 + q * 4); }
    } else {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < PA; ++i) { const int c = tid + 256 * i, row = c >> 2, q = c & 3; st[pl * PA + i] = *reinterpret_cast<const f32x4*>(p.A3 + pl * Mk + (long)(tm * BM + row) * p.K + k0 + q * 8); }
#pragma unroll
        for (int i = 0; i < PB; ++i) { const int c = tid + 256 * i, row = c >> 2, q = c & 3; st[3 * PA + pl * PB + i] = *reinterpret_cast<const f32x4*>(p.W3 + pl * Nk + (long)(tn * BN + row) * p.K + k0 + q * 8); }
      }
    }
    k0 += 32;
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    if constexpr (FLY) {
#pragma unroll
      for (int i = 0; i < FA + FB; ++i) {
        const bool isB = i >= FA;
        const int c = tid + 256 * (isB ? i - FA : i), row = c >> 3, q = c & 7;
        bf16x4 h1, h2, h3;
#pragma unroll
        for (int e = 0; e < 4; ++e) { __bf16 a1, a2, a3; split3(st[i][e], a1, a2, a3); h1[e] = a1; h2[e] = a2; h3[e] = a3; }
        __bf16* dst = (isB ? ldsB : lds) + row * LS + q * 4;
        const int PL = isB ? PLB : PLA;
        *reinterpret_cast<bf16x4*>(dst) = h1; *reinterpret_cast<bf16x4*>(dst + PL) = h2; *reinterpret_cast<bf16x4*>(dst + 2 * PL) = h3;
      }
    } else {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < PA; ++i) { const int c = tid + 256 * i, row = c >> 2, q = c & 3; *reinterpret_cast<f32x4*>(lds + pl * PLA + row * LS + q * 8) = st[pl * PA + i]; }
#pragma unroll
        for (int i = 0; i < PB; ++i) { const int c = tid + 256 * i, row = c >> 2, q = c & 3; *reinterpret_cast<f32x4*>(ldsB + pl * PLB + row * LS + q * 8) = st[3 * PA + pl * PB + i]; }
      }
    }
  };
  const int nk = p.K / 32;
  load();
  lstore();
  __syncthreads();
  f32x16 part[TM][TN];
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load();
    if (ACC2 && (kt & 1) == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][TM], b[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(lds + pl * PLA + (wm * TM * 32 + i * 32 + li) * LS + s * 16 + lh * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(ldsB + pl * PLB + (wn * TN * 32 + j * 32 + li) * LS + s * 16 + lh * 8);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {   // smallest terms first
          f32x16& c = ACC2 ? part[i][j] : acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    if (ACC2 && ((kt & 1) == 1 || kt + 1 == nk)) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] += part[i][j];
    }
    __syncthreads();
    if (kt + 1 < nk) lstore();
    __syncthreads();
  }
  Args q{nullptr, nullptr, p.C, p.M, p.N, p.K, 0, 0, nullptr, nullptr, nullptr};
  Tile<TM, TN, 32, 1>::store(q, lds_f, tm, tn, acc);
}

template <int TM, int TN, bool FLY, bool ACC2>
static void launch_b3(const Args3& p) {
  static bool attr = false;
  if (!attr) { CK(hipFuncSetAttribute((const void*)bf16x3_kernel<TM, TN, FLY, ACC2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); attr = true; }
  const size_t lds = std::max((size_t)3 * 64 * (TM + TN) * 40 * 2, (size_t)64 * 68 * 4);
  hipLaunchKernelGGL((bf16x3_kernel<TM, TN, FLY, ACC2>), dim3((p.M / (64 * TM)) * (p.N / (64 * TN))), dim3(256), lds, 0, p);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 4;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  std::vector<Shape> shapes = {
      {50176, 256, 2304, "layer3 3x3 (as plain K)"}, {12544, 512, 4608, "layer4 3x3 (as plain K)"}, {200704, 128, 1152, "layer2 3x3 (as plain K)"},
      {50176, 256, 1024, "layer3 conv1"}, {50176, 1024, 256, "layer3 conv3"}, {12544, 512, 2048, "layer4 conv1"}, {12544, 2048, 512, "layer4 conv3"},
      {16640, 512, 2048, "tab fc2"}, {16640, 2048, 512, "tab fc1"}, {200704, 128, 512, "layer2 conv1"}, {200704, 512, 128, "layer2 conv3"},
      {802816, 256, 64, "layer1 conv3"}};
  const int SETS = 3;
  size_t maxA = 0, maxW = 0, maxC = 0;
  for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N); }
  float *A[SETS], *W[SETS], *C[SETS];
  u16 *A3[SETS], *W3[SETS];
  std::vector<float> hA(maxA), hW(maxW);
  {
    unsigned s = 12345u;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f * (1.f + (s & 255) * 1e-3f); }
    for (auto& v : hW) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f * (1.f + (s & 255) * 1e-3f); }
  }
  for (int i = 0; i < SETS; ++i) {
    CK(hipMalloc(&A[i], maxA * 4)); CK(hipMalloc(&W[i], maxW * 4)); CK(hipMalloc(&C[i], maxC * 4));
    CK(hipMalloc(&A3[i], maxA * 6)); CK(hipMalloc(&W3[i], maxW * 6));
    CK(hipMemcpy(A[i], hA.data(), maxA * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W[i], hW.data(), maxW * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {   // warm the chip up
    Args p{A[0], W[0], C[0], 50176, 256, 2304, 0, 0, nullptr, nullptr, nullptr};
    for (int i = 0; i < 600; ++i) launch_tile<1, 1, 32, 1, 4>(p, 0);
    CK(hipDeviceSynchronize());
  }
  printf("fp32 operands as 3 bf16 terms, 6 products on v_mfma_f32_32x32x16_bf16; P = pre-split bf16 planes in HBM, F = split on the fly while staging; two-level sums (chains of 64) unless '1-level'\n");
  for (auto& sh : shapes) {
    const size_t nA = (size_t)sh.M * sh.K, nW = (size_t)sh.N * sh.K, nC = (size_t)sh.M * sh.N;
    for (int i = 0; i < SETS; ++i) {   // planes of THIS shape's operands (the operands are the leading nA / nW floats of the buffers)
      hipLaunchKernelGGL(split_planes_kernel, dim3(4096), dim3(256), 0, 0, A[i], A3[i], (long)nA);
      hipLaunchKernelGGL(split_planes_kernel, dim3(4096), dim3(256), 0, 0, W[i], W3[i], (long)nW);
    }
    CK(hipDeviceSynchronize());
    // float64 reference of the first 64 rows
    const int RR = 64;
    std::vector<double> ref((size_t)RR * sh.N);
    for (int r = 0; r < RR; ++r)
      for (int n = 0; n < sh.N; ++n) {
        double s = 0.0;
        const float* a = hA.data() + (size_t)r * sh.K; const float* w_ = hW.data() + (size_t)n * sh.K;
        for (int k = 0; k < sh.K; ++k) s += (double)a[k] * (double)w_[k];
        ref[(size_t)r * sh.N + n] = s;
      }
    // variants: 0 fp32 MFMA 64x64 | P = pre-split planes, F = split on the fly; tile; two-level sums unless "1-level"
    const char* names[] = {"fp32 64x64", "P 128x128", "F 128x128", "P 128x128 1-level", "F 64x128", "F 64x64", "F 128x64", "split pass"};
    constexpr int NV = 8;
    auto launch = [&](int v, int set) {
      if (v == 0) { Args p{A[set], W[set], C[set], sh.M, sh.N, sh.K, 0, 0, nullptr, nullptr, nullptr}; launch_tile<1, 1, 32, 1, 4>(p, 0); return; }
      Args3 p{A3[set], W3[set], A[set], W[set], C[set], sh.M, sh.N, sh.K};
      switch (v) {
        case 1: launch_b3<2, 2, false, true>(p); break;
        case 2: launch_b3<2, 2, true, true>(p); break;
        case 3: launch_b3<2, 2, false, false>(p); break;
        case 4: launch_b3<1, 2, true, true>(p); break;
        case 5: launch_b3<1, 1, true, true>(p); break;
        case 6: launch_b3<2, 1, true, true>(p); break;
        default: hipLaunchKernelGGL(split_planes_kernel, dim3(4096), dim3(256), 0, 0, A[set], A3[set], (long)nA);
      }
    };
    double best[NV], err[NV];
    for (int v = 0; v < NV; ++v) { best[v] = 1e30; err[v] = 0; }
    std::vector<float> out((size_t)RR * sh.N);
    for (int v = 0; v < NV - 1; ++v) {
      CK(hipMemsetAsync(C[0], 0xff, nC * 4, 0));
      launch(v, 0);
      CK(hipDeviceSynchronize()); CK(hipGetLastError());
      CK(hipMemcpy(out.data(), C[0], out.size() * 4, hipMemcpyDeviceToHost));
      double num = 0, den = 0;
      for (size_t i = 0; i < out.size(); ++i) { const double d = (double)out[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; }
      err[v] = sqrt(num / den);
    }
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < NV; ++v) {
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < 6; ++it) launch(v, (it + r) % SETS);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best[v] = std::min(best[v], (double)ms / 6);
      }
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    printf("(%d, %d, %d)\n", sh.M, sh.N, sh.K);
    for (int v = 0; v < NV - 1; ++v) printf("   %-20s %7.1f TF   relL2 vs float64 %.2e\n", names[v], fl / best[v] * 1e-9, err[v]);
    printf("   %-20s %7.0f us (A operand -> 3 bf16 planes)\n", names[NV - 1], best[NV - 1] * 1e3);
    fflush(stdout);
  }
  return 0;
}
