// Evaluation metrics of the STiL module on the device (own translation unit: it pulls in hipCUB's segmented radix sort).
//   * top-k / binary accuracy counters  -- torchmetrics.Accuracy as used by STiLModel.py:122-137,360-363,458-463,530
//   * exact AUROC (thresholds=None)     -- torchmetrics.AUROC binary / multiclass one-vs-rest macro, STiLModel.py:139-145
// AUROC is computed in integers: per class the scores are sorted, U2 = sum over positives of
// (2 * #negatives below + #negatives tied), AUROC = U2 / (2 n_pos n_neg) -- identical to the trapezoid area under the ROC
// curve built on distinct thresholds, with no floating-point accumulation and no atomics on floats.
#include "common.h"
#include <hipcub/hipcub.hpp>

// ---------------------------------------------------------------------------------------------- accuracy counters
// one wave per row: the target class is "in the top k" when fewer than k scores beat it (ties resolved towards the lower
// index, so k = 1 is exactly argmax's first-maximum rule).
__global__ __launch_bounds__(256) void topk_hits_kernel(const float* __restrict__ s, int ld, const long long* __restrict__ y,
                                                         int N, int K, int k, unsigned long long* __restrict__ cnt) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  const long long t = y[row];
  int hit = 0;
  if (t >= 0 && t < K) {
    const float* r = s + (long)row * ld;
    const float st = r[t];
    int beat = 0;
    for (int j = lane; j < K; j += 64) {
      const float v = r[j];
      beat += (v > st) || (v == st && j < t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) beat += __shfl_xor(beat, o, 64);
    hit = beat < k;
  }
  if (lane == 0) {
    if (hit) atomicAdd(cnt, 1ULL);
    atomicAdd(cnt + 1, 1ULL);
  }
}

__global__ __launch_bounds__(256) void binary_hits_kernel(const float* __restrict__ p, const long long* __restrict__ y, int N,
                                                           float thr, unsigned long long* __restrict__ cnt) {
  __shared__ int sh[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  int hit = 0;
  if (i < N) hit = ((p[i] > thr) ? 1 : 0) == (y[i] == 1 ? 1 : 0);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hit += __shfl_xor(hit, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = hit;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(cnt, (unsigned long long)(sh[0] + sh[1] + sh[2] + sh[3]));
    atomicAdd(cnt + 1, (unsigned long long)min(256, N - blockIdx.x * 256));
  }
}

extern "C" int stil_metric_topk(const float* scores, int ld, const long long* target, int N, int K, int k,
                                long long* hits_total, void* stream) {
  STIL_REQUIRE(N >= 0 && K >= 1 && k >= 1 && ld >= K, "metric_topk: bad shape N=%d K=%d k=%d ld=%d", N, K, k, ld);
  if (N == 0) return STIL_OK;
  hipLaunchKernelGGL(topk_hits_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, scores, ld, target, N, K, k,
                     (unsigned long long*)hits_total);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_metric_binary(const float* probs, const long long* target, int N, float threshold, long long* hits_total,
                                  void* stream) {
  STIL_REQUIRE(N >= 0, "metric_binary: bad N=%d", N);
  if (N == 0) return STIL_OK;
  hipLaunchKernelGGL(binary_hits_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, probs, target, N, threshold,
                     (unsigned long long*)hits_total);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// ---------------------------------------------------------------------------------------------- AUROC
struct AurocWs {
  float *kin, *kout;
  unsigned char *fin, *fout;
  int *negpre, *offs;
  void* cub;
  size_t cub_bytes, total;
};

static inline size_t al256(size_t n) { return (n + 255) / 256 * 256; }

static AurocWs auroc_layout(void* base, long N, int K) {
  AurocWs w;
  const size_t KN = (size_t)K * N;
  size_t cub = 0;
  (void)hipcub::DeviceSegmentedRadixSort::SortPairs((void*)nullptr, cub, (const float*)nullptr, (float*)nullptr,
                                              (const unsigned char*)nullptr, (unsigned char*)nullptr, (int)KN, K,
                                              (const int*)nullptr, (const int*)nullptr);
  char* p = (char*)base;
  size_t o = 0;
  w.kin = (float*)(p + o); o += al256(KN * 4);
  w.kout = (float*)(p + o); o += al256(KN * 4);
  w.fin = (unsigned char*)(p + o); o += al256(KN);
  w.fout = (unsigned char*)(p + o); o += al256(KN);
  w.negpre = (int*)(p + o); o += al256((size_t)K * (N + 1) * 4);
  w.offs = (int*)(p + o); o += al256((size_t)(K + 1) * 4);
  w.cub = (void*)(p + o); o += al256(cub);
  w.cub_bytes = cub;
  w.total = o;
  return w;
}

extern "C" size_t stil_auroc_workspace_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || (long)N * K >= (1L << 31)) return 0;
  return auroc_layout(nullptr, N, K).total;
}

// class-major copy: keys[c][i] = scores[i][c]; flag = 1 for the positives of class c (binary: target == 1)
__global__ __launch_bounds__(256) void auroc_scatter_kernel(const float* __restrict__ s, int ld, const long long* __restrict__ y,
                                                             int N, int K, int binary, float* __restrict__ keys,
                                                             unsigned char* __restrict__ flags, int* __restrict__ offs) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx <= K) offs[idx] = (int)(idx * N);
  if (idx >= (long)K * N) return;
  const int c = (int)(idx / N), i = (int)(idx % N);
  keys[idx] = s[(long)i * ld + c];
  flags[idx] = binary ? (y[i] == 1) : (y[i] == c);
}

// one block per class: exclusive prefix count of negatives over the sorted order (negpre[c][0..N])
__global__ __launch_bounds__(1024) void auroc_scan_kernel(const unsigned char* __restrict__ flags, int N, int* __restrict__ negpre) {
  __shared__ int shw[16];
  __shared__ int carry;
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned char* f = flags + (long)c * N;
  int* np = negpre + (long)c * (N + 1);
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < N) ? (f[i] ? 0 : 1) : 0;
    int x = v;  // inclusive scan within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(x, o, 64);
      if (lane >= o) x += t;
    }
    if (lane == 63) shw[w] = x;
    __syncthreads();
    int before = carry;
    for (int q = 0; q < w; ++q) before += shw[q];
    if (i < N) np[i] = before + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) np[N] = carry;
}

__global__ __launch_bounds__(256) void auroc_u_kernel(const float* __restrict__ keys, const unsigned char* __restrict__ flags,
                                                       const int* __restrict__ negpre, int N, float* __restrict__ auc) {
  __shared__ unsigned long long sh[4];
  const int c = blockIdx.x;
  const float* k = keys + (long)c * N;
  const unsigned char* f = flags + (long)c * N;
  const int* np = negpre + (long)c * (N + 1);
  unsigned long long u2 = 0;
  for (int i = threadIdx.x; i < N; i += 256) {
    if (!f[i]) continue;
    const float v = k[i];
    int lo = 0, hi = N;  // lower_bound
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (k[m] < v) lo = m + 1; else hi = m;
    }
    const int lb = lo;
    hi = N;  // upper_bound (search from lb)
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (k[m] <= v) lo = m + 1; else hi = m;
    }
    u2 += 2ULL * (unsigned long long)np[lb] + (unsigned long long)(np[lo] - np[lb]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) u2 += __shfl_xor(u2, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = u2;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long tot = sh[0] + sh[1] + sh[2] + sh[3];
    const double nneg = (double)np[N], npos = (double)(N - np[N]);
    // a class without positives or without negatives scores 0 (torchmetrics 0.11 returns an all-zero rate for it)
    auc[c] = (nneg > 0 && npos > 0) ? (float)((double)tot / (2.0 * npos * nneg)) : 0.f;
  }
}

__global__ void auroc_macro_kernel(const float* __restrict__ auc, int K, float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0;
  for (int c = 0; c < K; ++c) s += (double)auc[c];
  out[0] = (float)(s / K);
}

extern "C" int stil_auroc(const float* scores, int ld, const long long* target, int N, int K, float* auc_per_class, float* macro,
                          void* ws, size_t ws_bytes, void* stream) {
  STIL_REQUIRE(N >= 1 && K >= 1 && ld >= K, "auroc: bad shape N=%d K=%d ld=%d", N, K, ld);
  STIL_REQUIRE((long)N * K < (1L << 31), "auroc: N*K = %ld exceeds the 2^31 element limit of the segmented sort", (long)N * K);
  AurocWs w = auroc_layout(ws, N, K);
  STIL_REQUIRE(ws != nullptr && ws_bytes >= w.total, "auroc: workspace %zu < %zu bytes", ws_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  const long KN = (long)K * N;
  hipLaunchKernelGGL(auroc_scatter_kernel, dim3(cdiv(KN + 1, 256)), dim3(256), 0, st, scores, ld, target, N, K, K == 1 ? 1 : 0, w.kin,
                     w.fin, w.offs);
  STIL_LAUNCH_CHECK();
  size_t cb = w.cub_bytes;
  hipError_t e = hipcub::DeviceSegmentedRadixSort::SortPairs(w.cub, cb, (const float*)w.kin, w.kout, (const unsigned char*)w.fin, w.fout,
                                                             (int)KN, K, (const int*)w.offs, (const int*)w.offs + 1, 0, 32, st);
  if (e != hipSuccess) {
    stil_set_error("auroc: segmented sort failed: %s", hipGetErrorString(e));
    return STIL_EHIP;
  }
  hipLaunchKernelGGL(auroc_scan_kernel, dim3(K), dim3(1024), 0, st, (const unsigned char*)w.fout, N, w.negpre);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(auroc_u_kernel, dim3(K), dim3(256), 0, st, (const float*)w.kout, (const unsigned char*)w.fout, (const int*)w.negpre, N,
                     auc_per_class);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(auroc_macro_kernel, dim3(1), dim3(64), 0, st, (const float*)auc_per_class, K, macro);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
