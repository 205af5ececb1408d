// SAINT tabular encoder pieces (STiLModel_SAINT_backbone.py:159-184, SAINT/model_util.py:43-59,79-87,111-122):
// categorical embedding + positional encoding, the per-continuous-column simple_MLP(1 -> 100 -> d) that the reference
// runs as a Python loop over columns (one launch here), GEGLU, and a row softmax for the inter-sample (row) attention.
// Launch/HBM-bound work: deterministic batch-ordered reductions, no atomics.
#include "common.h"

// out[b, j, :] = embeds[idx(b,j)] + pos[j],  j < ncat1 = ncat + 1;  idx = (j ? int(x[b, cat_cols[j-1]]) : 0) + offs[j]
__global__ void saint_embed_fwd_kernel(const float* __restrict__ x, const int* __restrict__ cat_cols,
                                       const int* __restrict__ offs, const float* __restrict__ embeds,
                                       const float* __restrict__ pos, float* __restrict__ out, int B, int ncols, int ncat1,
                                       int nfeats, int d) {
  long total = (long)B * ncat1 * d;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % d);
    long t = i / d;
    int j = (int)(t % ncat1), b = (int)(t / ncat1);
    long idx = (j ? (long)x[(long)b * ncols + cat_cols[j - 1]] : 0) + offs[j];
    out[((long)b * nfeats + j) * d + c] = embeds[idx * d + c] + pos[(long)j * d + c];
  }
}
// block per embedding row r (owned by column rowcol[r]): d_embeds[r] (+)= sum_b [idx(b,j)==r] g[b,j,:]
__global__ void saint_embed_bwd_emb_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                           const int* __restrict__ cat_cols, const int* __restrict__ offs,
                                           const int* __restrict__ rowcol, float* __restrict__ demb, int B, int ncols,
                                           int nfeats, int d, int accumulate) {
  const int r = blockIdx.x, j = rowcol[r];
  const int code = r - offs[j];
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      int v = j ? (int)x[(long)b * ncols + cat_cols[j - 1]] : 0;
      if (v == code) s += g[((long)b * nfeats + j) * d + c];
    }
    long o = (long)r * d + c;
    demb[o] = accumulate ? demb[o] + s : s;
  }
}
// d_pos[j, :] (+)= sum_b g[b, j, :]
__global__ void saint_embed_bwd_pos_kernel(const float* __restrict__ g, float* __restrict__ dpos, int B, int nfeats, int d,
                                           int accumulate) {
  const int j = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g[((long)b * nfeats + j) * d + c];
    long o = (long)j * d + c;
    dpos[o] = accumulate ? dpos[o] + s : s;
  }
}

// per continuous column j (block j): out[b, tok0+j, :] = W2_j relu(w1_j * x[b, con_cols[j]] + b1_j) + b2_j
// ptrs[j*4 + {0,1,2,3}] = w1 [hid], b1 [hid], W2 [d, hid], b2 [d]
__global__ __launch_bounds__(256) void colmlp_fwd_kernel(const float* __restrict__ x, const int* __restrict__ con_cols,
                                                          const float* const* __restrict__ ptrs, float* __restrict__ out,
                                                          int B, int ncols, int nfeats, int tok0, int hid, int d) {
  extern __shared__ float sh[];  // W2 [d*hid], w1[hid], b1[hid], b2[d]
  const int j = blockIdx.x;
  const float* w1 = ptrs[j * 4 + 0]; const float* b1 = ptrs[j * 4 + 1];
  const float* W2 = ptrs[j * 4 + 2]; const float* b2 = ptrs[j * 4 + 3];
  float* sW2 = sh; float* sw1 = sh + d * hid; float* sb1 = sw1 + hid; float* sb2 = sb1 + hid;
  for (int i = threadIdx.x; i < d * hid; i += 256) sW2[i] = W2[i];
  for (int i = threadIdx.x; i < hid; i += 256) { sw1[i] = w1[i]; sb1[i] = b1[i]; }
  for (int i = threadIdx.x; i < d; i += 256) sb2[i] = b2[i];
  __syncthreads();
  const int col = con_cols[j];
  for (int e = threadIdx.x; e < B * d; e += 256) {
    int b = e / d, o = e - b * d;
    const float xv = x[(long)b * ncols + col];
    float s = sb2[o];
    for (int k = 0; k < hid; ++k) s += sW2[o * hid + k] * fmaxf(sw1[k] * xv + sb1[k], 0.f);
    out[((long)b * nfeats + tok0 + j) * d + o] = s;
  }
}
// gptrs like ptrs but for the gradients (+=, slots pre-zeroed or accumulate = 0 -> overwrite)
__global__ __launch_bounds__(256) void colmlp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                          const int* __restrict__ con_cols,
                                                          const float* const* __restrict__ ptrs,
                                                          float* const* __restrict__ gptrs, int B, int ncols, int nfeats,
                                                          int tok0, int hid, int d, int accumulate) {
  extern __shared__ float sh[];  // W2 [d*hid], w1[hid], b1[hid], gb[d], dh[hid], hv[hid]
  const int j = blockIdx.x, tid = threadIdx.x;
  const float* w1 = ptrs[j * 4 + 0]; const float* b1 = ptrs[j * 4 + 1]; const float* W2 = ptrs[j * 4 + 2];
  float* sW2 = sh; float* sw1 = sh + d * hid; float* sb1 = sw1 + hid; float* gb = sb1 + hid; float* dh = gb + d; float* hv = dh + hid;
  for (int i = tid; i < d * hid; i += 256) sW2[i] = W2[i];
  for (int i = tid; i < hid; i += 256) { sw1[i] = w1[i]; sb1[i] = b1[i]; }
  __syncthreads();
  const int col = con_cols[j];
  // each thread owns a fixed set of dW2 entries (o,k) and, for tid < hid, dw1[tid]/db1[tid]; tid < d: db2[tid]
  constexpr int MAXE = 16;
  float aw2[MAXE];
  const int ne = (d * hid + 255) / 256;
#pragma unroll
  for (int q = 0; q < MAXE; ++q) aw2[q] = 0.f;
  float aw1 = 0.f, ab1 = 0.f, ab2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float xv = x[(long)b * ncols + col];
    if (tid < d) gb[tid] = g[((long)b * nfeats + tok0 + j) * d + tid];
    if (tid < hid) hv[tid] = fmaxf(sw1[tid] * xv + sb1[tid], 0.f);
    __syncthreads();
    if (tid < hid) {
      float s = 0.f;
      for (int o = 0; o < d; ++o) s += sW2[o * hid + tid] * gb[o];
      s = hv[tid] > 0.f ? s : 0.f;
      aw1 += s * xv; ab1 += s;
    }
    if (tid < d) ab2 += gb[tid];
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
      int e = tid + q * 256;
      if (q < ne && e < d * hid) { int o = e / hid, k = e - o * hid; aw2[q] += gb[o] * hv[k]; }
    }
    __syncthreads();
  }
  float* gw1 = gptrs[j * 4 + 0]; float* gb1 = gptrs[j * 4 + 1]; float* gW2 = gptrs[j * 4 + 2]; float* gb2 = gptrs[j * 4 + 3];
  if (tid < hid) { gw1[tid] = (accumulate ? gw1[tid] : 0.f) + aw1; gb1[tid] = (accumulate ? gb1[tid] : 0.f) + ab1; }
  if (tid < d) gb2[tid] = (accumulate ? gb2[tid] : 0.f) + ab2;
#pragma unroll
  for (int q = 0; q < MAXE; ++q) {
    int e = tid + q * 256;
    if (q < ne && e < d * hid) gW2[e] = (accumulate ? gW2[e] : 0.f) + aw2[q];
  }
}

// GEGLU: h [R, 2H] -> out [R, H] = h[:, :H] * gelu(h[:, H:])
__global__ void geglu_fwd_kernel(const float* __restrict__ h, float* __restrict__ out, long R, int H) {
  long total = R * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / H; int c = (int)(i - r * H);
    out[i] = h[r * 2 * H + c] * gelu_f(h[r * 2 * H + H + c]);
  }
}
__global__ void geglu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h, float* __restrict__ dh, long R,
                                 int H) {
  long total = R * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / H; int c = (int)(i - r * H);
    float a = h[r * 2 * H + c], gt = h[r * 2 * H + H + c], gg = g[i];
    dh[r * 2 * H + c] = gg * gelu_f(gt);
    dh[r * 2 * H + H + c] = gg * a * gelu_grad_f(gt);
  }
}

// row softmax of [R, C] (in: logits, out: probabilities) and its backward dS = P * (dP - rowsum(dP * P))
__global__ __launch_bounds__(256) void row_softmax_fwd_kernel(const float* __restrict__ z, float* __restrict__ p, int C) {
  __shared__ float red[16];
  const float* zr = z + (long)blockIdx.x * C;
  float* pr = p + (long)blockIdx.x * C;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < C; k += 256) m = fmaxf(m, zr[k]);
  m = block_max(m, red);
  float s = 0.f;
  for (int k = threadIdx.x; k < C; k += 256) s += expf(zr[k] - m);
  s = block_sum(s, red);
  const float inv = 1.f / s;
  for (int k = threadIdx.x; k < C; k += 256) pr[k] = expf(zr[k] - m) * inv;
}
__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ p,
                                                               float* __restrict__ dz, int C) {
  __shared__ float red[16];
  const float* gr = g + (long)blockIdx.x * C;
  const float* pr = p + (long)blockIdx.x * C;
  float s = 0.f;
  for (int k = threadIdx.x; k < C; k += 256) s += gr[k] * pr[k];
  s = block_sum(s, red);
  for (int k = threadIdx.x; k < C; k += 256) dz[(long)blockIdx.x * C + k] = pr[k] * (gr[k] - s);
}

// ---------------------------------------------------------------- C ABI
static inline int ew_grid3(long n) { long g = (n + 255) / 256; return (int)(g < 16384 ? (g < 1 ? 1 : g) : 16384); }

extern "C" int stil_saint_embed_fwd(const float* x, const int* cat_cols, const int* cat_offsets, const float* embeds,
                                    const float* pos_enc, float* out, int B, int ncols, int ncat, int nfeats, int d,
                                    void* stream) {
  STIL_REQUIRE(x && cat_offsets && embeds && pos_enc && out && (ncat == 0 || cat_cols), "stil_saint_embed_fwd: null pointer");
  long total = (long)B * (ncat + 1) * d;
  hipLaunchKernelGGL(saint_embed_fwd_kernel, dim3(ew_grid3(total)), dim3(256), 0, (hipStream_t)stream, x, cat_cols,
                     cat_offsets, embeds, pos_enc, out, B, ncols, ncat + 1, nfeats, d);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_saint_embed_bwd(const float* g, const float* x, const int* cat_cols, const int* cat_offsets,
                                    const int* rowcol, int n_emb_rows, float* d_embeds, float* d_pos, int B, int ncols,
                                    int ncat, int nfeats, int d, int accumulate, void* stream) {
  STIL_REQUIRE(g && x && cat_offsets && rowcol && d_embeds && d_pos, "stil_saint_embed_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(saint_embed_bwd_emb_kernel, dim3(n_emb_rows), dim3(64), 0, s, g, x, cat_cols, cat_offsets, rowcol,
                     d_embeds, B, ncols, nfeats, d, accumulate);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(saint_embed_bwd_pos_kernel, dim3(ncat + 1), dim3(64), 0, s, g, d_pos, B, nfeats, d, accumulate);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// param_ptrs / grad_ptrs: device arrays of ncon*4 pointers {w1 [hid], b1 [hid], W2 [d,hid], b2 [d]} per column
extern "C" int stil_colmlp_fwd(const float* x, const int* con_cols, const float* const* param_ptrs, float* out, int B,
                               int ncols, int ncon, int nfeats, int tok0, int hid, int d, void* stream) {
  STIL_REQUIRE(x && con_cols && param_ptrs && out && ncon > 0, "stil_colmlp_fwd: bad arguments");
  size_t lds = ((size_t)d * hid + 2 * hid + d) * sizeof(float);
  STIL_REQUIRE(lds <= 60 * 1024, "stil_colmlp_fwd: hidden %d x dim %d does not fit LDS", hid, d);
  hipLaunchKernelGGL(colmlp_fwd_kernel, dim3(ncon), dim3(256), lds, (hipStream_t)stream, x, con_cols, param_ptrs, out, B,
                     ncols, nfeats, tok0, hid, d);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_colmlp_bwd(const float* g, const float* x, const int* con_cols, const float* const* param_ptrs,
                               float* const* grad_ptrs, int B, int ncols, int ncon, int nfeats, int tok0, int hid, int d,
                               int accumulate, void* stream) {
  STIL_REQUIRE(g && x && con_cols && param_ptrs && grad_ptrs && ncon > 0, "stil_colmlp_bwd: bad arguments");
  STIL_REQUIRE(d * hid <= 16 * 256 && hid <= 256 && d <= 256, "stil_colmlp_bwd: hidden %d x dim %d too large", hid, d);
  size_t lds = ((size_t)d * hid + 4 * hid + d) * sizeof(float);
  hipLaunchKernelGGL(colmlp_bwd_kernel, dim3(ncon), dim3(256), lds, (hipStream_t)stream, g, x, con_cols, param_ptrs,
                     grad_ptrs, B, ncols, nfeats, tok0, hid, d, accumulate);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_geglu_fwd(const float* h, float* out, long rows, int H, void* stream) {
  STIL_REQUIRE(h && out && rows > 0 && H > 0, "stil_geglu_fwd: bad arguments");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_grid3(rows * H)), dim3(256), 0, (hipStream_t)stream, h, out, rows, H);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_geglu_bwd(const float* g, const float* h, float* dh, long rows, int H, void* stream) {
  STIL_REQUIRE(g && h && dh && rows > 0 && H > 0, "stil_geglu_bwd: bad arguments");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_grid3(rows * H)), dim3(256), 0, (hipStream_t)stream, g, h, dh, rows, H);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_row_softmax_fwd(const float* z, float* p, int rows, int C, void* stream) {
  STIL_REQUIRE(z && p && rows > 0 && C > 0, "stil_row_softmax_fwd: bad arguments");
  hipLaunchKernelGGL(row_softmax_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, z, p, C);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_row_softmax_bwd(const float* g, const float* p, float* dz, int rows, int C, void* stream) {
  STIL_REQUIRE(g && p && dz && rows > 0 && C > 0, "stil_row_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, p, dz, C);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
