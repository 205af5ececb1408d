// Loss tail of STiLModel.training_step as fused row kernels (one block per sample row,
// wave-shuffle + LDS reductions, fixed-order sums => bit-stable across runs):
//   hard / soft-target cross-entropy      STiLModel.py:284,301-303
//   CGPL case partition + PGLS smoothing  STiLModel.py:259-299
//   CLIP (InfoNCE) LSE + dZ               utils/clip_loss.py:27-39
//   CLUBMean closed form                  models/Disentangle/utils/club.py:107-130
//   PrototypeLoss                         utils/prototype_loss.py:24-40
//   prototype class sums                  STiLModel.py:199-226
#include "common.h"

// ---------------------------------------------------------------- cross entropy
// hard labels: row_loss[r] = lse - z[y];  dlogits[r,k] = (softmax - onehot) * inv_rows
__global__ __launch_bounds__(256) void ce_hard_kernel(const float* __restrict__ z, int ld, const long long* __restrict__ y,
                                                       float* __restrict__ row_loss, float* __restrict__ dz, int ldd,
                                                       int K, float inv_rows) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* zr = z + (long)r * ld;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, zr[k]);
  m = block_max(m, red);
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += expf(zr[k] - m);
  s = block_sum(s, red);
  const float lse = m + logf(s);
  const int yy = (int)y[r];
  if (threadIdx.x == 0) row_loss[r] = lse - zr[yy];
  if (dz)
    for (int k = threadIdx.x; k < K; k += 256)
      dz[(long)r * ldd + k] = (expf(zr[k] - lse) - (k == yy ? 1.f : 0.f)) * inv_rows;
}

// probability targets with a per-row weight w[r]:
//   row_loss[r] = w * (-(sum_k q_k * (z_k - lse)));  dz = w * inv_rows * (softmax * sum(q) - q)
__global__ __launch_bounds__(256) void ce_soft_kernel(const float* __restrict__ z, int ld, const float* __restrict__ q,
                                                       int ldq, const float* __restrict__ w, float* __restrict__ row_loss,
                                                       float* __restrict__ dz, int ldd, int K, float inv_rows) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* zr = z + (long)r * ld;
  const float* qr = q + (long)r * ldq;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, zr[k]);
  m = block_max(m, red);
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += expf(zr[k] - m);
  s = block_sum(s, red);
  const float lse = m + logf(s);
  float a = 0.f, qs = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) { a += qr[k] * (zr[k] - lse); qs += qr[k]; }
  a = block_sum(a, red);
  qs = block_sum(qs, red);
  const float wr = w ? w[r] : 1.f;
  if (threadIdx.x == 0) row_loss[r] = -a * wr;
  if (dz)
    for (int k = threadIdx.x; k < K; k += 256)
      dz[(long)r * ldd + k] = wr * inv_rows * (expf(zr[k] - lse) * qs - qr[k]);
}

// out[0] (+)= scale * sum_i x[i]   single block, fixed order
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, int n, float scale,
                                                          float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}

// out = x * (*g) * c
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ g, float c,
                                 float* __restrict__ out, long n) {
  const float s = g[0] * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = x[i] * s;
}

// ---------------------------------------------------------------- F.normalize(dim=1), eps 1e-12
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ nrm, int R, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + (long)r * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i] * xr[i];
  const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int i = lane; i < D; i += 64) y[(long)r * D + i] = xr[i] / n;
  if (lane == 0) nrm[r] = n;
}
// dx = (g - y * (y . g)) / n
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                          const float* __restrict__ nrm, float* __restrict__ dx, int R,
                                                          int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* gr = g + (long)r * D;
  const float* yr = y + (long)r * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += gr[i] * yr[i];
  s = wave_sum(s);
  const float inv = 1.f / nrm[r];
  for (int i = lane; i < D; i += 64) dx[(long)r * D + i] = (gr[i] - yr[i] * s) * inv;
}

// ---------------------------------------------------------------- CLIP: row / column log-sum-exp of Z [B,B]
// In DOUBLE.  The gradient dZ = g/B (lam0 (softmax_rows - I) + lam1 (softmax_cols - I)) sums to exactly zero over the whole
// matrix, and the projector-bias gradient downstream is the image of that grand total through the COMMON component of the
// embeddings (every row's dZ . n1 carries the batch-mean embedding): an error shared by all rows does not average out, it is
// multiplied by the batch size.  With lse = m + logf(s) in fp32 every softmax row came out scaled by (1 - eta) with one eta of
// ~0.6 ulp(lse) = 1.5e-7 for all rows and columns; elementwise that is 4.5e-8 of |dZ| -- and 2e-4 of projector_imaging.bias'
// gradient (15x what ATen's fp32 path leaves there; tests/tools/itc_ladder.py, itc_noise.py, round-5 experiment log).  The
// log-sum-exps and the probabilities are therefore formed in double and dZ is rounded ONCE: its error is the half-ulp of the
// fp32 it is stored in (zero-mean, independent per element).  B^2 double exponentials: 65 k at B = 256, noise beside the step.
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += sh[i];  // fixed order: deterministic
  return r;
}
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ Z, double* __restrict__ lse, int R, int C) {
  __shared__ float red[16];
  __shared__ double redd[16];
  const float* zr = Z + (long)blockIdx.x * C;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < C; k += 256) m = fmaxf(m, zr[k]);
  m = block_max(m, red);
  double s = 0.0;
  for (int k = threadIdx.x; k < C; k += 256) s += exp((double)zr[k] - (double)m);
  s = block_sum_d(s, redd);
  if (threadIdx.x == 0) lse[blockIdx.x] = (double)m + log(s);
}
__global__ void lse_cols_kernel(const float* __restrict__ Z, double* __restrict__ lse, int R, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m = -INFINITY;
  for (int r = 0; r < R; ++r) m = fmaxf(m, Z[(long)r * C + c]);
  double s = 0.0;
  for (int r = 0; r < R; ++r) s += exp((double)Z[(long)r * C + c] - (double)m);
  lse[c] = (double)m + log(s);
}
// row_term[i] = lam0*(lse_r[i]-Z[i][i]) + lam1*(lse_c[i]-Z[i][i])
__global__ void clip_terms_kernel(const float* __restrict__ Z, const double* __restrict__ lse_r,
                                  const double* __restrict__ lse_c, float* __restrict__ terms, int B, float lam0,
                                  float lam1) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const double d = (double)Z[(long)i * B + i];
  terms[i] = (float)((double)lam0 * (lse_r[i] - d) + (double)lam1 * (lse_c[i] - d));
}
// dZ[r,c] = g * ( lam0/B (exp(z-lse_r[r]) - I) + lam1/B (exp(z-lse_c[c]) - I) ), formed in double, rounded once
__global__ void clip_dz_kernel(const float* __restrict__ Z, const double* __restrict__ lse_r,
                               const double* __restrict__ lse_c, const float* __restrict__ g, float* __restrict__ dZ,
                               int B, float lam0, float lam1) {
  long n = (long)B * B;
  const double gg = (double)g[0] / (double)B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / B), c = (int)(i - (long)r * B);
    const double z = (double)Z[i], e = (r == c) ? 1.0 : 0.0;
    dZ[i] = (float)(gg * ((double)lam0 * (exp(z - lse_r[r]) - e) + (double)lam1 * (exp(z - lse_c[c]) - e)));
  }
}

// ---------------------------------------------------------------- CLUB (closed form, centred)
// t[i] = sum_d (y-ybar)^2 + (ybar-mu)^2 - (mu-y)^2 ;  e[i] = sum_d (mu-y)^2
__global__ __launch_bounds__(256) void club_rows_kernel(const float* __restrict__ mu, const float* __restrict__ y,
                                                         const float* __restrict__ ybar, float* __restrict__ t,
                                                         float* __restrict__ e, int R, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  float a = 0.f, b = 0.f;
  for (int i = lane; i < D; i += 64) {
    float m = mu[(long)r * D + i], yy = y[(long)r * D + i], yb = ybar[i];
    float d0 = yy - yb, d1 = yb - m, d2 = m - yy;
    a += d0 * d0 + d1 * d1 - d2 * d2;
    b += d2 * d2;
  }
  a = wave_sum(a); b = wave_sum(b);
  if (lane == 0) { t[r] = a; e[r] = b; }
}
// dmu = gc (y - ybar)/B + ge 2(mu-y)/B ;  dy = gc (mu - mubar)/B - ge 2(mu-y)/B
__global__ void club_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ y,
                                const float* __restrict__ ybar, const float* __restrict__ mubar,
                                const float* __restrict__ gc, const float* __restrict__ ge, float* __restrict__ dmu,
                                float* __restrict__ dy, int R, int D) {
  long n = (long)R * D;
  const float c = gc[0] / (float)R, e2 = 2.f * ge[0] / (float)R;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int d = (int)(i % D);
    float m = mu[i], yy = y[i];
    dmu[i] = c * (yy - ybar[d]) + e2 * (m - yy);
    dy[i] = c * (m - mubar[d]) - e2 * (m - yy);
  }
}

// ---------------------------------------------------------------- CGPL + PGLS (teacher side, no grad)
struct CgplArgs {
  const float* zm; const float* zi; const float* zt; int ldz;   // teacher logits of the unlabelled rows
  const float* feat; int Dp;                                      // teacher multimodal embedding (normalised)
  const float* protos;                                            // [K, Dp]
  const unsigned char* mask_random;                               // [Bu]
  float* pseudo_label;   // [Bu,K]  r*q0 + (1-r)*tp
  float* pseudo_orig;    // [Bu,K]  q0 (optional)
  float* prediction;     // [Bu,K]  r*softmax(zm) + (1-r)*tp, zeroed unless use_pseudo
  unsigned char* flags;  // [Bu,4]  case id (1..4), mask1, conf, 0
  int* hard;             // [Bu] argmax of `prediction`
  float* w3;             // [3,Bu] row weights of the three unlabelled CE terms (m, i, t)
  const float* pred_in;  // optional [Bu,K]: distribution-aligned prediction replacing softmax(zm) (STiLModel.py:276-277)
  int K; float r, T, th; int use_pseudo;
};

__global__ __launch_bounds__(256) void cgpl_pgls_kernel(CgplArgs p) {
  extern __shared__ float sh[];  // [3][K] : case logits -> q0 ; zm -> pred ; tl -> tp
  __shared__ float red[16];
  __shared__ int ired[3];
  float* A = sh; float* Bm = sh + p.K; float* Ct = sh + 2 * p.K;
  const int u = blockIdx.x, K = p.K, tid = threadIdx.x;
  const float* zm = p.zm + (long)u * p.ldz;
  const float* zi = p.zi + (long)u * p.ldz;
  const float* zt = p.zt + (long)u * p.ldz;
  // top-1 of each head = argmax(softmax(z)), NOT argmax(z) (STiLModel.py:262-263): distinct logits whose probabilities
  // round to the same fp32 value tie, and torch.argmax then returns the FIRST of the tied indices.  Wave w (0..2) owns
  // head w: p_k = exp(z_k - max) / sum in fp32, per-lane running best (strictly greater -> earliest k of that lane),
  // then a (value, index) reduction that prefers the smaller index among equal values.
  if (tid < 192) {
    const int h = tid >> 6, lane = tid & 63;
    const float* z = h == 0 ? zm : (h == 1 ? zi : zt);
    float mxz = -INFINITY;
    for (int k = lane; k < K; k += 64) mxz = fmaxf(mxz, z[k]);
    mxz = wave_max(mxz);
    float se = 0.f;
    for (int k = lane; k < K; k += 64) se += expf(z[k] - mxz);
    se = wave_sum(se);
    float best = -1.f; int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
      const float pk = expf(z[k] - mxz) / se;
      if (pk > best || pk != pk) { best = pk; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) ired[h] = bi;
  }
  __syncthreads();
  const int a = ired[0], b = ired[1], d = ired[2];
  int cs;
  if (a == b && a == d) cs = 1; else if (a == b) cs = 2; else if (a == d) cs = 3; else cs = 4;
  for (int k = tid; k < K; k += 256) {
    float m = zm[k], v;
    if (cs == 1) v = (m + zi[k] + zt[k]) / 3.0f;
    else if (cs == 2) v = (m + zi[k]) / 2.0f;
    else if (cs == 3) v = (m + zt[k]) / 2.0f;
    else v = m;
    A[k] = v; Bm[k] = m;
    const float* pr = p.protos + (long)k * p.Dp;
    const float* f = p.feat + (long)u * p.Dp;
    float s = 0.f;
    for (int c = 0; c < p.Dp; ++c) s += f[c] * pr[c];
    Ct[k] = s / p.T;
  }
  __syncthreads();
  // three softmaxes in place
  for (int which = 0; which < 3; ++which) {
    float* v = sh + which * K;
    float m = -INFINITY;
    for (int k = tid; k < K; k += 256) m = fmaxf(m, v[k]);
    m = block_max(m, red);
    float s = 0.f;
    for (int k = tid; k < K; k += 256) { float e = expf(v[k] - m); v[k] = e; s += e; }
    s = block_sum(s, red);
    for (int k = tid; k < K; k += 256) v[k] = v[k] / s;
    __syncthreads();
  }
  float mx = -INFINITY;
  for (int k = tid; k < K; k += 256) {
    float tp = Ct[k];
    float pl = p.r * A[k] + (1.f - p.r) * tp;
    const float pm = p.pred_in ? p.pred_in[(long)u * K + k] : Bm[k];
    float pd = p.r * pm + (1.f - p.r) * tp;
    p.pseudo_label[(long)u * K + k] = pl;
    if (p.pseudo_orig) p.pseudo_orig[(long)u * K + k] = A[k];
    p.prediction[(long)u * K + k] = p.use_pseudo ? pd : 0.f;
    Bm[k] = pd;
    mx = fmaxf(mx, pd);
  }
  mx = block_max(mx, red);
  __syncthreads();
  if (tid == 0) {
    int bi = 0;
    if (p.use_pseudo) { float best = Bm[0]; for (int k = 1; k < K; ++k) if (Bm[k] > best) { best = Bm[k]; bi = k; } }
    const int mask1 = mx >= p.th ? 1 : 0;
    const int mr = p.mask_random[u] ? 1 : 0;
    p.flags[u * 4 + 0] = (unsigned char)cs;
    p.flags[u * 4 + 1] = (unsigned char)mask1;
    p.flags[u * 4 + 2] = (unsigned char)((p.use_pseudo && mask1) ? 1 : 0);  // confident for prototypes
    p.flags[u * 4 + 3] = 0;
    p.hard[u] = bi;
    const int Bu = gridDim.x;
    p.w3[u] = (mask1 && cs == 1) ? 1.f : 0.f;
    p.w3[Bu + u] = (mask1 && (cs == 1 || cs == 3 || (cs == 4 && mr))) ? 1.f : 0.f;
    p.w3[2 * Bu + u] = (mask1 && (cs == 1 || cs == 2 || (cs == 4 && !mr))) ? 1.f : 0.f;
  }
}

// distribution alignment (STiLModel.py:171-180): out = (p / qmean) renormalised per row
__global__ __launch_bounds__(256) void da_apply_kernel(const float* __restrict__ p, const float* __restrict__ qmean,
                                                        float* __restrict__ out, int K) {
  __shared__ float red[16];
  const float* pr = p + (long)blockIdx.x * K;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += pr[k] / qmean[k];
  s = block_sum(s, red);
  for (int k = threadIdx.x; k < K; k += 256) out[(long)blockIdx.x * K + k] = (pr[k] / qmean[k]) / s;
}

// ---------------------------------------------------------------- PrototypeLoss
// row_loss = -log(softmax(feat.P^T/T)[hard] + 1e-7) * conf ; dfeat_unit = conf/B * sum_k dz_k P_k / T
__global__ __launch_bounds__(256) void proto_loss_kernel(const float* __restrict__ feat, const float* __restrict__ protos,
                                                          const int* __restrict__ hard,
                                                          const unsigned char* __restrict__ conf,
                                                          float* __restrict__ row_loss, float* __restrict__ dfeat, int K,
                                                          int Dp, float T, float inv_rows) {
  extern __shared__ float sh[];  // [K]
  __shared__ float red[16];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* f = feat + (long)r * Dp;
  for (int k = tid; k < K; k += 256) {
    const float* pr = protos + (long)k * Dp;
    float s = 0.f;
    for (int c = 0; c < Dp; ++c) s += f[c] * pr[c];
    sh[k] = s / T;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int k = tid; k < K; k += 256) m = fmaxf(m, sh[k]);
  m = block_max(m, red);
  float s = 0.f;
  for (int k = tid; k < K; k += 256) { float e = expf(sh[k] - m); sh[k] = e; s += e; }
  s = block_sum(s, red);
  for (int k = tid; k < K; k += 256) sh[k] = sh[k] / s;
  __syncthreads();
  const int h = hard[r];
  const float cf = conf[r] ? 1.f : 0.f;
  const float sh_h = sh[h];
  if (tid == 0) row_loss[r] = -logf(sh_h + 1e-7f) * cf;
  if (dfeat) {
    const float coef = -(sh_h / (sh_h + 1e-7f)) * cf * inv_rows / T;
    for (int c = tid; c < Dp; c += 256) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += ((k == h ? 1.f : 0.f) - sh[k]) * protos[(long)k * Dp + c];
      dfeat[(long)r * Dp + c] = coef * acc;
    }
  }
}

// class_sum[k, 0:Dp] = sum_{r: conf, hard==k} wt_r * feat[r] ; class_sum[k, Dp] = sum wt_r  (block per class)
__global__ void proto_accum_kernel(const float* __restrict__ feat, const int* __restrict__ hard,
                                   const unsigned char* __restrict__ conf, float* __restrict__ out, int B, int B_l,
                                   int Dp, float inv_repeat) {
  const int k = blockIdx.x;
  for (int c = threadIdx.x; c <= Dp; c += blockDim.x) {
    float sl = 0.f, su = 0.f;   // the reference sums labelled and unlabelled parts separately, then l/repeat + u
    for (int r = 0; r < B; ++r) {
      if (!conf[r] || hard[r] != k) continue;
      float v = (c < Dp) ? feat[(long)r * Dp + c] : 1.f;
      if (r < B_l) sl += v; else su += v;
    }
    out[(long)k * (Dp + 1) + c] = sl * inv_repeat + su;
  }
}
// prototypes_sum += cs[:, :Dp] ; prototypes_count_sum += cs[:, Dp]
__global__ void proto_add_kernel(const float* __restrict__ cs, float* __restrict__ psum, float* __restrict__ pcnt, int K,
                                 int Dp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * (Dp + 1)) return;
  int k = i / (Dp + 1), c = i - k * (Dp + 1);
  if (c < Dp) psum[(long)k * Dp + c] += cs[i]; else pcnt[k] += cs[i];
}
// epoch end: prototypes = sum / count ; zero accumulators ; flag[0] = #classes with count < 1
__global__ void proto_commit_kernel(float* __restrict__ protos, float* __restrict__ psum, float* __restrict__ pcnt,
                                    int* __restrict__ bad, int K, int Dp) {
  int k = blockIdx.x;
  float cnt = pcnt[k];
  if (cnt < 1.f) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
  for (int c = threadIdx.x; c < Dp; c += blockDim.x) { protos[(long)k * Dp + c] = psum[(long)k * Dp + c] / cnt; }
}

// ---------------------------------------------------------------- C ABI
static inline int ew_grid2(long n) { long g = (n + 255) / 256; return (int)(g < 16384 ? (g < 1 ? 1 : g) : 16384); }

extern "C" int stil_ce_hard(const float* logits, int ld, const long long* labels, float* row_loss, float* dlogits,
                            int ldd, int rows, int K, float inv_rows, void* stream) {
  STIL_REQUIRE(logits && labels && row_loss && rows > 0 && K > 0, "stil_ce_hard: bad arguments");
  hipLaunchKernelGGL(ce_hard_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, row_loss, dlogits,
                     ldd, K, inv_rows);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_ce_soft(const float* logits, int ld, const float* targets, int ldq, const float* row_w,
                            float* row_loss, float* dlogits, int ldd, int rows, int K, float inv_rows, void* stream) {
  STIL_REQUIRE(logits && targets && row_loss && rows > 0 && K > 0, "stil_ce_soft: bad arguments");
  hipLaunchKernelGGL(ce_soft_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, ldq, row_w,
                     row_loss, dlogits, ldd, K, inv_rows);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_reduce_sum(const float* x, int n, float scale, float* out, int accumulate, void* stream) {
  STIL_REQUIRE(x && out && n >= 0, "stil_reduce_sum: bad arguments");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, scale, out, accumulate);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_scale_dev(const float* x, const float* g_dev, float c, float* out, long n, void* stream) {
  STIL_REQUIRE(x && g_dev && out, "stil_scale_dev: null pointer");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_grid2(n)), dim3(256), 0, (hipStream_t)stream, x, g_dev, c, out, n);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_l2norm_fwd(const float* x, float* y, float* norms, int rows, int D, void* stream) {
  STIL_REQUIRE(x && y && norms && rows > 0, "stil_l2norm_fwd: bad arguments");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, norms, rows, D);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_l2norm_bwd(const float* g, const float* y, const float* norms, float* dx, int rows, int D,
                               void* stream) {
  STIL_REQUIRE(g && y && norms && dx && rows > 0, "stil_l2norm_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, g, y, norms, dx, rows, D);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// Z: [B,B] logits (already / T). lse: [2,B] DOUBLES (rows, cols). terms: [B] scratch. loss: scalar.
extern "C" int stil_clip_fwd(const float* Z, double* lse, float* terms, float* loss, int B, float lam0, float lam1,
                             void* stream) {
  STIL_REQUIRE(Z && lse && terms && loss && B > 0, "stil_clip_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lse_rows_kernel, dim3(B), dim3(256), 0, s, Z, lse, B, B);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(lse_cols_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, Z, lse + B, B, B);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(clip_terms_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, Z, lse, lse + B, terms, B, lam0, lam1);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, terms, B, 1.f / (float)B, loss, 0);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_clip_bwd(const float* Z, const double* lse, const float* g_dev, float* dZ, int B, float lam0,
                             float lam1, void* stream) {
  STIL_REQUIRE(Z && lse && g_dev && dZ && B > 0, "stil_clip_bwd: bad arguments");
  hipLaunchKernelGGL(clip_dz_kernel, dim3(ew_grid2((long)B * B)), dim3(256), 0, (hipStream_t)stream, Z, lse, lse + B,
                     g_dev, dZ, B, lam0, lam1);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// ybar must hold the column means of y.  rows_tmp: [2,R] scratch.  out: club -> out2[0], est -> out2[1]
extern "C" int stil_club_fwd(const float* mu, const float* y, const float* ybar, float* rows_tmp, float* out2, int R,
                             int D, void* stream) {
  STIL_REQUIRE(mu && y && ybar && rows_tmp && out2 && R > 0, "stil_club_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(club_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, mu, y, ybar, rows_tmp, rows_tmp + R, R, D);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, rows_tmp, R, 0.5f / (float)R, out2, 0);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, rows_tmp + R, R, 1.f / (float)R, out2 + 1, 0);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_club_bwd(const float* mu, const float* y, const float* ybar, const float* mubar,
                             const float* g_club_dev, const float* g_est_dev, float* dmu, float* dy, int R, int D,
                             void* stream) {
  STIL_REQUIRE(mu && y && ybar && mubar && g_club_dev && g_est_dev && dmu && dy, "stil_club_bwd: null pointer");
  hipLaunchKernelGGL(club_bwd_kernel, dim3(ew_grid2((long)R * D)), dim3(256), 0, (hipStream_t)stream, mu, y, ybar, mubar,
                     g_club_dev, g_est_dev, dmu, dy, R, D);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_cgpl_pgls(const float* zm, const float* zi, const float* zt, int ldz, const float* feat_u,
                              const float* prototypes, const unsigned char* mask_random, float* pseudo_label,
                              float* pseudo_orig, float* prediction, unsigned char* flags, int* hard, float* w3,
                              const float* pred_in, int Bu, int K, int Dp, float rate_pseudo, float T, float th,
                              int use_pseudo, void* stream) {
  STIL_REQUIRE(zm && zi && zt && feat_u && prototypes && mask_random && pseudo_label && prediction && flags && hard && w3,
               "stil_cgpl_pgls: null pointer");
  STIL_REQUIRE(Bu > 0 && K > 0 && (size_t)3 * K * sizeof(float) <= 60 * 1024, "stil_cgpl_pgls: bad shape (Bu=%d K=%d)", Bu, K);
  CgplArgs p;
  p.zm = zm; p.zi = zi; p.zt = zt; p.ldz = ldz; p.feat = feat_u; p.Dp = Dp; p.protos = prototypes;
  p.mask_random = mask_random; p.pseudo_label = pseudo_label; p.pseudo_orig = pseudo_orig; p.prediction = prediction;
  p.flags = flags; p.hard = hard; p.w3 = w3; p.pred_in = pred_in; p.K = K; p.r = rate_pseudo; p.T = T; p.th = th; p.use_pseudo = use_pseudo;
  hipLaunchKernelGGL(cgpl_pgls_kernel, dim3(Bu), dim3(256), (size_t)3 * K * sizeof(float), (hipStream_t)stream, p);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_da_apply(const float* probs, const float* queue_mean, float* out, int rows, int K, void* stream) {
  STIL_REQUIRE(probs && queue_mean && out && rows > 0 && K > 0, "stil_da_apply: bad arguments");
  hipLaunchKernelGGL(da_apply_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, probs, queue_mean, out, K);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_proto_loss(const float* feat, const float* prototypes, const int* hard, const unsigned char* conf,
                               float* row_loss, float* dfeat_unit, int rows, int K, int Dp, float T, void* stream) {
  STIL_REQUIRE(feat && prototypes && hard && conf && row_loss && rows > 0, "stil_proto_loss: bad arguments");
  STIL_REQUIRE((size_t)K * sizeof(float) <= 60 * 1024, "stil_proto_loss: K=%d too large", K);
  hipLaunchKernelGGL(proto_loss_kernel, dim3(rows), dim3(256), (size_t)K * sizeof(float), (hipStream_t)stream, feat,
                     prototypes, hard, conf, row_loss, dfeat_unit, K, Dp, T, 1.f / (float)rows);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_proto_accum(const float* feat_e, const int* hard, const unsigned char* conf, float* class_sum_cnt,
                                int B, int B_l, int K, int Dp, float repeat_ratio, void* stream) {
  STIL_REQUIRE(feat_e && hard && conf && class_sum_cnt && repeat_ratio > 0.f, "stil_proto_accum: bad arguments");
  hipLaunchKernelGGL(proto_accum_kernel, dim3(K), dim3(192), 0, (hipStream_t)stream, feat_e, hard, conf, class_sum_cnt,
                     B, B_l, Dp, 1.f / repeat_ratio);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_proto_add(const float* class_sum_cnt, float* prototypes_sum, float* prototypes_count_sum, int K,
                              int Dp, void* stream) {
  STIL_REQUIRE(class_sum_cnt && prototypes_sum && prototypes_count_sum, "stil_proto_add: null pointer");
  hipLaunchKernelGGL(proto_add_kernel, dim3(cdiv((long)K * (Dp + 1), 256)), dim3(256), 0, (hipStream_t)stream,
                     class_sum_cnt, prototypes_sum, prototypes_count_sum, K, Dp);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
// bad_count_dev[0] must be zeroed by the caller; >0 afterwards == the reference's assert (STiLModel.py:412) fails
extern "C" int stil_proto_commit(float* prototypes, float* prototypes_sum, float* prototypes_count_sum,
                                 int* bad_count_dev, int K, int Dp, void* stream) {
  STIL_REQUIRE(prototypes && prototypes_sum && prototypes_count_sum && bad_count_dev, "stil_proto_commit: null pointer");
  hipLaunchKernelGGL(proto_commit_kernel, dim3(K), dim3(128), 0, (hipStream_t)stream, prototypes, prototypes_sum,
                     prototypes_count_sum, bad_count_dev, K, Dp);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}


// ---- hard pseudo-labels (MMatch.py:223-226): idx = argmax_k p (first maximum), mask = max >= th, one-hot row
__global__ __launch_bounds__(64) void onehot_argmax_kernel(const float* __restrict__ p, int K, float th, float* __restrict__ onehot,
                                                            float* __restrict__ roww, int* __restrict__ idx) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* r = p + (long)row * K;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < K; j += 64) {
    const float v = r[j];
    if (v > best) { best = v; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  for (int j = lane; j < K; j += 64) onehot[(long)row * K + j] = (j == bi) ? 1.f : 0.f;
  if (lane == 0) {
    roww[row] = best >= th ? 1.f : 0.f;
    idx[row] = bi;
  }
}

extern "C" int stil_onehot_argmax(const float* probs, int rows, int K, float threshold, float* onehot, float* row_mask, int* idx,
                                  void* stream) {
  STIL_REQUIRE(probs && onehot && row_mask && idx && rows > 0 && K > 0, "stil_onehot_argmax: bad arguments");
  hipLaunchKernelGGL(onehot_argmax_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, probs, K, threshold, onehot, row_mask, idx);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}


// ---------------------------------------------------------------- CoMatch / SimMatch baselines
// CoMatch pseudo-label-graph contrastive loss (models/MatchModel/CoMatch.py:104-117) on S = f_s0 . [f_s1 ; queue_s]^T / T:
//   sim = exp(S);  m = (Q >= th);  w = Q m / sum_j(Q m);  p = sim m / sum_j sim;  row_loss = -sum_j log(p + 1e-7) m w
//   dS  = (softmax(S) * Csum - c) * inv_rows,   c = w p / (p + 1e-7),   Csum = sum_j c
// One block per unlabelled sample; the diagonal of Q is 1, so every row has at least one positive.
__global__ __launch_bounds__(256) void contrast_graph_kernel(const float* __restrict__ S, int lds, const float* __restrict__ Q, int ldq,
                                                              float th, float* __restrict__ row_loss, float* __restrict__ dS, int ldd,
                                                              int N, float inv_rows) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* sr = S + (long)r * lds;
  const float* qr = Q + (long)r * ldq;
  float D = 0.f, Qs = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    D += expf(sr[j]);
    const float q = qr[j];
    Qs += q >= th ? q : 0.f;
  }
  D = block_sum(D, red);
  Qs = block_sum(Qs, red);
  float a = 0.f, C = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float q = qr[j];
    if (q >= th) {
      const float p = expf(sr[j]) / D, w = q / Qs;
      a += logf(p + 1e-7f) * w;
      C += w * (p / (p + 1e-7f));
    }
  }
  a = block_sum(a, red);
  C = block_sum(C, red);
  if (threadIdx.x == 0) row_loss[r] = -a;
  if (dS)
    for (int j = threadIdx.x; j < N; j += 256) {
      const float q = qr[j], sm = expf(sr[j]) / D;
      const float c = q >= th ? (q / Qs) * (sm / (sm + 1e-7f)) : 0.f;
      dS[(long)r * ldd + j] = (sm * C - c) * inv_rows;
    }
}

extern "C" int stil_contrast_graph(const float* S, int lds, const float* Q, int ldq, float threshold, float* row_loss, float* dS,
                                   int ldd, int rows, int N, float inv_rows, void* stream) {
  STIL_REQUIRE(S && Q && row_loss && rows > 0 && N > 0 && lds >= N && ldq >= N && (!dS || ldd >= N), "stil_contrast_graph: bad arguments");
  hipLaunchKernelGGL(contrast_graph_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, lds, Q, ldq, threshold, row_loss, dS, ldd,
                     N, inv_rows);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// SimMatch label unfolding / aggregation (models/MatchModel/simmatch_model.py:289-302) for one unlabelled sample per block:
//   teacher[j] = tpo[j] * p[labels[j]] / sum_j(..)        (tpo = softmax(feat_ku . bank / tt), p = aligned class probabilities)
//   pseudo[k]  = p[k] * c + (sum_{j: labels[j] == k} tpo[j]) * (1 - c)    (bank entries added in index order, like scatter_add)
__global__ __launch_bounds__(256) void simmatch_unfold_kernel(const float* __restrict__ tpo, const float* __restrict__ p,
                                                               const long long* __restrict__ labels, float* __restrict__ teacher,
                                                               float* __restrict__ pseudo, int N, int K, float c_smooth) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* tr = tpo + (long)r * N;
  const float* pr = p + (long)r * K;
  float s = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) s += tr[j] * pr[labels[j]];
  s = block_sum(s, red);
  for (int j = threadIdx.x; j < N; j += 256) teacher[(long)r * N + j] = tr[j] * pr[labels[j]] / s;
  if (pseudo)
    for (int k = threadIdx.x; k < K; k += 256) {
      float acc = 0.f;
      if (c_smooth < 1.f)
        for (int j = 0; j < N; ++j) acc += labels[j] == k ? tr[j] : 0.f;
      pseudo[(long)r * K + k] = c_smooth < 1.f ? pr[k] * c_smooth + acc * (1.f - c_smooth) : pr[k];
    }
}

extern "C" int stil_simmatch_unfold(const float* tpo, const float* probs, const long long* labels, float* teacher, float* pseudo,
                                    int rows, int N, int K, float c_smooth, void* stream) {
  STIL_REQUIRE(tpo && probs && labels && teacher && rows > 0 && N > 0 && K > 0, "stil_simmatch_unfold: bad arguments");
  hipLaunchKernelGGL(simmatch_unfold_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, tpo, probs, labels, teacher, pseudo, N, K,
                     c_smooth);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// FreeMatch self-adaptive thresholding (models/MatchModel/FreeMatchFolder/freematch_model.py:132-168), one workgroup:
//   time_p <- m time_p + (1-m) mean_r max_k probs;  p_model <- m p_model + (1-m) mean_r probs;  label_hist <- m label_hist +
//   (1-m) bincount(argmax)/R;  mask[r] = max_k probs[r] >= time_p * p_model[idx_r] / max(p_model);  onehot[r] = e_idx_r.
// scratch: R floats.  Fixed-order sums.
__global__ __launch_bounds__(256) void freematch_update_kernel(const float* __restrict__ probs, int R, int K, float* __restrict__ p_model,
                                                                float* __restrict__ label_hist, float* __restrict__ time_p, float m,
                                                                float* __restrict__ mask, float* __restrict__ onehot, int* __restrict__ idx,
                                                                float* __restrict__ scratch) {
  __shared__ float red[16];
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float* pr = probs + (long)r * K;
    float best = pr[0];
    int bi = 0;
    for (int k = 1; k < K; ++k)
      if (pr[k] > best) { best = pr[k]; bi = k; }
    scratch[r] = best;
    idx[r] = bi;
    s += best;
    for (int k = 0; k < K; ++k) onehot[(long)r * K + k] = k == bi ? 1.f : 0.f;
  }
  s = block_sum(s, red);      // (block_sum ends with a barrier: scratch / idx are visible to the whole workgroup)
  const float tp = time_p[0] * m + (1.f - m) * (s / (float)R);
  float pmax = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) {
    float cs = 0.f, cnt = 0.f;
    for (int r = 0; r < R; ++r) { cs += probs[(long)r * K + k]; cnt += idx[r] == k ? 1.f : 0.f; }
    const float pm = p_model[k] * m + (1.f - m) * (cs / (float)R);
    p_model[k] = pm;
    label_hist[k] = label_hist[k] * m + (1.f - m) * (cnt / (float)R);
    pmax = fmaxf(pmax, pm);
  }
  pmax = block_max(pmax, red);
  __syncthreads();
  if (threadIdx.x == 0) time_p[0] = tp;
  for (int r = threadIdx.x; r < R; r += 256) mask[r] = scratch[r] >= tp * (p_model[idx[r]] / pmax) ? 1.f : 0.f;
}

extern "C" int stil_freematch_update(const float* probs, int rows, int K, float* p_model, float* label_hist, float* time_p,
                                     float momentum, float* mask, float* onehot, int* idx, float* scratch, void* stream) {
  STIL_REQUIRE(probs && p_model && label_hist && time_p && mask && onehot && idx && scratch && rows > 0 && K > 0,
               "stil_freematch_update: bad arguments");
  hipLaunchKernelGGL(freematch_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, probs, rows, K, p_model, label_hist, time_p,
                     momentum, mask, onehot, idx, scratch);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// FreeMatch fairness ("entropy") loss (freematch_utils.py:18-47) over the rows with mask != 0, one workgroup:
//   a = normalise(p_model * inf0(1/label_hist));  hist = bincount(argmax softmax(z))/n;  v = mean_r softmax(z) * inf0(1/hist);
//   loss = sum_k a_k log(v_k / sum(v) + 1e-12);  dz = d loss / d z (rows outside the mask: 0);  n == 0: loss = 0, dz = 0.
// P: [R,K] scratch (the softmax rows);  vec: 4*K floats of scratch.
__global__ __launch_bounds__(256) void freematch_entropy_kernel(const float* __restrict__ z, const float* __restrict__ mask, int R, int K,
                                                                 const float* __restrict__ p_model, const float* __restrict__ label_hist,
                                                                 float* __restrict__ loss, float* __restrict__ dz, float* __restrict__ P,
                                                                 int* __restrict__ pred, float* __restrict__ vec) {
  __shared__ float red[16];
  float* a = vec;            // modulated prob model
  float* sc = vec + K;       // inf0(1 / hist_s)
  float* mmv = vec + 2 * K;  // modulated mean prob
  float* dmean = vec + 3 * K;
  float n = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const bool on = mask[r] != 0.f;
    n += on ? 1.f : 0.f;
    const float* zr = z + (long)r * K;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, zr[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(zr[k] - mx);
    float best = -1.f;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      const float p = expf(zr[k] - mx) / se;
      P[(long)r * K + k] = p;
      if (p > best) { best = p; bi = k; }
    }
    pred[r] = on ? bi : -1;
  }
  n = block_sum(n, red);
  if (n == 0.f) {
    if (threadIdx.x == 0) loss[0] = 0.f;
    for (long i = threadIdx.x; i < (long)R * K; i += 256) dz[i] = 0.f;
    return;
  }
  float asum = 0.f, vsum = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    float cnt = 0.f, cs = 0.f;
    for (int r = 0; r < R; ++r)
      if (pred[r] >= 0) { cnt += pred[r] == k ? 1.f : 0.f; cs += P[(long)r * K + k]; }
    const float h = cnt / n, lh = label_hist[k];
    const float s1 = 1.f / lh, s2 = 1.f / h;
    a[k] = p_model[k] * (isinf(s1) ? 0.f : s1);
    sc[k] = isinf(s2) ? 0.f : s2;
    mmv[k] = (cs / n) * sc[k];
    asum += a[k];
    vsum += mmv[k];
  }
  asum = block_sum(asum, red);
  vsum = block_sum(vsum, red);
  float l = 0.f, bs = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float ak = a[k] / asum, mk = mmv[k] / vsum;
    const float b = ak / (mk + 1e-12f);
    l += ak * logf(mk + 1e-12f);
    bs += b * mk;
    a[k] = b;
  }
  l = block_sum(l, red);
  bs = block_sum(bs, red);
  if (threadIdx.x == 0) loss[0] = l;
  for (int k = threadIdx.x; k < K; k += 256) dmean[k] = sc[k] * (a[k] - bs) / vsum;
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float* dr = dz + (long)r * K;
    if (pred[r] < 0) {
      for (int k = 0; k < K; ++k) dr[k] = 0.f;
      continue;
    }
    const float* pr = P + (long)r * K;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += pr[k] * dmean[k];
    for (int k = 0; k < K; ++k) dr[k] = pr[k] * (dmean[k] - dot) / n;
  }
}

extern "C" int stil_freematch_entropy(const float* logits, const float* mask, int rows, int K, const float* p_model,
                                      const float* label_hist, float* loss, float* dlogits, float* P, int* pred, float* vec,
                                      void* stream) {
  STIL_REQUIRE(logits && mask && p_model && label_hist && loss && dlogits && P && pred && vec && rows > 0 && K > 0,
               "stil_freematch_entropy: bad arguments");
  hipLaunchKernelGGL(freematch_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, mask, rows, K, p_model, label_hist,
                     loss, dlogits, P, pred, vec);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// ---- logged partition ratios (STiLModel.py:307-311): out = {mean(mask1), mean(case == 1), .. == 2, .. == 3, .. == 4}
// flags: [rows, ld] bytes, columns (CGPL case id 1..4, mask1, ...) as written by cgpl_pgls_kernel.  One block, fixed order.
__global__ __launch_bounds__(256) void flag_ratios_kernel(const unsigned char* __restrict__ flags, int ld, int rows, float* __restrict__ out) {
  __shared__ int sh[5][4];
  int c[5] = {0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < rows; r += 256) {
    const int cs = flags[(long)r * ld], m1 = flags[(long)r * ld + 1];
    c[0] += m1 != 0;
    c[1] += cs == 1; c[2] += cs == 2; c[3] += cs == 3; c[4] += cs == 4;
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    int v = c[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) out[threadIdx.x] = (float)(sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]) / (float)rows;
}

extern "C" int stil_flag_ratios(const unsigned char* flags, int ld, int rows, float* out5, void* stream) {
  STIL_REQUIRE(flags && out5 && rows > 0 && ld >= 2, "stil_flag_ratios: bad arguments");
  hipLaunchKernelGGL(flag_ratios_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, flags, ld, rows, out5);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
