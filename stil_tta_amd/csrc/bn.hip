// NHWC BatchNorm (train statistics / apply / backward), eval-mode affine folding, 3x3-s2 max-pool
// and the stem im2col.  HBM-bound kernels: 16-byte coalesced accesses along the channel axis,
// per-thread register accumulation, LDS cross-row reduction, fixed-order (deterministic) finals.
// Replaces nn.BatchNorm2d / nn.ReLU / residual add / nn.MaxPool2d of models/resnets.py:112-132,248-252.
#include "common.h"

static inline int pick_ctile(int C) {
  if (C % 256 == 0) return 256;
  if (C % 128 == 0) return 128;
  if (C % 64 == 0) return 64;
  return 0;
}

// ---- statistics: shifted one-pass sums  S1 = sum(x - K), S2 = sum((x - K)^2).
// The shift K[c] is a PILOT MEAN over <= 256 rows spread across the tensor (bn_pilot_kernel): |K - mean| << std, so
// var = S2/M - (S1/M)^2 has no cancellation (a single sample such as x[0,c] -- a zero-padded corner pixel -- is up
// to several sigma off and cost ~3x the forward error of ATen's two-pass variance).
__global__ __launch_bounds__(256) void bn_pilot_kernel(const float* __restrict__ x, float* __restrict__ K, int M, int C) {
  __shared__ float sh[8 * 32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;  // 32 channels x 8 row lanes
  const int c = blockIdx.x * 32 + cl;
  const int n = M < 256 ? M : 256;
  const long stride = M / n;
  float s = 0.f;
  if (c < C)
    for (int j = lane; j < n; j += 8) s += x[(long)j * stride * C + c];
  sh[lane * 32 + cl] = s;
  __syncthreads();
  if (lane == 0 && c < C) {
    float t = 0.f;
    for (int l = 0; l < 8; ++l) t += sh[l * 32 + cl];
    K[c] = t / (float)n;
  }
}

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, const float* __restrict__ Kp,
                                                                float* __restrict__ part,
                                                                int M, int C, int ctile, int rows_per_chunk) {
  __shared__ float sh[2 * 1024];
  const int tpr = ctile >> 2, rpb = 256 / tpr;
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int c = blockIdx.x * ctile + tc * 4;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  const float4 K = *reinterpret_cast<const float4*>(Kp + c);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int m = m0 + rl; m < m1; m += rpb) {
    float4 v = *reinterpret_cast<const float4*>(x + (long)m * C + c);
    float d;
    d = v.x - K.x; s1.x += d; s2.x += d * d;
    d = v.y - K.y; s1.y += d; s2.y += d * d;
    d = v.z - K.z; s1.z += d; s2.z += d * d;
    d = v.w - K.w; s1.w += d; s2.w += d * d;
  }
  float* a = sh + rl * ctile + tc * 4;
  float* b = sh + 1024 + rl * ctile + tc * 4;
  a[0] = s1.x; a[1] = s1.y; a[2] = s1.z; a[3] = s1.w;
  b[0] = s2.x; b[1] = s2.y; b[2] = s2.z; b[3] = s2.w;
  __syncthreads();
  if (threadIdx.x < ctile) {
    float t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rpb; ++r) { t1 += sh[r * ctile + threadIdx.x]; t2 += sh[1024 + r * ctile + threadIdx.x]; }
    const int cc = blockIdx.x * ctile + threadIdx.x;
    const int nch = gridDim.y;
    part[(long)blockIdx.y * C + cc] = t1;
    part[(long)(nch + blockIdx.y) * C + cc] = t2;
  }
}

// stats[0]=mean, [1]=rstd, [2]=a=gamma*rstd, [3]=beta ; updates running stats (momentum, unbiased var)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ Kp, const float* __restrict__ part, int nch, int M, int C,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float* __restrict__ rmean, float* __restrict__ rvar, long long* nbt,
                                      float* __restrict__ stats, float eps, float momentum) {
  __shared__ float sh[2 * 8 * 32];
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  int c;
  float tot[2];
  if (!chunk_reduce<8, 2>(part, nch, C, sh, c, tot)) return;
  const float s1 = tot[0], s2 = tot[1];
  const float invM = 1.f / (float)M;
  const float d = s1 * invM;
  const float mean = Kp[c] + d;
  float var = s2 * invM - d * d;
  var = fmaxf(var, 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  const float a = gamma[c] * rstd;
  stats[c] = mean; stats[C + c] = rstd; stats[2 * C + c] = a; stats[3 * C + c] = beta[c];
  if (rmean) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    float unb = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

// Same outputs from the per-tile (mean_t, M2_t) partials the conv GEMM's epilogue wrote (stil_gemm_nt colstats).
// With n_t rows in tile t:  mean = S1/M,  M2 = S3 + (S2 - S1^2/M),  S1 = sum n_t mean_t, S2 = sum n_t mean_t^2, S3 = sum M2_t.
// The three sums run in DOUBLE (every term is an exact product of floats, so the S2 - S1^2/M cancellation costs nothing),
// split over `nsplit` blocks per 32 columns (stage 1) and added in a fixed order (stage 2): deterministic.
// 256-thread workgroups (8 row lanes x 32 columns): these short dependent kernels run while the other stream's GEMM keeps
// every CU full of 256-thread workgroups -- a 1024-thread workgroup had to wait for four of them to retire on ONE CU
// (55 us instead of 8 us per launch under the two-stream step).
__global__ __launch_bounds__(256) void bn_tiles_stage1_kernel(const float* __restrict__ ts, int nt, int tile_rows, int M, int C,
                                                                int tiles_per_split, double* __restrict__ part) {
  __shared__ double sh[3 * 8 * 32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int t0 = blockIdx.y * tiles_per_split, t1 = min(nt, t0 + tiles_per_split);
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < C)
    for (int t = t0 + lane; t < t1; t += 8) {
      const double n = (double)min(tile_rows, M - t * tile_rows);
      const double m = (double)ts[((long)t * 2) * C + c];
      s1 += n * m; s2 += n * m * m; s3 += (double)ts[((long)t * 2 + 1) * C + c];
    }
  sh[lane * 32 + cl] = s1; sh[256 + lane * 32 + cl] = s2; sh[512 + lane * 32 + cl] = s3;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  double a = 0.0, b = 0.0, d = 0.0;
  for (int l = 0; l < 8; ++l) { a += sh[l * 32 + cl]; b += sh[256 + l * 32 + cl]; d += sh[512 + l * 32 + cl]; }
  const int nsplit = gridDim.y;
  part[((long)0 * nsplit + blockIdx.y) * C + c] = a;
  part[((long)1 * nsplit + blockIdx.y) * C + c] = b;
  part[((long)2 * nsplit + blockIdx.y) * C + c] = d;
}

__global__ __launch_bounds__(256) void bn_tiles_stage2_kernel(const double* __restrict__ part, int nsplit, int M, int C,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float* __restrict__ rmean, float* __restrict__ rvar, long long* nbt,
                                      float* __restrict__ stats, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    s1 += part[((long)0 * nsplit + k) * C + c]; s2 += part[((long)1 * nsplit + k) * C + c]; s3 += part[((long)2 * nsplit + k) * C + c];
  }
  const double mean_d = s1 / (double)M;
  double m2 = s3 + (s2 - s1 * mean_d);
  const float mean = (float)mean_d;
  const float var = fmaxf((float)(m2 / (double)M), 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  stats[c] = mean; stats[C + c] = rstd; stats[2 * C + c] = gamma[c] * rstd; stats[3 * C + c] = beta[c];
  if (rmean) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    const float unb = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

// Both stages in ONE launch when a single split covers every tile (nt <= 256: small per-GPU batches, where the step is bound by
// its ~1300 dependent launches and not by bytes).  Same sums in the same order as stage 1 + stage 2 with nsplit = 1: identical bits.
__global__ __launch_bounds__(256) void bn_tiles_fused_kernel(const float* __restrict__ ts, int nt, int tile_rows, int M, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ rmean, float* __restrict__ rvar, long long* nbt,
                                                              float* __restrict__ stats, float eps, float momentum) {
  __shared__ double sh[3 * 8 * 32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < C)
    for (int t = lane; t < nt; t += 8) {
      const double n = (double)min(tile_rows, M - t * tile_rows);
      const double m = (double)ts[((long)t * 2) * C + c];
      s1 += n * m; s2 += n * m * m; s3 += (double)ts[((long)t * 2 + 1) * C + c];
    }
  sh[lane * 32 + cl] = s1; sh[256 + lane * 32 + cl] = s2; sh[512 + lane * 32 + cl] = s3;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  double a = 0.0, b = 0.0, d = 0.0;
  for (int l = 0; l < 8; ++l) { a += sh[l * 32 + cl]; b += sh[256 + l * 32 + cl]; d += sh[512 + l * 32 + cl]; }
  double t1 = 0.0, t2 = 0.0, t3 = 0.0;   // stage 2's loop over its one split
  t1 += a; t2 += b; t3 += d;
  const double mean_d = t1 / (double)M;
  double m2 = t3 + (t2 - t1 * mean_d);
  const float mean = (float)mean_d;
  const float var = fmaxf((float)(m2 / (double)M), 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  stats[c] = mean; stats[C + c] = rstd; stats[2 * C + c] = gamma[c] * rstd; stats[3 * C + c] = beta[c];
  if (rmean) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    const float unb = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

// eval mode: ab[0] = a = gamma / sqrt(running_var + eps), ab[1] = beta, ab[2] = running_mean.  The conv epilogue
// applies (y - mean) * a + beta: the subtractive form keeps ATen's accuracy (no x*a - mean*a cancellation).
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                      float* __restrict__ ab, int C, float eps) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = gamma[c] / sqrtf(rvar[c] + eps);
  ab[c] = a; ab[C + c] = beta[c]; ab[2 * C + c] = rmean[c];
}

// z = relu?( (x - mean)*a + beta (+ resid) );  rstats (optional): resid is the RAW output of the shortcut's conv and rstats that
// layer's statistics block -- its BatchNorm (no ReLU) is applied here, (resid - rmean)*ra + rbeta, exactly as its own apply pass
// would have (models/resnets.py:126-129: identity = self.downsample(x); out += identity)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                        const float* __restrict__ resid, const float* __restrict__ rstats,
                                                        float* __restrict__ z, long total4, int C, int relu) {
  const float* A = stats + 2 * C;
  const float* B = stats + 3 * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 a = *reinterpret_cast<const float4*>(A + c);
    const float4 b = *reinterpret_cast<const float4*>(B + c);
    const float4 mu = *reinterpret_cast<const float4*>(stats + c);
    v.x = (v.x - mu.x) * a.x + b.x; v.y = (v.y - mu.y) * a.y + b.y; v.z = (v.z - mu.z) * a.z + b.z; v.w = (v.w - mu.w) * a.w + b.w;
    if (resid) {
      float4 r = reinterpret_cast<const float4*>(resid)[i];
      if (rstats) {
        const float4 rm = *reinterpret_cast<const float4*>(rstats + c);
        const float4 ra = *reinterpret_cast<const float4*>(rstats + 2 * C + c);
        const float4 rb = *reinterpret_cast<const float4*>(rstats + 3 * C + c);
        r.x = (r.x - rm.x) * ra.x + rb.x; r.y = (r.y - rm.y) * ra.y + rb.y; r.z = (r.z - rm.z) * ra.z + rb.z; r.w = (r.w - rm.w) * ra.w + rb.w;
      }
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4*>(z)[i] = v;
  }
}

// ---- backward pass 1: per-channel sums of g and g*xhat, g = dz * (z > 0) if relu.  Optionally stores g.
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                              const float* __restrict__ x, const float* __restrict__ stats,
                                                              float* __restrict__ gout, float* __restrict__ part, int M,
                                                              int C, int ctile, int rows_per_chunk, int relu) {
  __shared__ float sh[2 * 1024];
  const int tpr = ctile >> 2, rpb = 256 / tpr;
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int c = blockIdx.x * ctile + tc * 4;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  const float4 mu = *reinterpret_cast<const float4*>(stats + c);
  const float4 rs = *reinterpret_cast<const float4*>(stats + C + c);
  const float4 ap = *reinterpret_cast<const float4*>(stats + 2 * C + c);
  const float4 bp = *reinterpret_cast<const float4*>(stats + 3 * C + c);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int m = m0 + rl; m < m1; m += rpb) {
    const long o = (long)m * C + c;
    float4 g = *reinterpret_cast<const float4*>(dz + o);
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    if (relu == 1) {
      const float4 zz = *reinterpret_cast<const float4*>(z + o);
      g.x = zz.x > 0.f ? g.x : 0.f; g.y = zz.y > 0.f ? g.y : 0.f;
      g.z = zz.z > 0.f ? g.z : 0.f; g.w = zz.w > 0.f ? g.w : 0.f;
    } else if (relu == 2) {  // no residual: z = relu((x - mean)*a + beta), the sign is recomputed exactly as bn_apply did
      g.x = ((v.x - mu.x) * ap.x + bp.x) > 0.f ? g.x : 0.f; g.y = ((v.y - mu.y) * ap.y + bp.y) > 0.f ? g.y : 0.f;
      g.z = ((v.z - mu.z) * ap.z + bp.z) > 0.f ? g.z : 0.f; g.w = ((v.w - mu.w) * ap.w + bp.w) > 0.f ? g.w : 0.f;
    }
    if (gout) *reinterpret_cast<float4*>(gout + o) = g;
    s1.x += g.x; s2.x += g.x * ((v.x - mu.x) * rs.x);
    s1.y += g.y; s2.y += g.y * ((v.y - mu.y) * rs.y);
    s1.z += g.z; s2.z += g.z * ((v.z - mu.z) * rs.z);
    s1.w += g.w; s2.w += g.w * ((v.w - mu.w) * rs.w);
  }
  float* a = sh + rl * ctile + tc * 4;
  float* b = sh + 1024 + rl * ctile + tc * 4;
  a[0] = s1.x; a[1] = s1.y; a[2] = s1.z; a[3] = s1.w;
  b[0] = s2.x; b[1] = s2.y; b[2] = s2.z; b[3] = s2.w;
  __syncthreads();
  if (threadIdx.x < ctile) {
    float t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rpb; ++r) { t1 += sh[r * ctile + threadIdx.x]; t2 += sh[1024 + r * ctile + threadIdx.x]; }
    const int cc = blockIdx.x * ctile + threadIdx.x;
    const int nch = gridDim.y;
    part[(long)blockIdx.y * C + cc] = t1;
    part[(long)(nch + blockIdx.y) * C + cc] = t2;
  }
}

// dgamma/dbeta (+)= ; coef[0]=gamma*rstd, [1]=dbeta/M, [2]=dgamma/M
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int nch, int M, int C,
                                    const float* __restrict__ gamma, const float* __restrict__ stats,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                    int accumulate) {
  __shared__ float sh[2 * 8 * 32];
  int c;
  float tot[2];
  if (!chunk_reduce<8, 2>(part, nch, C, sh, c, tot)) return;
  const float s1 = tot[0], s2 = tot[1];
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
  const float invM = 1.f / (float)M;
  coef[c] = gamma[c] * stats[C + c];
  coef[C + c] = s1 * invM;
  coef[2 * C + c] = s2 * invM;
}

// The same over 32 lanes x 8 column quads per workgroup (C % 4 == 0): one dwordx4 per lane and chunk, a dependent chain of
// nch / 32 loads instead of nch / 8 (the kernel is that chain: 25 us alone, 110-160 us beside the other stream's GEMMs, on
// the main stream of every BatchNorm backward).  Lane sums are added in lane order: deterministic.
__global__ __launch_bounds__(256) void bn_bwd_final4_kernel(const float* __restrict__ part, int nch, int M, int C,
                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                     int accumulate) {
  __shared__ float4 sh[2 * 32 * 8];
  const int cq = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int c = (blockIdx.x * 8 + cq) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (c < C)
    for (int i = lane; i < nch; i += 32) {
      const float4 u = *reinterpret_cast<const float4*>(part + (long)i * C + c);
      const float4 v = *reinterpret_cast<const float4*>(part + (long)(nch + i) * C + c);
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
  sh[lane * 8 + cq] = a; sh[256 + lane * 8 + cq] = b;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  for (int l = 1; l < 32; ++l) {
    const float4 u = sh[l * 8 + cq], v = sh[256 + l * 8 + cq];
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
  }
  const float s1[4] = {a.x, a.y, a.z, a.w}, s2[4] = {b.x, b.y, b.z, b.w};
  const float invM = 1.f / (float)M;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cc = c + j;
    if (dbeta) dbeta[cc] = accumulate ? dbeta[cc] + s1[j] : s1[j];
    if (dgamma) dgamma[cc] = accumulate ? dgamma[cc] + s2[j] : s2[j];
    coef[cc] = gamma[cc] * stats[C + cc];
    coef[C + cc] = s1[j] * invM;
    coef[2 * C + cc] = s2[j] * invM;
  }
}

// ---- backward statistics from per-tile partials: the input-gradient GEMM that PRODUCED dz left, per 64-row tile t and column,
// ts[(2t)*C + c] = sum g', ts[(2t+1)*C + c] = sum g' * xhat (stil_gemm_nt `bstats`), so no pass over dz and x is needed.  Two short
// kernels combine them in DOUBLE in a fixed order (stage 1: nsplit blocks per 32 columns, 8 tile lanes each; final: the
// nsplit sums), then write what bn_bwd_final_kernel writes.
__global__ __launch_bounds__(256) void bn_bwd_tiles_stage1_kernel(const float* __restrict__ ts, int nt, int C, int tiles_per_split,
                                                                   double* __restrict__ part) {
  __shared__ double sh[2 * 8 * 32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int t0 = blockIdx.y * tiles_per_split, t1 = min(nt, t0 + tiles_per_split);
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int t = t0 + lane; t < t1; t += 8) {
      s1 += (double)ts[((long)t * 2) * C + c];
      s2 += (double)ts[((long)t * 2 + 1) * C + c];
    }
  sh[lane * 32 + cl] = s1; sh[256 + lane * 32 + cl] = s2;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  double a = 0.0, b = 0.0;
  for (int l = 0; l < 8; ++l) { a += sh[l * 32 + cl]; b += sh[256 + l * 32 + cl]; }
  const int nsplit = gridDim.y;
  part[((long)0 * nsplit + blockIdx.y) * C + c] = a;
  part[((long)1 * nsplit + blockIdx.y) * C + c] = b;
}
__global__ __launch_bounds__(256) void bn_bwd_tiles_final_kernel(const double* __restrict__ part, int nsplit, int M, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ coef, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < nsplit; ++k) { a += part[((long)0 * nsplit + k) * C + c]; b += part[((long)1 * nsplit + k) * C + c]; }
  const float s1 = (float)a, s2 = (float)b;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
  const float invM = 1.f / (float)M;
  coef[c] = gamma[c] * stats[C + c];
  coef[C + c] = s1 * invM;
  coef[2 * C + c] = s2 * invM;
}

// ... and in one launch when a single split covers every tile (see bn_tiles_fused_kernel): identical bits.
__global__ __launch_bounds__(256) void bn_bwd_tiles_fused_kernel(const float* __restrict__ ts, int nt, int M, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ coef, int accumulate) {
  __shared__ double sh[2 * 8 * 32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int t = lane; t < nt; t += 8) {
      s1 += (double)ts[((long)t * 2) * C + c];
      s2 += (double)ts[((long)t * 2 + 1) * C + c];
    }
  sh[lane * 32 + cl] = s1; sh[256 + lane * 32 + cl] = s2;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  double a = 0.0, b = 0.0;
  for (int l = 0; l < 8; ++l) { a += sh[l * 32 + cl]; b += sh[256 + l * 32 + cl]; }
  double ta = 0.0, tb = 0.0;
  ta += a; tb += b;
  const float f1 = (float)ta, f2 = (float)tb;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + f1 : f1;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + f2 : f2;
  const float invM = 1.f / (float)M;
  coef[c] = gamma[c] * stats[C + c];
  coef[C + c] = f1 * invM;
  coef[2 * C + c] = f2 * invM;
}

// dx = gamma*rstd * (g - dbeta/M - xhat*dgamma/M)
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                         const float* __restrict__ x, const float* __restrict__ stats,
                                                         const float* __restrict__ coef, float* __restrict__ dx,
                                                         long total4, int C, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    float4 g = reinterpret_cast<const float4*>(dz)[i];
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = *reinterpret_cast<const float4*>(stats + c);
    if (relu == 1) {
      const float4 zz = reinterpret_cast<const float4*>(z)[i];
      g.x = zz.x > 0.f ? g.x : 0.f; g.y = zz.y > 0.f ? g.y : 0.f;
      g.z = zz.z > 0.f ? g.z : 0.f; g.w = zz.w > 0.f ? g.w : 0.f;
    } else if (relu == 2) {
      const float4 ap = *reinterpret_cast<const float4*>(stats + 2 * C + c);
      const float4 bp = *reinterpret_cast<const float4*>(stats + 3 * C + c);
      g.x = ((v.x - mu.x) * ap.x + bp.x) > 0.f ? g.x : 0.f; g.y = ((v.y - mu.y) * ap.y + bp.y) > 0.f ? g.y : 0.f;
      g.z = ((v.z - mu.z) * ap.z + bp.z) > 0.f ? g.z : 0.f; g.w = ((v.w - mu.w) * ap.w + bp.w) > 0.f ? g.w : 0.f;
    }
    const float4 rs = *reinterpret_cast<const float4*>(stats + C + c);
    const float4 k1 = *reinterpret_cast<const float4*>(coef + c);
    const float4 k2 = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 k3 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    float4 o;
    o.x = k1.x * (g.x - k2.x - (v.x - mu.x) * rs.x * k3.x);
    o.y = k1.y * (g.y - k2.y - (v.y - mu.y) * rs.y * k3.y);
    o.z = k1.z * (g.z - k2.z - (v.z - mu.z) * rs.z * k3.z);
    o.w = k1.w * (g.w - k2.w - (v.w - mu.w) * rs.w * k3.w);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
}

// ---- max-pool 3x3 s2 p1 (NHWC).  idx = ky*3+kx of the FIRST maximum in scan order (ATen tie rule).
// four channels per thread (C % 4 == 0): 16-byte loads / stores, one (n, oy, ox) decode per 4 elements
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                           int N, int H, int W, int C, int OH, int OW) {
  const int C4 = C >> 2;
  const long total = (long)N * OH * OW * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + ((long)(n * H + iy) * W + ix) * C + c);
        const int tap = ky * 3 + kx;
        if (v.x > best.x || v.x != v.x) { best.x = v.x; b0 = tap; }
        if (v.y > best.y || v.y != v.y) { best.y = v.y; b1 = tap; }
        if (v.z > best.z || v.z != v.z) { best.z = v.z; b2 = tap; }
        if (v.w > best.w || v.w != v.w) { best.w = v.w; b3 = tap; }
      }
    }
    const long o = ((long)(n * OH + oy) * OW + ox) * C + c;
    *reinterpret_cast<float4*>(y + o) = best;
    *reinterpret_cast<uchar4*>(idx + o) = make_uchar4((unsigned char)b0, (unsigned char)b1, (unsigned char)b2, (unsigned char)b3);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                           float* __restrict__ dx, int N, int H, int W, int C, int OH, int OW) {
  const int C4 = C >> 2;
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int ix = (int)(t % W); t /= W;
    const int iy = (int)(t % H);
    const int n = (int)(t / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= OH) continue;
      const int ky = iy - (oy * 2 - 1);
      if (ky < 0 || ky > 2) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= OW) continue;
        const int kx = ix - (ox * 2 - 1);
        if (kx < 0 || kx > 2) continue;
        const long o = ((long)(n * OH + oy) * OW + ox) * C + c;
        const uchar4 k = *reinterpret_cast<const uchar4*>(idx + o);
        const float4 g = *reinterpret_cast<const float4*>(dy + o);
        const int tap = ky * 3 + kx;
        if (k.x == tap) s.x += g.x;
        if (k.y == tap) s.y += g.y;
        if (k.z == tap) s.z += g.z;
        if (k.w == tap) s.w += g.w;
      }
    }
    *reinterpret_cast<float4*>(dx + ((long)(n * H + iy) * W + ix) * C + c) = s;
  }
}

// ---- stem im2col: NCHW image -> col[M, Kp], k = c*KH*KW + ky*KW + kx (the reference weight order), zero padded
// col[m][k] (k = (c, ky, kx), zero-padded to Kp, Kp % 4 == 0) from an NCHW image.  One thread owns one 16-byte quad
// of a row and walks down the rows: the k -> (c, ky, kx) decode comes from a small LDS table, the m -> (n, oy, ox)
// decode is two divisions per quad, stores are full float4 lines (HBM-write bound: 2 GB at B=256 / 224 px).
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, float* __restrict__ col, int N, int Cin,
                                                           int H, int W, int OH, int OW, int KH, int KW, int stride, int pad,
                                                           int Kp, int rows_per_block) {
  extern __shared__ int ktab[];  // Kp entries: (c * KH + ky) << 8 | kx, or -1 for the padding columns
  const int K = Cin * KH * KW, Q = Kp >> 2;
  for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
    int v = -1;
    if (k < K) {
      const int c = k / (KH * KW), r = k - c * KH * KW, ky = r / KW, kx = r - ky * KW;
      v = ((c * KH + ky) << 8) | kx;
    }
    ktab[k] = v;
  }
  __syncthreads();
  const long M = (long)N * OH * OW;
  const long m0 = (long)blockIdx.x * rows_per_block;
  const int quads = rows_per_block * Q;
  for (int i = threadIdx.x; i < quads; i += blockDim.x) {
    const int rr = i / Q, q = i - rr * Q;
    const long m = m0 + rr;
    if (m >= M) break;
    const int ox = (int)(m % OW);
    const long t = m / OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = ktab[q * 4 + j];
      v[j] = 0.f;
      if (e >= 0) {
        const int cky = e >> 8, kx = e & 255, c = cky / KH, ky = cky - c * KH;
        const int iy = iy0 + ky, ix = ix0 + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = x[((long)(n * Cin + c) * H + iy) * W + ix];
      }
    }
    *reinterpret_cast<float4*>(col + m * Kp + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---------------------------------------------------------------------------------------
static inline int bn_chunks(int M, int C, int ctile) {
  int colblocks = C / ctile;
  int want = cdiv(1024, colblocks);
  int maxc = cdiv(M, 64);
  int n = want < maxc ? want : maxc;
  return n < 1 ? 1 : n;
}

extern "C" size_t stil_bn_workspace_bytes(int M, int C) {
  int ct = pick_ctile(C);
  if (!ct) return 0;
  return ((size_t)2 * bn_chunks(M, C, ct) + 1) * C * sizeof(float);
}

extern "C" int stil_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, long long* num_batches_tracked, const float* resid, float* z,
                                 float* stats, int M, int C, int relu, float eps, float momentum, float* workspace,
                                 size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(x && gamma && beta && z && stats && workspace, "stil_bn_train_fwd: null pointer");
  int ct = pick_ctile(C);
  STIL_REQUIRE(ct != 0, "stil_bn_train_fwd: C=%d must be a multiple of 64", C);
  int nch = bn_chunks(M, C, ct);
  STIL_REQUIRE(workspace_bytes >= ((size_t)2 * nch + 1) * C * sizeof(float), "stil_bn_train_fwd: workspace too small");
  float* pilot = workspace + (size_t)2 * nch * C;
  int rpc = cdiv(M, nch);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_pilot_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, x, pilot, M, C);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C / ct, nch), dim3(256), 0, s, x, pilot, workspace, M, C, ct, rpc);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, pilot, workspace, nch, M, C, gamma, beta,
                     running_mean, running_var, num_batches_tracked, stats, eps, momentum);
  STIL_LAUNCH_CHECK();
  long total4 = (long)M * C / 4;
  int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, s, x, stats, resid, (const float*)nullptr, z, total4, C, relu);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float* ab, int C, float eps, void* stream) {
  STIL_REQUIRE(gamma && beta && running_mean && running_var && ab, "stil_bn_eval_affine: null pointer");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, ab, C, eps);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// Training forward when the producing GEMM already wrote per-tile statistics (tilestats: [2*cdiv(M,tile_rows), C]).
static inline int bn_tiles_nsplit(int nt) { return cdiv(nt, 256); }
extern "C" size_t stil_bn_tiles_workspace_bytes(int M, int C, int tile_rows) {
  if (M <= 0 || C <= 0 || tile_rows <= 0) return 0;
  return (size_t)3 * bn_tiles_nsplit(cdiv(M, tile_rows)) * C * sizeof(double);
}
extern "C" int stil_bn_train_fwd_tiles(const float* x, const float* tilestats, int tile_rows, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, long long* num_batches_tracked,
                                       const float* resid, const float* resid_stats, float* z, float* stats, int M, int C, int relu,
                                       float eps, float momentum, void* workspace, size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(x && tilestats && gamma && beta && stats && workspace, "stil_bn_train_fwd_tiles: null pointer");
  STIL_REQUIRE(z || !resid, "stil_bn_train_fwd_tiles: statistics-only call (z == NULL) cannot take a residual");
  STIL_REQUIRE(tile_rows > 0 && M > 0 && C % 4 == 0, "stil_bn_train_fwd_tiles: bad shape M=%d C=%d tile_rows=%d", M, C, tile_rows);
  STIL_REQUIRE(workspace_bytes >= stil_bn_tiles_workspace_bytes(M, C, tile_rows) && ((uintptr_t)workspace % 8) == 0,
               "stil_bn_train_fwd_tiles: workspace too small or not 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int nt = cdiv(M, tile_rows), nsplit = bn_tiles_nsplit(nt);
  if (nsplit == 1) {   // one launch: every tile fits one split
    hipLaunchKernelGGL(bn_tiles_fused_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, tilestats, nt, tile_rows, M, C, gamma, beta,
                       running_mean, running_var, num_batches_tracked, stats, eps, momentum);
    STIL_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(bn_tiles_stage1_kernel, dim3(cdiv(C, 32), nsplit), dim3(256), 0, s, tilestats, nt, tile_rows, M, C,
                       cdiv(nt, nsplit), (double*)workspace);
    STIL_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_tiles_stage2_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const double*)workspace, nsplit, M, C, gamma, beta,
                       running_mean, running_var, num_batches_tracked, stats, eps, momentum);
    STIL_LAUNCH_CHECK();
  }
  if (!z) return STIL_OK;   // statistics only: the consumer applies them while it stages its operand
  long total4 = (long)M * C / 4;
  int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  STIL_REQUIRE(!resid_stats || resid, "stil_bn_train_fwd_tiles: resid_stats without resid");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, s, x, stats, resid, resid_stats, z, total4, C, relu);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// dz: grad wrt the block output; z: that output; x: conv output (pre-BN).  relu: 0 = none, 1 = mask from z (z > 0),
// 2 = mask recomputed from x and the saved statistics (only valid without a residual input: saves reading z twice).
// gout (optional): receives g = dz*mask (the gradient of the residual branch).  coef: [3*C] scratch.
extern "C" int stil_bn_train_bwd(const float* dz, const float* z, const float* x, const float* gamma,
                                 const float* stats, float* dx, float* gout, float* dgamma, float* dbeta, float* coef,
                                 int M, int C, int relu, int accumulate, float* workspace, size_t workspace_bytes,
                                 void* stream) {
  STIL_REQUIRE(dz && x && gamma && stats && dx && coef && workspace, "stil_bn_train_bwd: null pointer");
  STIL_REQUIRE(relu >= 0 && relu <= 2 && (relu != 1 || z), "stil_bn_train_bwd: relu must be 0, 1 (needs z) or 2");
  int ct = pick_ctile(C);
  STIL_REQUIRE(ct != 0, "stil_bn_train_bwd: C=%d must be a multiple of 64", C);
  int nch = bn_chunks(M, C, ct);
  STIL_REQUIRE(workspace_bytes >= (size_t)2 * nch * C * sizeof(float), "stil_bn_train_bwd: workspace too small");
  int rpc = cdiv(M, nch);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C / ct, nch), dim3(256), 0, s, dz, z, x, stats, gout, workspace, M, C,
                     ct, rpc, relu);
  STIL_LAUNCH_CHECK();
  if (C % 4 == 0 && ((uintptr_t)workspace % 16) == 0)
    hipLaunchKernelGGL(bn_bwd_final4_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, workspace, nch, M, C, gamma, stats, dgamma,
                       dbeta, coef, accumulate);
  else
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, workspace, nch, M, C, gamma, stats, dgamma,
                       dbeta, coef, accumulate);
  STIL_LAUNCH_CHECK();
  long total4 = (long)M * C / 4;
  int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  // if g was materialised, read it (already masked) instead of dz,z
  hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(grid), dim3(256), 0, s, gout ? gout : dz, z, x, stats, coef, dx, total4, C,
                     gout ? 0 : relu);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// BatchNorm backward when the GEMM that produced dz already left the per-tile sums (stil_gemm_nt `bstats`): statistics from
// `tilestats` [2*ntiles][C] (two short kernels), then the dx pass of stil_bn_train_bwd.  relu as there (the mask the sums were
// formed with: 0 = dz is final, 2 = recomputed from x and the statistics; 1 = from z).
static inline int bn_bwd_tiles_nsplit(int nt) { return cdiv(nt, 256); }
extern "C" size_t stil_bn_bwd_tiles_workspace_bytes(int ntiles, int C) {
  if (ntiles <= 0 || C <= 0) return 0;
  return (size_t)2 * bn_bwd_tiles_nsplit(ntiles) * C * sizeof(double);
}
extern "C" int stil_bn_train_bwd_tiles(const float* dz, const float* z, const float* x, const float* gamma, const float* stats,
                                       const float* tilestats, int ntiles, float* dx, float* dgamma, float* dbeta, float* coef,
                                       int M, int C, int relu, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(dz && x && gamma && stats && tilestats && dx && coef && workspace, "stil_bn_train_bwd_tiles: null pointer");
  STIL_REQUIRE(relu >= 0 && relu <= 2 && (relu != 1 || z), "stil_bn_train_bwd_tiles: relu must be 0, 1 (needs z) or 2");
  STIL_REQUIRE(ntiles > 0 && M > 0 && C % 4 == 0, "stil_bn_train_bwd_tiles: bad shape M=%d C=%d ntiles=%d", M, C, ntiles);
  STIL_REQUIRE(workspace_bytes >= stil_bn_bwd_tiles_workspace_bytes(ntiles, C) && ((uintptr_t)workspace % 8) == 0,
               "stil_bn_train_bwd_tiles: workspace too small or not 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int nsplit = bn_bwd_tiles_nsplit(ntiles);
  if (nsplit == 1) {
    hipLaunchKernelGGL(bn_bwd_tiles_fused_kernel, dim3(cdiv(C, 32)), dim3(256), 0, s, tilestats, ntiles, M, C, gamma, stats, dgamma, dbeta,
                       coef, accumulate);
    STIL_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(bn_bwd_tiles_stage1_kernel, dim3(cdiv(C, 32), nsplit), dim3(256), 0, s, tilestats, ntiles, C, cdiv(ntiles, nsplit),
                       (double*)workspace);
    STIL_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_tiles_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const double*)workspace, nsplit, M, C, gamma, stats,
                       dgamma, dbeta, coef, accumulate);
    STIL_LAUNCH_CHECK();
  }
  long total4 = (long)M * C / 4;
  int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(grid), dim3(256), 0, s, dz, z, x, stats, coef, dx, total4, C, relu);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, int OH,
                                     int OW, void* stream) {
  STIL_REQUIRE(x && y && idx, "stil_maxpool3x3s2_fwd: null pointer");
  STIL_REQUIRE(OH == (H + 2 - 3) / 2 + 1 && OW == (W + 2 - 3) / 2 + 1, "stil_maxpool3x3s2_fwd: bad output dims");
  STIL_REQUIRE(C % 4 == 0, "stil_maxpool3x3s2_fwd: C=%d must be a multiple of 4", C);
  long total = (long)N * OH * OW * C / 4;
  int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, H, W, C, OH, OW);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C,
                                     int OH, int OW, void* stream) {
  STIL_REQUIRE(dy && idx && dx, "stil_maxpool3x3s2_bwd: null pointer");
  STIL_REQUIRE(C % 4 == 0, "stil_maxpool3x3s2_bwd: C=%d must be a multiple of 4", C);
  long total = (long)N * H * W * C / 4;
  int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, N, H, W, C, OH, OW);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_im2col_nchw(const float* x, float* col, int N, int Cin, int H, int W, int OH, int OW, int KH,
                                int KW, int stride, int pad, int Kp, void* stream) {
  STIL_REQUIRE(x && col && Kp >= Cin * KH * KW, "stil_im2col_nchw: null pointer or Kp too small");
  STIL_REQUIRE(Kp % 4 == 0 && Kp <= 4096 && KW < 256 && ((uintptr_t)col % 16) == 0, "stil_im2col_nchw: Kp=%d must be a multiple of 4 (<= 4096), KW < 256, col 16-byte aligned", Kp);
  const long M = (long)N * OH * OW;
  const int rpb = 32;
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3(cdiv(M, rpb)), dim3(256), Kp * sizeof(int), (hipStream_t)stream, x, col, N, Cin, H, W,
                     OH, OW, KH, KW, stride, pad, Kp, rpb);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
