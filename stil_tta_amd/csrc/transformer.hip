// Transformer-side kernels: LayerNorm, fused small-sequence attention (S <= 128 lives in LDS),
// activation backward, dropout / drop-path, tabular embedding, token means, transposes.
// Replaces the ATen sequences of models/Transformer.py:63-88,165-174,240-259 and
// models/Disentangle/utils/disentangle_transformer.py:49-94,151-169.
#include "common.h"

// ---------------------------------------------------------------- LayerNorm (one wave per row)
// VEC4 (D % 4 == 0, 16-byte aligned rows): every lane owns quads of consecutive columns -- one dwordx4 per operand
// instead of four dwords (these passes are bound by the number of memory instructions, not by bytes)
template <bool VEC4>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             float* __restrict__ mean_rstd, int rows, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * D;
  float* yr = y + (long)row * D;
  if constexpr (VEC4) {
    float s = 0.f;
    for (int i = lane * 4; i < D; i += 256) { const float4 v = *reinterpret_cast<const float4*>(xr + i); s += (v.x + v.y) + (v.z + v.w); }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + i);
      const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int i = lane * 4; i < D; i += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + i), gm = *reinterpret_cast<const float4*>(gamma + i),
                   bt = *reinterpret_cast<const float4*>(beta + i);
      float4 o;
      o.x = (v.x - mean) * rstd * gm.x + bt.x; o.y = (v.y - mean) * rstd * gm.y + bt.y;
      o.z = (v.z - mean) * rstd * gm.z + bt.z; o.w = (v.w - mean) * rstd * gm.w + bt.w;
      *reinterpret_cast<float4*>(yr + i) = o;
    }
    if (lane == 0) { mean_rstd[2 * row] = mean; mean_rstd[2 * row + 1] = rstd; }
    return;
  }
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int i = lane; i < D; i += 64) { float d = xr[i] - mean; v += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)D + eps);
  for (int i = lane; i < D; i += 64) yr[i] = (xr[i] - mean) * rstd * gamma[i] + beta[i];
  if (lane == 0) { mean_rstd[2 * row] = mean; mean_rstd[2 * row + 1] = rstd; }
}

// dx per row; per-block column partials of dgamma (g*xhat) and dbeta (g): part[blk][2][D]
template <bool VEC4>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ mean_rstd, float* __restrict__ dx,
                                                             float* __restrict__ part, int rows, int D,
                                                             int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [4][2][D]
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float* my = sh + (long)w * 2 * D;
  for (int i = lane; i < 2 * D; i += 64) my[i] = 0.f;
  for (int row = r0 + w; row < r1; row += 4) {
    const float* xr = x + (long)row * D;
    const float* gr = g + (long)row * D;
    float* dr = dx + (long)row * D;
    const float mean = mean_rstd[2 * row], rstd = mean_rstd[2 * row + 1];
    float a = 0.f, b = 0.f;
    if constexpr (VEC4) {   // lane-private column quads: no race on `my`
      for (int i = lane * 4; i < D; i += 256) {
        const float4 gv = *reinterpret_cast<const float4*>(gr + i), xv = *reinterpret_cast<const float4*>(xr + i),
                     gm = *reinterpret_cast<const float4*>(gamma + i);
        const float h0 = (xv.x - mean) * rstd, h1 = (xv.y - mean) * rstd, h2 = (xv.z - mean) * rstd, h3 = (xv.w - mean) * rstd;
        const float g0 = gv.x * gm.x, g1 = gv.y * gm.y, g2 = gv.z * gm.z, g3 = gv.w * gm.w;
        a += (g0 + g1) + (g2 + g3); b += (g0 * h0 + g1 * h1) + (g2 * h2 + g3 * h3);
        float4 m0 = *reinterpret_cast<float4*>(my + i), m1 = *reinterpret_cast<float4*>(my + D + i);
        m0.x += gv.x * h0; m0.y += gv.y * h1; m0.z += gv.z * h2; m0.w += gv.w * h3;
        m1.x += gv.x; m1.y += gv.y; m1.z += gv.z; m1.w += gv.w;
        *reinterpret_cast<float4*>(my + i) = m0; *reinterpret_cast<float4*>(my + D + i) = m1;
      }
      a = wave_sum(a) / (float)D; b = wave_sum(b) / (float)D;
      for (int i = lane * 4; i < D; i += 256) {
        const float4 gv = *reinterpret_cast<const float4*>(gr + i), xv = *reinterpret_cast<const float4*>(xr + i),
                     gm = *reinterpret_cast<const float4*>(gamma + i);
        float4 o;
        o.x = rstd * (gv.x * gm.x - a - (xv.x - mean) * rstd * b); o.y = rstd * (gv.y * gm.y - a - (xv.y - mean) * rstd * b);
        o.z = rstd * (gv.z * gm.z - a - (xv.z - mean) * rstd * b); o.w = rstd * (gv.w * gm.w - a - (xv.w - mean) * rstd * b);
        *reinterpret_cast<float4*>(dr + i) = o;
      }
      continue;
    }
    for (int i = lane; i < D; i += 64) {
      float gg = gr[i] * gamma[i], xh = (xr[i] - mean) * rstd;
      a += gg; b += gg * xh;
      my[i] += gr[i] * xh;      // lane-private columns: no race
      my[D + i] += gr[i];
    }
    a = wave_sum(a) / (float)D; b = wave_sum(b) / (float)D;
    for (int i = lane; i < D; i += 64) {
      float xh = (xr[i] - mean) * rstd;
      dr[i] = rstd * (gr[i] * gamma[i] - a - xh * b);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256)
    part[(long)blockIdx.x * 2 * D + i] = sh[i] + sh[2 * D + i] + sh[4 * D + i] + sh[6 * D + i];
}

__global__ __launch_bounds__(256) void ln_final_kernel(const float* __restrict__ part, int nblk, int D, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, int accumulate) {
  // part: [nblk][2][D]  ->  view as 2*D columns of nblk chunks
  __shared__ float sh[8 * 32];
  int c;
  float tot[1];
  if (!chunk_reduce<8, 1>(part, nblk, 2 * D, sh, c, tot)) return;
  float* dst = c < D ? dgamma + c : dbeta + (c - D);
  *dst = accumulate ? *dst + tot[0] : tot[0];
}

// ---------------------------------------------------------------- fused attention, one block per (batch, head)
struct AttnArgs {
  const float* qkv;  // [B, T, 3, H, d]
  float* out;        // fwd: [B, T, H*d]   (rows q_off..q_off+Sq)
  float* probs;      // [B, H, Sq, Skv] softmax (pre-dropout)
  const unsigned char* mask;  // optional keep-mask [B, H, Sq, Skv]
  const float* dout;  // bwd
  float* dqkv;        // bwd: accumulated (+=) into [B, T, 3, H, d]
  int B, T, H, d, q_off, Sq, kv_off, Skv;
  float scale, drop_scale;
};

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int d = p.d, dp = d + 4, Sq = p.Sq, Skv = p.Skv, d4 = d >> 2;
  float* Qs = sh;                 // [Sq][d]
  float* Ks = Qs + Sq * d;        // [Skv][dp]  (K, then V)
  float* Ps = Ks + Skv * dp;      // [Sq][Skv]
  const long ld = 3L * p.H * d;
  const float* base = p.qkv + (long)b * p.T * ld + h * d;
  const int tid = threadIdx.x;
  for (int i = tid; i < Sq * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Qs + r * d + c * 4) = *reinterpret_cast<const float4*>(base + (long)(p.q_off + r) * ld + c * 4);
  }
  for (int i = tid; i < Skv * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Ks + r * dp + c * 4) =
        *reinterpret_cast<const float4*>(base + (long)(p.kv_off + r) * ld + p.H * d + c * 4);
  }
  __syncthreads();
  for (int e = tid; e < Sq * Skv; e += 256) {
    int i = e / Skv, j = e - i * Skv;
    const float4* q = reinterpret_cast<const float4*>(Qs + i * d);
    const float4* k = reinterpret_cast<const float4*>(Ks + j * dp);
    float s = 0.f;
    for (int c = 0; c < d4; ++c) {
      float4 a = q[c], bb = k[c];
      s += a.x * bb.x; s += a.y * bb.y; s += a.z * bb.z; s += a.w * bb.w;
    }
    Ps[e] = s * p.scale;
  }
  __syncthreads();
  // V overwrites K (all score reads are done)
  for (int i = tid; i < Skv * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Ks + r * dp + c * 4) =
        *reinterpret_cast<const float4*>(base + (long)(p.kv_off + r) * ld + 2 * p.H * d + c * 4);
  }
  const int w = tid >> 6, lane = tid & 63;
  float* pg = p.probs + ((long)(b * p.H + h) * Sq) * Skv;
  const unsigned char* mk = p.mask ? p.mask + ((long)(b * p.H + h) * Sq) * Skv : nullptr;
  for (int i = w; i < Sq; i += 4) {
    float* pr = Ps + i * Skv;
    float m = -INFINITY;
    for (int j = lane; j < Skv; j += 64) m = fmaxf(m, pr[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Skv; j += 64) { float e = expf(pr[j] - m); pr[j] = e; s += e; }
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int j = lane; j < Skv; j += 64) {
      float pv = pr[j] * inv;
      pg[(long)i * Skv + j] = pv;
      if (mk) pv = mk[(long)i * Skv + j] ? pv * p.drop_scale : 0.f;
      pr[j] = pv;
    }
  }
  __syncthreads();
  float* ob = p.out + ((long)b * p.T + p.q_off) * (p.H * d) + h * d;
  for (int e = tid; e < Sq * d4; e += 256) {
    int i = e / d4, c = e - i * d4;
    const float* pr = Ps + i * Skv;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Skv; ++j) {
      const float pj = pr[j];
      const float4 v = *reinterpret_cast<const float4*>(Ks + j * dp + c * 4);
      acc.x += pj * v.x; acc.y += pj * v.y; acc.z += pj * v.z; acc.w += pj * v.w;
    }
    *reinterpret_cast<float4*>(ob + (long)i * (p.H * d) + c * 4) = acc;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int d = p.d, dp = d + 4, Sq = p.Sq, Skv = p.Skv, d4 = d >> 2;
  const int Smax = Sq > Skv ? Sq : Skv;
  float* Ab = sh;                  // [Smax][dp]  dO, later Q
  float* Bb = Ab + Smax * dp;      // [Smax][dp]  V, later K
  float* dS = Bb + Smax * dp;      // [Sq][Skv]
  float* Pd = dS + Sq * Skv;       // [Sq][Skv]  dropped probs
  const long ld = 3L * p.H * d;
  const float* base = p.qkv + (long)b * p.T * ld + h * d;
  float* dbase = p.dqkv + (long)b * p.T * ld + h * d;
  const float* dob = p.dout + ((long)b * p.T + p.q_off) * (p.H * d) + h * d;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int i = tid; i < Sq * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Ab + r * dp + c * 4) = *reinterpret_cast<const float4*>(dob + (long)r * (p.H * d) + c * 4);
  }
  for (int i = tid; i < Skv * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Bb + r * dp + c * 4) =
        *reinterpret_cast<const float4*>(base + (long)(p.kv_off + r) * ld + 2 * p.H * d + c * 4);
  }
  __syncthreads();
  const float* pg = p.probs + ((long)(b * p.H + h) * Sq) * Skv;
  const unsigned char* mk = p.mask ? p.mask + ((long)(b * p.H + h) * Sq) * Skv : nullptr;
  // dPd = dO V^T ; dP = dPd * mask * drop_scale
  for (int e = tid; e < Sq * Skv; e += 256) {
    int i = e / Skv, j = e - i * Skv;
    const float4* a = reinterpret_cast<const float4*>(Ab + i * dp);
    const float4* v = reinterpret_cast<const float4*>(Bb + j * dp);
    float s = 0.f;
    for (int c = 0; c < d4; ++c) {
      float4 x = a[c], y = v[c];
      s += x.x * y.x; s += x.y * y.y; s += x.z * y.z; s += x.w * y.w;
    }
    float keep = 1.f;
    if (mk) keep = mk[e] ? p.drop_scale : 0.f;
    const float pv = pg[e];
    dS[e] = s * keep;      // dP
    Pd[e] = pv * keep;     // dropped probs (for dV)
  }
  __syncthreads();
  // dS = P * (dP - rowsum(dP*P)) * scale
  for (int i = w; i < Sq; i += 4) {
    float s = 0.f;
    for (int j = lane; j < Skv; j += 64) s += dS[i * Skv + j] * pg[(long)i * Skv + j];
    s = wave_sum(s);
    for (int j = lane; j < Skv; j += 64) dS[i * Skv + j] = pg[(long)i * Skv + j] * (dS[i * Skv + j] - s) * p.scale;
  }
  __syncthreads();
  // dV[j][c] += sum_i Pd[i][j] dO[i][c]
  for (int e = tid; e < Skv * d4; e += 256) {
    int j = e / d4, c = e - j * d4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < Sq; ++i) {
      const float pj = Pd[i * Skv + j];
      const float4 v = *reinterpret_cast<const float4*>(Ab + i * dp + c * 4);
      acc.x += pj * v.x; acc.y += pj * v.y; acc.z += pj * v.z; acc.w += pj * v.w;
    }
    float4* dst = reinterpret_cast<float4*>(dbase + (long)(p.kv_off + j) * ld + 2 * p.H * d + c * 4);
    float4 o = *dst;
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *dst = o;
  }
  __syncthreads();
  // K -> Bb ; Q -> Ab
  for (int i = tid; i < Skv * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Bb + r * dp + c * 4) =
        *reinterpret_cast<const float4*>(base + (long)(p.kv_off + r) * ld + p.H * d + c * 4);
  }
  for (int i = tid; i < Sq * d4; i += 256) {
    int r = i / d4, c = i - r * d4;
    *reinterpret_cast<float4*>(Ab + r * dp + c * 4) = *reinterpret_cast<const float4*>(base + (long)(p.q_off + r) * ld + c * 4);
  }
  __syncthreads();
  // dQ[i][c] += sum_j dS[i][j] K[j][c]
  for (int e = tid; e < Sq * d4; e += 256) {
    int i = e / d4, c = e - i * d4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Skv; ++j) {
      const float s = dS[i * Skv + j];
      const float4 v = *reinterpret_cast<const float4*>(Bb + j * dp + c * 4);
      acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w;
    }
    float4* dst = reinterpret_cast<float4*>(dbase + (long)(p.q_off + i) * ld + c * 4);
    float4 o = *dst;
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *dst = o;
  }
  // dK[j][c] += sum_i dS[i][j] Q[i][c]   (different destination rows/slot than dQ: no hazard within the block)
  for (int e = tid; e < Skv * d4; e += 256) {
    int j = e / d4, c = e - j * d4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < Sq; ++i) {
      const float s = dS[i * Skv + j];
      const float4 v = *reinterpret_cast<const float4*>(Ab + i * dp + c * 4);
      acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w;
    }
    float4* dst = reinterpret_cast<float4*>(dbase + (long)(p.kv_off + j) * ld + p.H * d + c * 4);
    float4 o = *dst;
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *dst = o;
  }
}

// ---------------------------------------------------------------- MFMA attention (v_mfma_f32_16x16x4_f32, exact fp32)
// Same contract as attn_fwd_kernel / attn_bwd_kernel (one workgroup per (batch, head)); the six matrix products run on
// the matrix pipe as 16x16 output tiles (sequence lengths are padded to multiples of 16 with zero rows, so S = 65 costs
// 80, not 96 or 128), every operand is read from LDS once per MFMA instead of once per FMA.  fp32 MFMA has the VALU's
// FLOP rate: the gain is LDS traffic and instruction count (the VALU kernels run at ~10 % of the FMA rate).
// Operand maps of 16x16x4: A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15], C/D[col = l & 15][row = 4 (l >> 4) + r].
// LDS row strides are == 20 (mod 32) floats: a matrix read "row = l & 15, k = l >> 4" is then conflict-free
// (s i mod 64 takes 16 distinct multiples of 4 for s = 4 * odd), and the transposed reads (row = l >> 4, column = l & 15)
// that the backward needs for dS^T / Pd^T / the B operands overlap on at most 12 of 64 banks.  The operands the LATER
// phases need (V in forward; K and Q in backward) are fetched into registers at kernel start, so only one global-memory
// latency is exposed per workgroup.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ static inline int attn_pad(int n) { return ((n + 4 - 20 + 31) / 32) * 32 + 20; }   // smallest s >= n + 4 with s == 20 (mod 32)
__host__ __device__ static inline int attn_up16(int n) { return (n + 15) & ~15; }

// rows [r0, r1) of a [*, d] matrix with row stride ld (floats) -> LDS rows of stride dp; rows >= nvalid are zero-filled
__device__ __forceinline__ void attn_stage(float* dst, int dp, const float* src, long ld, int nvalid, int nrows, int d4) {
  for (int i = threadIdx.x; i < nrows * d4; i += blockDim.x) {
    const int r = i / d4, c = i - r * d4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nvalid) v = *reinterpret_cast<const float4*>(src + (long)r * ld + c * 4);
    *reinterpret_cast<float4*>(dst + r * dp + c * 4) = v;
  }
}

// The k loops run 4 MFMA steps (16 k) per iteration with the 8 operand reads issued first: K is a multiple of 16.
// acc += X[ti*16 .. +16, :K] . Y[tj*16 .. +16, :K]^T   (both row-major in LDS: "NT" product, e.g. Q K^T, dO V^T)
__device__ __forceinline__ f32x4 attn_tile_nt(const float* X, int sx, const float* Y, int sy, int ti, int tj, int K, int lane) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* a = X + (ti * 16 + (lane & 15)) * sx + (lane >> 4);
  const float* b = Y + (tj * 16 + (lane & 15)) * sy + (lane >> 4);
  for (int k = 0; k < K; k += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { av[u] = a[k + 4 * u]; bv[u] = b[k + 4 * u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
  return acc;
}
// acc += X[ti*16 .. +16, :K] . Y[:K, tc*16 .. +16]        ("NN": P V, dS K)
__device__ __forceinline__ f32x4 attn_tile_nn(const float* X, int sx, const float* Y, int sy, int ti, int tc, int K, int lane) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* a = X + (ti * 16 + (lane & 15)) * sx + (lane >> 4);
  const float* b = Y + (lane >> 4) * sy + tc * 16 + (lane & 15);
  for (int k = 0; k < K; k += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { av[u] = a[k + 4 * u]; bv[u] = b[(k + 4 * u) * sy]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
  return acc;
}
// acc += X[:K, tj*16 .. +16]^T . Y[:K, tc*16 .. +16]      ("TN": Pd^T dO, dS^T Q)
__device__ __forceinline__ f32x4 attn_tile_tn(const float* X, int sx, const float* Y, int sy, int tj, int tc, int K, int lane) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* a = X + (lane >> 4) * sx + tj * 16 + (lane & 15);
  const float* b = Y + (lane >> 4) * sy + tc * 16 + (lane & 15);
  for (int k = 0; k < K; k += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { av[u] = a[(k + 4 * u) * sx]; bv[u] = b[(k + 4 * u) * sy]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
  return acc;
}

// register prefetch of a [nvalid, d] matrix (row stride ld) for a later attn_unstage into LDS: NPF float4 per thread
// cover nrows * d / 4 <= blockDim.x * NPF elements (the launcher checks); rows >= nvalid read as zero
template <int NPF>
__device__ __forceinline__ void attn_prefetch(float4 (&r)[NPF], const float* src, long ld, int nvalid, int nrows, int d4) {
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    const int i = threadIdx.x + blockDim.x * u;
    const int row = i / d4, c = i - row * d4;
    r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nrows * d4 && row < nvalid) r[u] = *reinterpret_cast<const float4*>(src + (long)row * ld + c * 4);
  }
}
template <int NPF>
__device__ __forceinline__ void attn_unstage(float* dst, int dp, const float4 (&r)[NPF], int nrows, int d4) {
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    const int i = threadIdx.x + blockDim.x * u;
    const int row = i / d4, c = i - row * d4;
    if (i < nrows * d4) *reinterpret_cast<float4*>(dst + row * dp + c * 4) = r[u];
  }
}
constexpr int ATTN_NPF = 16;   // 256-thread workgroups: up to 128 rows x 128 floats (or 256 x 64) per prefetched matrix; NT threads: 16 * 256 / NT

// NT threads = NT / 64 waves per (batch, head): every phase deals its 16x16 tiles round-robin over the waves, so a 512-thread
// workgroup walks each phase in half the tile-steps with the same LDS footprint -- twice the waves per CU for a kernel that is
// bound by its own latency chain at 2 workgroups per CU (used when a phase has >= 16 tiles).
template <int NT>
__global__ __launch_bounds__(NT) void attn_fwd_mfma_kernel(AttnArgs p) {
  constexpr int NW = NT / 64, NPF = ATTN_NPF * 256 / NT;
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int d = p.d, d4 = d >> 2, Sq = p.Sq, Skv = p.Skv, SqP = attn_up16(Sq), SkP = attn_up16(Skv);
  const int dq = d + 4, dp = attn_pad(d), sp = attn_pad(SkP);   // Q, K: read "row = l & 15" only -> stride d + 4 (== 4 mod 64) is conflict-free
  float* Qs = sh;                  // [SqP][dq]
  float* KV = Qs + SqP * dq;       // K as [SkP][dq], then V as [SkP][dp]
  float* Ps = KV + SkP * dp;       // [SqP][sp]
  const long ld = 3L * p.H * d;
  const float* base = p.qkv + (long)b * p.T * ld + h * d;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  float4 vreg[NPF];
  attn_stage(Qs, dq, base + (long)p.q_off * ld, ld, Sq, SqP, d4);
  attn_stage(KV, dq, base + (long)p.kv_off * ld + p.H * d, ld, Skv, SkP, d4);
  attn_prefetch<NPF>(vreg, base + (long)p.kv_off * ld + 2 * p.H * d, ld, Skv, SkP, d4);   // V: in flight under Q K^T
  __syncthreads();
  const int ntq = SqP >> 4, ntk = SkP >> 4, ntc = d >> 4;
  for (int t = w; t < ntq * ntk; t += NW) {                       // S = scale * Q K^T
    const int ti = t / ntk, tj = t - ti * ntk;
    const f32x4 acc = attn_tile_nt(Qs, dq, KV, dq, ti, tj, d, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ps[(ti * 16 + (lane >> 4) * 4 + r) * sp + tj * 16 + (lane & 15)] = acc[r] * p.scale;
  }
  __syncthreads();
  attn_unstage<NPF>(KV, dp, vreg, SkP, d4);                 // V overwrites K
  float* pg = p.probs + ((long)(b * p.H + h) * Sq) * Skv;
  const unsigned char* mk = p.mask ? p.mask + ((long)(b * p.H + h) * Sq) * Skv : nullptr;
  for (int i = w; i < Sq; i += NW) {                              // softmax over the valid columns; padding stays 0
    float* pr = Ps + i * sp;
    float m = -INFINITY;
    for (int j = lane; j < Skv; j += 64) m = fmaxf(m, pr[j]);
    m = wave_max(m);
    float sacc = 0.f;
    for (int j = lane; j < Skv; j += 64) { float e = expf(pr[j] - m); pr[j] = e; sacc += e; }
    sacc = wave_sum(sacc);
    const float inv = 1.f / sacc;
    for (int j = lane; j < Skv; j += 64) {
      float pv = pr[j] * inv;
      pg[(long)i * Skv + j] = pv;
      if (mk) pv = mk[(long)i * Skv + j] ? pv * p.drop_scale : 0.f;
      pr[j] = pv;
    }
  }
  __syncthreads();
  float* ob = p.out + ((long)b * p.T + p.q_off) * (p.H * d) + h * d;
  for (int t = w; t < ntq * ntc; t += NW) {                       // O = P V
    const int ti = t / ntc, tc = t - ti * ntc;
    const f32x4 acc = attn_tile_nn(Ps, sp, KV, dp, ti, tc, SkP, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ti * 16 + (lane >> 4) * 4 + r;
      if (i < Sq) ob[(long)i * (p.H * d) + tc * 16 + (lane & 15)] = acc[r];
    }
  }
}

// stride for matrices that are only read "row = l >> 4, column = l & 15" (B operands K, Q of the last phase): == 16 (mod 64)
__host__ __device__ static inline int attn_padT(int d) { return ((d - 16 + 63) / 64) * 64 + 16; }

template <int NT>
__global__ __launch_bounds__(NT) void attn_bwd_mfma_kernel(AttnArgs p) {
  constexpr int NW = NT / 64, NPF = ATTN_NPF * 256 / NT;
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int d = p.d, d4 = d >> 2, Sq = p.Sq, Skv = p.Skv, SqP = attn_up16(Sq), SkP = attn_up16(Skv);
  const int dq = d + 4, dp = attn_pad(d), dt = attn_padT(d), sp = attn_pad(SkP);
  // Each region is sized by its LARGER occupant (dt > dp for d = 32, 48, 96, 112; dq > dt for d = 16, 80, 144): attn_mfma_lds
  float* R0 = sh;                                // dO as [SqP][dp], later Q as [SqP][dt]
  float* R1 = R0 + SqP * (dp > dt ? dp : dt);    // V as [SkP][dq], later K as [SkP][dt]
  float* X = R1 + SkP * (dq > dt ? dq : dt);     // [SqP][sp]: dropped probabilities Pd, later dP -> dS (time-shared: one tile matrix, not two)
  const long ld = 3L * p.H * d;
  const float* base = p.qkv + (long)b * p.T * ld + h * d;
  float* dbase = p.dqkv + (long)b * p.T * ld + h * d;
  const float* dob = p.dout + ((long)b * p.T + p.q_off) * (p.H * d) + h * d;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const float* pg = p.probs + ((long)(b * p.H + h) * Sq) * Skv;
  const unsigned char* mk = p.mask ? p.mask + ((long)(b * p.H + h) * Sq) * Skv : nullptr;
  const int ntq = SqP >> 4, ntk = SkP >> 4, ntc = d >> 4;
  float4 kreg[NPF], qreg[NPF];
  attn_stage(R0, dp, dob, (long)p.H * d, Sq, SqP, d4);
  attn_stage(R1, dq, base + (long)p.kv_off * ld + 2 * p.H * d, ld, Skv, SkP, d4);
  for (int e = tid; e < SqP * SkP; e += NT) {                   // Pd = P * keep (zero in the padding)
    const int i = e / SkP, j = e - i * SkP;
    float v = 0.f;
    if (i < Sq && j < Skv) {
      const long g = (long)i * Skv + j;
      v = pg[g] * (mk ? (mk[g] ? p.drop_scale : 0.f) : 1.f);
    }
    X[i * sp + j] = v;
  }
  attn_prefetch<NPF>(kreg, base + (long)p.kv_off * ld + p.H * d, ld, Skv, SkP, d4);   // K, Q: needed only by the last phase
  attn_prefetch<NPF>(qreg, base + (long)p.q_off * ld, ld, Sq, SqP, d4);
  __syncthreads();
  for (int t = w; t < ntk * ntc; t += NW) {                       // dV += Pd^T dO
    const int tj = t / ntc, tc = t - tj * ntc;
    const f32x4 acc = attn_tile_tn(X, sp, R0, dp, tj, tc, SqP, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = tj * 16 + (lane >> 4) * 4 + r;
      if (j < Skv) dbase[(long)(p.kv_off + j) * ld + 2 * p.H * d + tc * 16 + (lane & 15)] += acc[r];
    }
  }
  __syncthreads();                                               // Pd is dead: X is free for dP
  for (int t = w; t < ntq * ntk; t += NW) {                       // dP = (dO V^T) * keep
    const int ti = t / ntk, tj = t - ti * ntk;
    const f32x4 acc = attn_tile_nt(R0, dp, R1, dq, ti, tj, d, lane);
    const int j = tj * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ti * 16 + (lane >> 4) * 4 + r;
      float dpv = 0.f;
      if (i < Sq && j < Skv) dpv = acc[r] * (mk ? (mk[(long)i * Skv + j] ? p.drop_scale : 0.f) : 1.f);
      X[i * sp + j] = dpv;
    }
  }
  __syncthreads();                                               // dO and V are dead: R0 / R1 take Q / K
  attn_unstage<NPF>(R1, dt, kreg, SkP, d4);
  attn_unstage<NPF>(R0, dt, qreg, SqP, d4);
  for (int i = w; i < Sq; i += NW) {                              // dS = P * (dP - rowsum(dP * P)) * scale
    float sacc = 0.f;
    for (int j = lane; j < Skv; j += 64) sacc += X[i * sp + j] * pg[(long)i * Skv + j];
    sacc = wave_sum(sacc);
    for (int j = lane; j < Skv; j += 64) X[i * sp + j] = pg[(long)i * Skv + j] * (X[i * sp + j] - sacc) * p.scale;
  }
  __syncthreads();
  for (int t = w; t < ntq * ntc; t += NW) {                       // dQ += dS K
    const int ti = t / ntc, tc = t - ti * ntc;
    const f32x4 acc = attn_tile_nn(X, sp, R1, dt, ti, tc, SkP, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ti * 16 + (lane >> 4) * 4 + r;
      if (i < Sq) dbase[(long)(p.q_off + i) * ld + tc * 16 + (lane & 15)] += acc[r];
    }
  }
  for (int t = w; t < ntk * ntc; t += NW) {                       // dK += dS^T Q   (other slot of dqkv than dQ: no hazard)
    const int tj = t / ntc, tc = t - tj * ntc;
    const f32x4 acc = attn_tile_tn(X, sp, R0, dt, tj, tc, SqP, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = tj * 16 + (lane >> 4) * 4 + r;
      if (j < Skv) dbase[(long)(p.kv_off + j) * ld + p.H * d + tc * 16 + (lane & 15)] += acc[r];
    }
  }
}

// ---------------------------------------------------------------- elementwise
// kind 1: dx = dy * (ref > 0)   (ref = relu output) ; kind 2: dx = dy * gelu'(ref)  (ref = pre-activation)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref, float* __restrict__ dx,
                               long n, int kind) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float r = ref[i];
    dx[i] = kind == 1 ? (r > 0.f ? dy[i] : 0.f) : dy[i] * gelu_grad_f(r);
  }
}

// four elements per thread (n % 4 == 0, 16-byte aligned pointers): one dwordx4 per operand instead of four dwords -- these
// passes are bound by the number of memory instructions long before they are bound by bytes
__global__ void act_bwd4_kernel(const float4* __restrict__ dy, const float4* __restrict__ ref, float4* __restrict__ dx,
                                long n4, int kind) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 r = ref[i], g = dy[i];
    float4 o;
    if (kind == 1) {
      o.x = r.x > 0.f ? g.x : 0.f; o.y = r.y > 0.f ? g.y : 0.f; o.z = r.z > 0.f ? g.z : 0.f; o.w = r.w > 0.f ? g.w : 0.f;
    } else {
      o.x = g.x * gelu_grad_f(r.x); o.y = g.y * gelu_grad_f(r.y); o.z = g.z * gelu_grad_f(r.z); o.w = g.w * gelu_grad_f(r.w);
    }
    dx[i] = o;
  }
}
__global__ void drop_add4_kernel(const float4* __restrict__ x, const float4* __restrict__ resid,
                                 const uchar4* __restrict__ emask, const unsigned char* __restrict__ rmask,
                                 float4* __restrict__ out, long n4, int rowlen4, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 k = make_float4(scale, scale, scale, scale);
    if (emask) { const uchar4 e = emask[i]; if (!e.x) k.x = 0.f; if (!e.y) k.y = 0.f; if (!e.z) k.z = 0.f; if (!e.w) k.w = 0.f; }
    if (rmask && !rmask[i / rowlen4]) k = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = x[i];
    float4 v = make_float4(a.x * k.x, a.y * k.y, a.z * k.z, a.w * k.w);
    if (resid) { const float4 r = resid[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    out[i] = v;
  }
}

// out = (resid ? resid : 0) + x * emask[i] * rmask[i / rowlen] * scale
__global__ void drop_add_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                const unsigned char* __restrict__ emask, const unsigned char* __restrict__ rmask,
                                float* __restrict__ out, long n, int rowlen, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float k = scale;
    if (emask && !emask[i]) k = 0.f;
    if (rmask && !rmask[i / rowlen]) k = 0.f;
    float v = x[i] * k;
    if (resid) v += resid[i];
    out[i] = v;
  }
}

// out = a*x + b*y (y optional)
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, long n,
                             float a, float b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// keep-mask generation: counter-based hash (seed, offset + i) -> uniform [0,1) >= p
__device__ __forceinline__ unsigned int hash_u32(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned int)(x >> 16);
}
__global__ void rng_mask_kernel(unsigned char* __restrict__ out, long n, unsigned long long seed,
                                unsigned long long offset, float p, const unsigned long long* __restrict__ step) {
  // the per-step part of the counter lives in DEVICE memory so that a captured hipGraph draws fresh masks on replay
  const unsigned long long st = step ? step[0] * 0xD1B54A32D192ED03ULL : 0ULL;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned int r = hash_u32(seed * 0x9E3779B97F4A7C15ULL + st + offset + (unsigned long long)i);
    float u = (float)(r >> 8) * (1.0f / 16777216.0f);
    out[i] = u >= p ? 1 : 0;
  }
}

__global__ void counter_inc_kernel(unsigned long long* c) { if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1; }

// out[c][r] = in[r][c]
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float t[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int k = ty; k < 32; k += 8) {
    int r = r0 + k, c = c0 + tx;
    t[k][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = c0 + k, r = r0 + tx;
    if (r < R && c < C) out[(long)c * R + r] = t[tx][k];
  }
}

// conv weight re-layout: w (Cout,Cin,KH,KW) -> wf [Cout][KH*KW][Cin] (forward) and wd [Cin][KH*KW][Cout] (dgrad)
__global__ void conv_w_layout_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                     int Cout, int Cin, int taps) {
  long total = (long)Cout * Cin * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int tap = (int)(i % taps);
    long t = i / taps;
    int ci = (int)(t % Cin), co = (int)(t / Cin);
    float v = w[i];
    if (wf) wf[((long)co * taps + tap) * Cin + ci] = v;
    if (wd) wd[((long)ci * taps + tap) * Cout + co] = v;
  }
}

// phase (py,px) of a stride-s dgrad:  wsub[ci][ky'][kx'][co] = w[co][ci][ky0 + s*(KHs-1-ky')][kx0 + s*(KWs-1-kx')]
__global__ void conv_w_phase_kernel(const float* __restrict__ w, float* __restrict__ wsub, int Cout, int Cin, int KH,
                                    int KW, int s, int ky0, int kx0, int KHs, int KWs) {
  long total = (long)Cin * KHs * KWs * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int co = (int)(i % Cout);
    long t = i / Cout;
    int kx = (int)(t % KWs); t /= KWs;
    int ky = (int)(t % KHs);
    int ci = (int)(t / KHs);
    int sy = ky0 + s * (KHs - 1 - ky), sx = kx0 + s * (KWs - 1 - kx);
    wsub[i] = w[(((long)co * Cin + ci) * KH + sy) * KW + sx];
  }
}

// ---------------------------------------------------------------- tabular embedding
// h[b,0]=cls ; h[b,1+j]=cat_emb[int(x[b,j])+off[j]] (j<ncat) ; h[b,1+j]=x[b,j]*w+bias (j>=ncat) ; + colemb[t]
__global__ void tab_embed_fwd_kernel(const float* __restrict__ x, const int* __restrict__ offs,
                                     const float* __restrict__ emb, const float* __restrict__ con_w,
                                     const float* __restrict__ con_b, const float* __restrict__ cls,
                                     const float* __restrict__ colemb, float* __restrict__ h, int B, int ncols, int ncat,
                                     int D) {
  const int T = ncols + 1;
  long total = (long)B * T * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int dd = (int)(i % D);
    long r = i / D;
    int t = (int)(r % T), b = (int)(r / T);
    float v;
    if (t == 0) v = cls[dd];
    else if (t - 1 < ncat) {
      long code = (long)x[(long)b * ncols + t - 1] + offs[t - 1];
      v = emb[code * D + dd];
    } else v = x[(long)b * ncols + t - 1] * con_w[dd] + con_b[dd];
    h[i] = v + colemb[(long)t * D + dd];
  }
}

// per token position t: dcol[t,d] = sum_b g[b,t,d] ; tmpw[t,d] = sum_b g[b,t,d]*x[b,t-1] (continuous tokens)
__global__ void tab_embed_bwd_tok_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                         float* __restrict__ dcol, float* __restrict__ tmpw, int B, int ncols, int ncat,
                                         int D, int accumulate) {
  const int T = ncols + 1, t = blockIdx.x;
  const bool con = (t >= 1 + ncat);
  for (int dd = threadIdx.x; dd < D; dd += blockDim.x) {
    float s = 0.f, sw = 0.f;
    for (int b = 0; b < B; ++b) {
      float gv = g[((long)b * T + t) * D + dd];
      s += gv;
      if (con) sw += gv * x[(long)b * ncols + t - 1];
    }
    long o = (long)t * D + dd;
    tmpw[o] = sw;
    tmpw[(long)T * D + o] = s;
    dcol[o] = accumulate ? dcol[o] + s : s;
  }
}
// dcls, dcon_w, dcon_b from the per-token sums
__global__ void tab_embed_bwd_fin_kernel(const float* __restrict__ tmpw, float* __restrict__ dcls,
                                         float* __restrict__ dcon_w, float* __restrict__ dcon_b, int ncols, int ncat,
                                         int D, int accumulate) {
  const int T = ncols + 1;
  int dd = blockIdx.x * blockDim.x + threadIdx.x;
  if (dd >= D) return;
  float sw = 0.f, sb = 0.f;
  for (int t = 1 + ncat; t < T; ++t) { sw += tmpw[(long)t * D + dd]; sb += tmpw[(long)T * D + (long)t * D + dd]; }
  float c = tmpw[(long)T * D + dd];
  dcls[dd] = accumulate ? dcls[dd] + c : c;
  if (ncols > ncat) {
    dcon_w[dd] = accumulate ? dcon_w[dd] + sw : sw;
    dcon_b[dd] = accumulate ? dcon_b[dd] + sb : sb;
  }
}
// embedding rows: block per table row r (column rowcol[r]); deterministic loop over the batch
__global__ void tab_embed_bwd_emb_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                         const int* __restrict__ offs, const int* __restrict__ rowcol,
                                         float* __restrict__ demb, int B, int ncols, int D, int accumulate) {
  const int T = ncols + 1, r = blockIdx.x, j = rowcol[r];
  const int code = r - offs[j];
  for (int dd = threadIdx.x; dd < D; dd += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b)
      if ((int)x[(long)b * ncols + j] == code) s += g[((long)b * T + j + 1) * D + dd];
    long o = (long)r * D + dd;
    demb[o] = accumulate ? demb[o] + s : s;
  }
}

// ---------------------------------------------------------------- token mean over dim 1 of [B,T,D]
__global__ void tokmean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int D) {
  long total = (long)B * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int dd = (int)(i % D), b = (int)(i / D);
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[((long)b * T + t) * D + dd];
    y[i] = s / (float)T;
  }
}
__global__ void tokmean_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int B, int T, int D) {
  long total = (long)B * T * D;
  const float inv = 1.f / (float)T;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int dd = (int)(i % D);
    int b = (int)(i / ((long)T * D));
    dx[i] = g[(long)b * D + dd] * inv;
  }
}

// ---------------------------------------------------------------- C ABI
static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g < 16384 ? (g < 1 ? 1 : g) : 16384); }

extern "C" int stil_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd,
                                  int rows, int D, float eps, void* stream) {
  STIL_REQUIRE(x && gamma && beta && y && mean_rstd && rows > 0 && D > 0, "stil_layernorm_fwd: bad arguments");
  const bool v4 = D % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16) == 0;
  if (v4) hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                             mean_rstd, rows, D, eps);
  else hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                          mean_rstd, rows, D, eps);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

static inline int ln_blocks(int rows) { int nb = cdiv(rows, 32); return nb > 1024 ? 1024 : nb; }
extern "C" size_t stil_layernorm_bwd_workspace_bytes(int rows, int D) { return (size_t)ln_blocks(rows) * 2 * D * sizeof(float); }

extern "C" int stil_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean_rstd,
                                  float* dx, float* dgamma, float* dbeta, int rows, int D, int accumulate,
                                  float* workspace, size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(g && x && gamma && mean_rstd && dx && dgamma && dbeta && workspace, "stil_layernorm_bwd: null pointer");
  STIL_REQUIRE(D <= 4096, "stil_layernorm_bwd: D=%d > 4096", D);
  static bool attr_set = false;
  if (!attr_set) {  // 8*D floats of LDS: > 64 KiB for D > 2048 (SAINT's row LayerNorm has D = 32 * nfeats = 2080)
    hipError_t e = hipFuncSetAttribute((const void*)layernorm_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)layernorm_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    if (e != hipSuccess) { stil_set_error("layernorm_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return STIL_EHIP; }
    attr_set = true;
  }
  int nb = ln_blocks(rows);
  STIL_REQUIRE(workspace_bytes >= (size_t)nb * 2 * D * sizeof(float), "stil_layernorm_bwd: workspace too small");
  int rpb = cdiv(rows, nb);
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = D % 4 == 0 && (((uintptr_t)g | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx) % 16) == 0;
  if (v4) hipLaunchKernelGGL(layernorm_bwd_kernel<true>, dim3(nb), dim3(256), (size_t)8 * D * sizeof(float), s, g, x, gamma, mean_rstd,
                             dx, workspace, rows, D, rpb);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<false>, dim3(nb), dim3(256), (size_t)8 * D * sizeof(float), s, g, x, gamma, mean_rstd,
                          dx, workspace, rows, D, rpb);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(ln_final_kernel, dim3(cdiv(2 * D, 32)), dim3(256), 0, s, workspace, nb, D, dgamma, dbeta, accumulate);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

static int attn_check(int T, int H, int d, int q_off, int Sq, int kv_off, int Skv) {
  STIL_REQUIRE(d % 4 == 0 && d > 0, "attention: head dim %d must be a multiple of 4", d);
  STIL_REQUIRE(Sq > 0 && Skv > 0 && q_off >= 0 && kv_off >= 0 && q_off + Sq <= T && kv_off + Skv <= T,
               "attention: window out of range (T=%d q=[%d,+%d) kv=[%d,+%d))", T, q_off, Sq, kv_off, Skv);
  (void)H;
  return STIL_OK;
}
// LDS of the MFMA kernels (padded tiles) and their eligibility: head dim a multiple of 16, everything in 160 KiB
static size_t attn_mfma_lds(int Sq, int Skv, int d, bool bwd) {
  const int SqP = attn_up16(Sq), SkP = attn_up16(Skv), dp = attn_pad(d), sp = attn_pad(SkP);
  const int dt = attn_padT(d), dq = d + 4;
  const size_t fl = bwd ? (size_t)SqP * (dp > dt ? dp : dt) + (size_t)SkP * (dq > dt ? dq : dt) + (size_t)SqP * sp     // dO/Q, V/K, Pd -> dP -> dS
                        : (size_t)SqP * (d + 4) + (size_t)SkP * dp + (size_t)SqP * sp;        // Q, K/V, P
  return fl * sizeof(float);
}
static bool attn_mfma_ok(int Sq, int Skv, int d, size_t lds) {
  const int big = attn_up16(Sq > Skv ? Sq : Skv);
  return d % 16 == 0 && d >= 16 && lds <= 160 * 1024 && Sq > 0 && Skv > 0 && (long)big * (d / 4) <= 256L * ATTN_NPF;
}

// 8 waves per (batch, head) once the score matrix has >= 16 tiles of 16 x 16
static bool attn_big_wg(int Sq, int Skv) { return (attn_up16(Sq) >> 4) * (attn_up16(Skv) >> 4) >= 16; }

static bool g_attn_attr_set = false;
static int attn_set_attr() {
  if (g_attn_attr_set) return STIL_OK;
  hipError_t e1 = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e1 == hipSuccess) e1 = hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e1 == hipSuccess) e1 = hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    stil_set_error("attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return STIL_EHIP;
  }
  g_attn_attr_set = true;
  return STIL_OK;
}

extern "C" int stil_attention_fwd(const float* qkv, float* out, float* probs, const unsigned char* mask, int B, int T,
                                  int H, int d, int q_off, int Sq, int kv_off, int Skv, float scale, float drop_p,
                                  void* stream) {
  STIL_REQUIRE(qkv && out && probs, "stil_attention_fwd: null pointer");
  int rc = attn_check(T, H, d, q_off, Sq, kv_off, Skv);
  if (rc) return rc;
  size_t lds = ((size_t)Sq * d + (size_t)Skv * (d + 4) + (size_t)Sq * Skv) * sizeof(float);
  const size_t lds_m = attn_mfma_lds(Sq, Skv, d, false);   // MFMA path: d % 16 == 0 and its padded tiles fit
  // matrix-pipe kernel unless it would keep fewer workgroups resident than the VALU kernel (both are latency-bound)
  const bool big = attn_big_wg(Sq, Skv);   // 8-wave matrix-pipe workgroups: twice the waves of the VALU kernel's even at one workgroup fewer
  const bool mfma = attn_mfma_ok(Sq, Skv, d, lds_m) && (lds > 160 * 1024 || big || (160 * 1024) / lds_m >= (160 * 1024) / lds);
  STIL_REQUIRE(mfma || lds <= 160 * 1024, "stil_attention_fwd: needs %zu B of LDS (> 160 KiB): Sq=%d Skv=%d d=%d", lds, Sq, Skv, d);
  if ((rc = attn_set_attr())) return rc;
  AttnArgs p;
  p.qkv = qkv; p.out = out; p.probs = probs; p.mask = mask; p.dout = nullptr; p.dqkv = nullptr;
  p.B = B; p.T = T; p.H = H; p.d = d; p.q_off = q_off; p.Sq = Sq; p.kv_off = kv_off; p.Skv = Skv;
  p.scale = scale; p.drop_scale = mask ? 1.f / (1.f - drop_p) : 1.f;
  if (mfma && big) hipLaunchKernelGGL(attn_fwd_mfma_kernel<512>, dim3(B * H), dim3(512), lds_m, (hipStream_t)stream, p);
  else if (mfma) hipLaunchKernelGGL(attn_fwd_mfma_kernel<256>, dim3(B * H), dim3(256), lds_m, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, p);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_attention_bwd(const float* dout, const float* qkv, const float* probs, const unsigned char* mask,
                                  float* dqkv, int B, int T, int H, int d, int q_off, int Sq, int kv_off, int Skv,
                                  float scale, float drop_p, void* stream) {
  STIL_REQUIRE(dout && qkv && probs && dqkv, "stil_attention_bwd: null pointer");
  int rc = attn_check(T, H, d, q_off, Sq, kv_off, Skv);
  if (rc) return rc;
  int Smax = Sq > Skv ? Sq : Skv;
  size_t lds = ((size_t)2 * Smax * (d + 4) + (size_t)2 * Sq * Skv) * sizeof(float);
  const size_t lds_m = attn_mfma_lds(Sq, Skv, d, true);
  // matrix-pipe kernel unless it would keep fewer workgroups resident than the VALU kernel (both are latency-bound)
  const bool big = attn_big_wg(Sq, Skv);
  const bool mfma = attn_mfma_ok(Sq, Skv, d, lds_m) && (lds > 160 * 1024 || big || (160 * 1024) / lds_m >= (160 * 1024) / lds);
  STIL_REQUIRE(mfma || lds <= 160 * 1024, "stil_attention_bwd: needs %zu B of LDS (> 160 KiB)", lds);
  if ((rc = attn_set_attr())) return rc;
  AttnArgs p;
  p.qkv = qkv; p.out = nullptr; p.probs = const_cast<float*>(probs); p.mask = mask; p.dout = dout; p.dqkv = dqkv;
  p.B = B; p.T = T; p.H = H; p.d = d; p.q_off = q_off; p.Sq = Sq; p.kv_off = kv_off; p.Skv = Skv;
  p.scale = scale; p.drop_scale = mask ? 1.f / (1.f - drop_p) : 1.f;
  if (mfma && big) hipLaunchKernelGGL(attn_bwd_mfma_kernel<512>, dim3(B * H), dim3(512), lds_m, (hipStream_t)stream, p);
  else if (mfma) hipLaunchKernelGGL(attn_bwd_mfma_kernel<256>, dim3(B * H), dim3(256), lds_m, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, p);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_act_bwd(const float* dy, const float* ref, float* dx, long n, int kind, void* stream) {
  STIL_REQUIRE(dy && ref && dx && (kind == 1 || kind == 2), "stil_act_bwd: bad arguments");
  if (n % 4 == 0 && (((uintptr_t)dy | (uintptr_t)ref | (uintptr_t)dx) % 16) == 0)
    hipLaunchKernelGGL(act_bwd4_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const float4*)ref,
                       (float4*)dx, n / 4, kind);
  else
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, ref, dx, n, kind);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_drop_add(const float* x, const float* resid, const unsigned char* emask, const unsigned char* rmask,
                             float* out, long n, int rowlen, float scale, void* stream) {
  STIL_REQUIRE(x && out && rowlen > 0, "stil_drop_add: bad arguments");
  if (n % 4 == 0 && rowlen % 4 == 0 && (((uintptr_t)x | (uintptr_t)resid | (uintptr_t)out) % 16) == 0 && ((uintptr_t)emask % 4) == 0)
    hipLaunchKernelGGL(drop_add4_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)resid,
                       (const uchar4*)emask, rmask, (float4*)out, n / 4, rowlen / 4, scale);
  else
    hipLaunchKernelGGL(drop_add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, resid, emask, rmask, out,
                       n, rowlen, scale);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_axpby(const float* x, const float* y, float* out, long n, float a, float b, void* stream) {
  STIL_REQUIRE(x && out, "stil_axpby: null pointer");
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, out, n, a, b);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_counter_inc(unsigned long long* counter_dev, void* stream) {
  STIL_REQUIRE(counter_dev, "stil_counter_inc: null pointer");
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter_dev);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_rng_mask(unsigned char* out, long n, unsigned long long seed, unsigned long long offset, float p,
                             const unsigned long long* step_dev, void* stream) {
  STIL_REQUIRE(out && p >= 0.f && p < 1.f, "stil_rng_mask: bad arguments");
  hipLaunchKernelGGL(rng_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset, p, step_dev);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_transpose(const float* in, float* out, int R, int C, void* stream) {
  STIL_REQUIRE(in && out && R > 0 && C > 0, "stil_transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, in, out, R, C);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_conv_weight_layout(const float* w, float* w_fwd, float* w_dgrad, int Cout, int Cin, int KH, int KW,
                                       void* stream) {
  STIL_REQUIRE(w && (w_fwd || w_dgrad), "stil_conv_weight_layout: null pointer");
  long total = (long)Cout * Cin * KH * KW;
  hipLaunchKernelGGL(conv_w_layout_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, w_fwd, w_dgrad,
                     Cout, Cin, KH * KW);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_conv_weight_layout_phase(const float* w, float* w_sub, int Cout, int Cin, int KH, int KW, int stride,
                                             int ky0, int kx0, int KHs, int KWs, void* stream) {
  STIL_REQUIRE(w && w_sub && KHs > 0 && KWs > 0 && ky0 + stride * (KHs - 1) < KH && kx0 + stride * (KWs - 1) < KW,
               "stil_conv_weight_layout_phase: bad arguments");
  long total = (long)Cin * KHs * KWs * Cout;
  hipLaunchKernelGGL(conv_w_phase_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, w_sub, Cout, Cin, KH,
                     KW, stride, ky0, kx0, KHs, KWs);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_tab_embed_fwd(const float* x, const int* cat_offsets, const float* cat_emb, const float* con_w,
                                  const float* con_b, const float* cls, const float* colemb, float* h, int B,
                                  int ncols, int ncat, int D, void* stream) {
  STIL_REQUIRE(x && cls && colemb && h && (ncat == 0 || (cat_offsets && cat_emb)) && (ncols == ncat || (con_w && con_b)),
               "stil_tab_embed_fwd: null pointer");
  long total = (long)B * (ncols + 1) * D;
  hipLaunchKernelGGL(tab_embed_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, cat_offsets,
                     cat_emb, con_w, con_b, cls, colemb, h, B, ncols, ncat, D);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" size_t stil_tab_embed_bwd_workspace_bytes(int ncols, int D) { return (size_t)2 * (ncols + 1) * D * sizeof(float); }

extern "C" int stil_tab_embed_bwd(const float* g, const float* x, const int* cat_offsets, const int* rowcol,
                                  int n_emb_rows, float* d_emb, float* d_con_w, float* d_con_b, float* d_cls,
                                  float* d_colemb, int B, int ncols, int ncat, int D, int accumulate, float* workspace,
                                  size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(g && x && d_cls && d_colemb && workspace, "stil_tab_embed_bwd: null pointer");
  STIL_REQUIRE(workspace_bytes >= stil_tab_embed_bwd_workspace_bytes(ncols, D), "stil_tab_embed_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tab_embed_bwd_tok_kernel, dim3(ncols + 1), dim3(256), 0, s, g, x, d_colemb, workspace, B, ncols,
                     ncat, D, accumulate);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(tab_embed_bwd_fin_kernel, dim3(cdiv(D, 64)), dim3(64), 0, s, workspace, d_cls, d_con_w, d_con_b,
                     ncols, ncat, D, accumulate);
  STIL_LAUNCH_CHECK();
  if (ncat > 0 && n_emb_rows > 0) {
    STIL_REQUIRE(cat_offsets && rowcol && d_emb, "stil_tab_embed_bwd: null categorical pointers");
    hipLaunchKernelGGL(tab_embed_bwd_emb_kernel, dim3(n_emb_rows), dim3(256), 0, s, g, x, cat_offsets, rowcol, d_emb, B,
                       ncols, D, accumulate);
    STIL_LAUNCH_CHECK();
  }
  return STIL_OK;
}

extern "C" int stil_tokmean_fwd(const float* x, float* y, int B, int T, int D, void* stream) {
  STIL_REQUIRE(x && y && T > 0, "stil_tokmean_fwd: bad arguments");
  hipLaunchKernelGGL(tokmean_fwd_kernel, dim3(ew_grid((long)B * D)), dim3(256), 0, (hipStream_t)stream, x, y, B, T, D);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_tokmean_bwd(const float* g, float* dx, int B, int T, int D, void* stream) {
  STIL_REQUIRE(g && dx && T > 0, "stil_tokmean_bwd: bad arguments");
  hipLaunchKernelGGL(tokmean_bwd_kernel, dim3(ew_grid((long)B * T * D)), dim3(256), 0, (hipStream_t)stream, g, dx, B, T, D);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
