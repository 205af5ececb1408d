// fp32-exact MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD).
//
//  * gemm_nt_kernel  : C[M,N] = epilogue( alpha * Agather[M,K] . W[N,K]^T )
//        Agather is either a plain row-major matrix or an implicit-im2col gather of an
//        NHWC tensor (forward conv, or the transposed-conv gather used for dgrad).  It
//        replaces the ATen ops the reference issues for nn.Conv2d / nn.Linear forward and
//        their input-gradients (models/resnets.py:112-132, models/Transformer.py:27-33,63-88).
//  * gemm_tn_kernel  : P[z][N,K] = sum_m dY[m,N] . Xgather[m,K]  (weight gradients, split over m)
//  * wgrad_reduce    : dW = (+=) sum_z P[z], written in the reference's (Cout,Cin,KH,KW) layout
//  * colsum          : bias gradients
//
// Tiling: 256 threads = 4 waves (2x2); each wave owns TMxTN 32x32 accumulators; BK = 16.
// LDS rows are padded to 20 floats: ds_read_b128 of 4 consecutive k is then conflict-free
// for every 16-lane group (i*20 mod 64 distinct multiples of 4), and one b128 read feeds
// four MFMA k-steps.  The k order inside a 8-wide group is permuted identically for A and B.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// a = a1 + a2 + a3 in bf16 terms (8 + 8 + 8 mantissa bits; both remainders are exact in fp32)
__device__ __forceinline__ void split3(float a, __bf16& a1, __bf16& a2, __bf16& a3) {
  a1 = (__bf16)a;
  const float r1 = a - (float)a1;
  a2 = (__bf16)r1;
  const float r2 = r1 - (float)a2;
  a3 = (__bf16)r2;
}

struct ConvGeom {
  int H, W, C;    // source NHWC dims (rows of the source have stride ld)
  int OH, OW;     // the GEMM row index m enumerates (n, oy, ox)
  int KH, KW, stride, pad_y, pad_x;
  int mode;       // 0: forward gather  iy = oy*stride - pad + ky ; 1: dgrad gather  iy = (oy + pad - ky)/stride
};

struct GemmNTArgs {
  const float* A;
  const float* Bw;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  ConvGeom g;
  const float* bias;
  const float* sub;    // per-column subtrahend (eval BatchNorm running mean)
  const float* scale;
  const float* var;    // optional: scale[n] is divided by sqrtf(var[n] + var_eps) -- eval BatchNorm's a = gamma / sqrt(running_var + eps)
  float var_eps;       // formed here exactly as bn_eval_affine_kernel forms it (one launch per layer and step fewer)
  const float* shift;
  const float* resid;
  int ldr;
  float* pre;
  int act;  // 0 none, 1 relu, 2 gelu
  float alpha;
  int vecA, vecB;  // 1: 16-byte aligned rows -> float4 loads
  // output row map: GEMM row m = (n, oy', ox') over the OHxOW grid is written to row
  // (n*oOH + oy'*os + opy)*oOW + ox'*os + opx of C (os == 1: identity).  Used by the phase-decomposed strided dgrad.
  int os, opy, opx, oOH, oOW;
  // optional per-tile column statistics of the stored output (training BatchNorm without a statistics pass over C):
  // colstats[(tm*2 + 0)*N + col] = mean over the tile's valid rows, [(tm*2 + 1)*N + col] = sum of squared deviations
  float* colstats;
  // optional training-BatchNorm + ReLU of the PRODUCING layer applied to the A operand as it is staged (the consumer conv
  // of a bottleneck reads the raw conv output y of the layer before it; z = relu((y - mean) * a + beta) is never
  // materialised): abn = that layer's statistics block [4][C] = mean, rstd, a = gamma * rstd, beta (bn.hip)
  const float* abn;
  const float* rmask;  // optional [rows, ldc]-shaped tensor (row stride ldm): the stored value is zeroed where rmask <= 0 -- the ReLU
  int ldm;             // mask of the block output whose gradient this launch produces (pre-masked residual gradient)
  int wide;   // 1: 16-byte epilogue (64x64 tiles; N, ldc, ldr % 4 == 0 and C / resid / pre / per-column vectors 16-byte aligned)
  // optional BatchNorm-BACKWARD statistics of the layer whose output gradient this launch produces (wide epilogue only): the
  // stored value g (after residual / mask) is the gradient w.r.t. z = [relu](BN(y)); per tile and column the epilogue leaves
  //   bstats[(t*2 + 0)*N + col] = sum_rows g',   bstats[(t*2 + 1)*N + col] = sum_rows g' * xhat,   xhat = (y - mean) * rstd
  // with g' = g (bs_relu 0: g already carries the ReLU mask, or the layer has no ReLU) or g * [(y - mean)*a + beta > 0]
  // (bs_relu 2: the mask recomputed from y exactly as the forward formed z) -- what bn_bwd_partial_kernel computes in a pass of
  // its own over g and y (models/resnets.py:112-132 backward).  bs_y: that layer's raw conv output, indexed like the output
  // (row stride ldbs); bs_stats: its statistics block [4][N] (mean, rstd, a, beta); t = bs_tile0 + this launch's row-tile index.
  const float* bs_y;
  const float* bs_stats;
  float* bstats;
  int ldbs, bs_relu, bs_tile0;
  // split-K (64x64 tiles; grids below one workgroup per CU): `splits` workgroups share an output tile, each walks its own range
  // of k-tiles and writes its raw accumulators to a slab; the one that draws the last arrival ticket adds the slabs in split
  // order (deterministic) and runs the epilogue.  Nobody waits.  tickets: one int per tile, zero before the first launch; the
  // last arriver leaves it zero.
  int splits;
  float* split_slabs;
  int* split_tickets;
  // tile order (speed only; every tile computes the same bits wherever it runs): 0 = row-tile major (all column tiles of a row tile
  // are consecutive, an XCD's contiguous range is a band of rows), n > 0 = column PANELS of n tiles, row-tile major inside a panel:
  // an XCD's range is then (rows of a band) x (n column tiles), whose slice of W (n x 64 x K floats) stays in that XCD's L2 while
  // the band's A row panels stream through once per panel.  Wide outputs with a W that does not fit the 4 MB L2 (N = 2048, K = 512:
  // 4.2 MB) re-fetched W for every group of row tiles in flight: 794 MB per launch against 38 MB algorithmic (round-5 PMC).
  int panel;
};

__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid, int vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // blocks b and b+8 share an XCD (speed only): give each XCD a contiguous range of logical tiles.
  int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// branch-free 16-byte load: the address is clamped to a valid one and the value selected afterwards, so the
// compiler can issue every load of a tile up front and wait once, right before the LDS store (a divergent
// branch per load made it wait vmcnt(0) inside the load block, serialising prefetch and MFMA).
__device__ __forceinline__ float4 ld4_sel(const float* p, const float* safe, bool ok) {
  const float4 v = *reinterpret_cast<const float4*>(ok ? p : safe);
  return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Epilogue shared by the tile kernels: optional per-tile column statistics of the raw product (training BatchNorm without
// a statistics pass), then alpha / eval-BN affine / bias / residual / activation and the (optionally re-mapped) store.
// acc: this wave's TM x TN 32x32 accumulators of output tile (tm, tn); red: >= 2 * 64 * TN floats of LDS scratch.
template <int TM, int TN, bool PLAIN>
__device__ __forceinline__ void gemm_nt_epilogue(const GemmNTArgs& p, f32x16 (&acc)[TM][TN], float* red, int tm, int tn, int wm, int wn,
                                                 int li, int lh) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const ConvGeom& g = p.g;
  // per-tile column mean / M2 (Welford partials, combined in fixed order by bn_stats_final_tiles_kernel)
  if (p.colstats) {
    // red: [2 (wm)][BN] floats of LDS scratch that no other wave is still reading
    const int rows_valid = min(BM, p.M - tm * BM);
    float mean_t[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row_m = tm * BM + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row_m < p.M) sacc += acc[i][j][r] * p.alpha;
        }
      sacc += __shfl_xor(sacc, 32, 64);
      if (lh == 0) red[wm * BN + wn * TN * 32 + j * 32 + li] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cs = wn * TN * 32 + j * 32 + li;
      mean_t[j] = (red[cs] + red[BN + cs]) / (float)rows_valid;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row_m = tm * BM + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float d = acc[i][j][r] * p.alpha - mean_t[j];
          if (row_m < p.M) q += d * d;
        }
      q += __shfl_xor(q, 32, 64);
      if (lh == 0) red[wm * BN + wn * TN * 32 + j * 32 + li] = q;
    }
    __syncthreads();
    if (wm == 0 && lh == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cs = wn * TN * 32 + j * 32 + li, col = tn * BN + cs;
        if (col < p.N) {
          p.colstats[((long)tm * 2) * p.N + col] = mean_t[j];
          p.colstats[((long)tm * 2 + 1) * p.N + col] = red[cs] + red[BN + cs];
        }
      }
    }
  }

  // Wide epilogue (64x64 tiles): the accumulators go through LDS once (the operand buffers are free by now) so that every
  // lane handles FOUR consecutive columns of a row: residual loads, `pre` stores and output stores are 16 bytes per lane
  // (a wave instruction covers 4 rows x 256 B) instead of one dword per lane (2 rows x 128 B) -- a quarter of the memory
  // instructions for the same bytes.  The short-K products, whose time is their epilogue's traffic, are bound by the
  // number of those instructions.  Same per-element arithmetic in the same order as the scalar path below: identical bits.
  if constexpr (TM == 1 && TN == 1) {
    if (p.wide) {
      constexpr int TS = BN + 4;            // 68 floats: rows 16-byte aligned, ds_read_b128 rows conflict-free
      float* T = red;                       // 64 x 68 floats = 17 KB <= the 20 KB of operand buffers
      __syncthreads();                      // the statistics scratch (same LDS) is done with
#pragma unroll
      for (int r = 0; r < 16; ++r)
        T[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TS + wn * 32 + li] = acc[0][0][r];
      __syncthreads();
      const int tid = threadIdx.x, cq = (tid & 15) * 4, col = tn * BN + cq;
      if (col < p.N) {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bia = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : zero4;
        const float4 sb = p.sub ? *reinterpret_cast<const float4*>(p.sub + col) : zero4;
        float4 sc = p.scale ? *reinterpret_cast<const float4*>(p.scale + col) : one4;
        if (p.var) {
          const float4 vv = *reinterpret_cast<const float4*>(p.var + col);
          sc.x = sc.x / sqrtf(vv.x + p.var_eps); sc.y = sc.y / sqrtf(vv.y + p.var_eps);
          sc.z = sc.z / sqrtf(vv.z + p.var_eps); sc.w = sc.w / sqrtf(vv.w + p.var_eps);
        }
        const float4 sh = p.shift ? *reinterpret_cast<const float4*>(p.shift + col) : zero4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int rl = ps * 16 + (tid >> 4), row_m = tm * BM + rl;
          if (row_m >= p.M) continue;
          long row = row_m;
          if (!PLAIN && p.os != 1) {
            const int ohw = g.OH * g.OW;
            const int n = row_m / ohw, rem = row_m - n * ohw;
            const int oy = rem / g.OW, ox = rem - oy * g.OW;
            row = ((long)n * p.oOH + oy * p.os + p.opy) * p.oOW + ox * p.os + p.opx;
          }
          float4 v = *reinterpret_cast<const float4*>(T + rl * TS + cq);
          v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
          v.x = (v.x - sb.x) * sc.x + sh.x + bia.x; v.y = (v.y - sb.y) * sc.y + sh.y + bia.y;
          v.z = (v.z - sb.z) * sc.z + sh.z + bia.z; v.w = (v.w - sb.w) * sc.w + sh.w + bia.w;
          if (p.resid) {
            const float4 rr = *reinterpret_cast<const float4*>(p.resid + row * p.ldr + col);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          if (p.pre) *reinterpret_cast<float4*>(p.pre + row * p.ldc + col) = v;
          if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (p.act == 2) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
          if (p.rmask) {
            const float4 mk = *reinterpret_cast<const float4*>(p.rmask + row * p.ldm + col);
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
          }
          *reinterpret_cast<float4*>(p.C + row * p.ldc + col) = v;
          if (p.bstats) *reinterpret_cast<float4*>(T + rl * TS + cq) = v;   // the stored gradient, for the statistics loop below (own slot)
        }
      }
      if (p.bstats) {   // block-uniform.  A loop of its own (the affine / bias registers are dead by now: no spills in the hot kernels)
        float4 bs1 = make_float4(0.f, 0.f, 0.f, 0.f), bs2 = bs1;   // this thread's 4 rows x 4 columns of (g', g' * xhat)
        if (col < p.N) {
          const float4 bmu = *reinterpret_cast<const float4*>(p.bs_stats + col);
          const float4 brs = *reinterpret_cast<const float4*>(p.bs_stats + p.N + col);
          const float4 bap = *reinterpret_cast<const float4*>(p.bs_stats + 2 * p.N + col);
          const float4 bbp = *reinterpret_cast<const float4*>(p.bs_stats + 3 * p.N + col);
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int rl = ps * 16 + (tid >> 4), row_m = tm * BM + rl;
            if (row_m >= p.M) continue;
            long row = row_m;
            if (!PLAIN && p.os != 1) {
              const int ohw = g.OH * g.OW;
              const int n = row_m / ohw, rem = row_m - n * ohw;
              const int oy = rem / g.OW, ox = rem - oy * g.OW;
              row = ((long)n * p.oOH + oy * p.os + p.opy) * p.oOW + ox * p.os + p.opx;
            }
            const float4 yy = *reinterpret_cast<const float4*>(p.bs_y + row * p.ldbs + col);
            float4 gg = *reinterpret_cast<const float4*>(T + rl * TS + cq);
            if (p.bs_relu == 2) {   // the sign of z = (y - mean)*a + beta, as bn_bwd_partial_kernel / the operand staging form it
              gg.x = ((yy.x - bmu.x) * bap.x + bbp.x) > 0.f ? gg.x : 0.f; gg.y = ((yy.y - bmu.y) * bap.y + bbp.y) > 0.f ? gg.y : 0.f;
              gg.z = ((yy.z - bmu.z) * bap.z + bbp.z) > 0.f ? gg.z : 0.f; gg.w = ((yy.w - bmu.w) * bap.w + bbp.w) > 0.f ? gg.w : 0.f;
            }
            bs1.x += gg.x; bs2.x += gg.x * ((yy.x - bmu.x) * brs.x);
            bs1.y += gg.y; bs2.y += gg.y * ((yy.y - bmu.y) * brs.y);
            bs1.z += gg.z; bs2.z += gg.z * ((yy.z - bmu.z) * brs.z);
            bs1.w += gg.w; bs2.w += gg.w * ((yy.w - bmu.w) * brs.w);
          }
        }
        // 16 row-threads per column quad -> LDS -> one fixed-order sum per column (deterministic)
        __syncthreads();                    // everybody is done reading T
        float* R1 = red;                    // [16][64] sums of g', then [16][64] sums of g' * xhat  (8 KB <= T's 17 KB)
        float* R2 = red + 16 * 64;
        const int rt = tid >> 4;
        *reinterpret_cast<float4*>(R1 + rt * 64 + cq) = bs1;
        *reinterpret_cast<float4*>(R2 + rt * 64 + cq) = bs2;
        __syncthreads();
        if (tid < 128) {
          const int c = tid & 63, which = tid >> 6, gc = tn * BN + c;
          const float* R = which ? R2 : R1;
          float t = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) t += R[r * 64 + c];
          if (gc < p.N) p.bstats[((long)(p.bs_tile0 + tm) * 2 + which) * p.N + gc] = t;
        }
      }
      return;
    }
  }

  // epilogue: C/D map of 32x32 tiles: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = tn * BN + wn * TN * 32 + j * 32 + li;
    if (col >= p.N) continue;
    const float bia = p.bias ? p.bias[col] : 0.f;
    const float sb = p.sub ? p.sub[col] : 0.f;
    float sc = p.scale ? p.scale[col] : 1.f;
    if (p.var) sc = sc / sqrtf(p.var[col] + p.var_eps);
    const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row_m = tm * BM + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row_m >= p.M) continue;
        long row = row_m;
        if (!PLAIN && p.os != 1) {
          const int ohw = g.OH * g.OW;
          const int n = row_m / ohw, rem = row_m - n * ohw;
          const int oy = rem / g.OW, ox = rem - oy * g.OW;
          row = ((long)n * p.oOH + oy * p.os + p.opy) * p.oOW + ox * p.os + p.opx;
        }
        float v = acc[i][j][r] * p.alpha;
        v = (v - sb) * sc + sh + bia;
        if (p.resid) v += p.resid[row * p.ldr + col];
        if (p.pre) p.pre[row * p.ldc + col] = v;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = gelu_f(v);
        if (p.rmask) v = p.rmask[row * p.ldm + col] > 0.f ? v : 0.f;
        p.C[row * p.ldc + col] = v;
      }
    }
  }
}

// ACC2: two-level accumulation.  v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain, so a K-long reduction done in ONE
// accumulator carries ~sqrt(K) eps of rounding noise -- measured 5x ATen-CPU's error at K = 4608 (oneDNN keeps many
// partial sums).  With ACC2 the MFMAs of FLUSH consecutive k-tiles (64 products) chain into a partial accumulator that
// is then added to the master one: noise ~ (8 + sqrt(K/64)) eps, at the cost of 16 VGPRs per 32x32 tile and 16 v_add
// per 32 MFMAs.  Used for K >= 512 (shorter reductions gain nothing).
// PLAIN: the A operand is a plain row-major matrix (1x1 convolution with stride 1 / no padding, nn.Linear) and the output
// map is the identity: the (n, oy, ox) row decode, the tap walk and the output re-map -- ~500 integer instructions per
// workgroup, as many as the MFMAs of a K = 64 tile take -- are compiled out.
// amdgpu_waves_per_eu: a register budget for the instantiations the step spends its time in -- 64x64 tiles at 6 waves per SIMD
// (the conv-gather two-level kernel would otherwise take 88 registers = 5 waves), plain 128x64 two-level at 4 (120 instead of
// 140 registers); none of them spills (checked in the ISA: private_segment_fixed_size 0).
// BNA: the A operand is the raw output of a training conv+BN+ReLU layer; its BatchNorm + ReLU are applied between the
// global load and the LDS store (per-channel mean / a / beta from a 3 x C table staged in LDS once per workgroup; the
// zero padding of a conv gather stays zero).  Same arithmetic as bn_apply_kernel, so the products are those of the
// materialised path bit for bit.
// NB: number of LDS operand buffers.  2 = the next k-tile is stored while the current one is read (one barrier per k-tile);
// 1 = one buffer, two barriers per k-tile, half the LDS: a 32-deep tile then costs what two 16-deep buffers cost and as
// many workgroups fit a CU (the barriers are free: other workgroups' MFMAs fill them).
// B3: the OPT-IN split-precision mode (`precision bf16x3`; never the bench line's `value`): every staged fp32 operand element is
// split into three bf16 terms as it goes to LDS (three planes per operand, rows of 32 bf16 + 8 pad) and a 32-deep k-tile is
// 2 x 6 v_mfma_f32_32x32x16_bf16 (the six term pairs of weight >= 2^-16, smallest first, fp32 accumulate) instead of 16
// v_mfma_f32_32x32x2_f32: 384 matrix-pipe cycles instead of 1024.  With the two-level sums (always on here) its distance from a
// float64 product is at or below the fp32-exact chain's on every shape of the step (profiles/r05_bf16x3_lab.txt).  Same C/D
// layout as the fp32 MFMA: gathers, split-K and every epilogue are shared.
template <int TM, int TN, int BK, bool VEC, bool ACC2 = false, bool PLAIN = false, bool BNA = false, int NB = 2, bool B3 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(B3 ? (PLAIN ? 4 : 3) : (((BK == 16 || NB == 1) && VEC) ? (TM * TN == 1 ? ((NB == 1 && ACC2) ? 4 : ((BNA && ACC2 && !PLAIN) ? 5 : 6)) : ((TM * TN == 2 && ACC2 && (PLAIN || TN == 2)) ? 4 : 1)) : 1))))
void gemm_nt_kernel(GemmNTArgs p) {
  static_assert(!B3 || (BK == 32 && NB == 1 && VEC && ACC2), "the bf16x3 mode runs 32-deep k-tiles in one LDS buffer with two-level sums");
  constexpr int BM = 64 * TM, BN = 64 * TN, LS = BK + 4;   // LS = 20 / 36: conflict-free ds_read_b128 (i*LS mod 64 distinct)
  constexpr int KQ = BK / 4, RP = 256 / KQ;                 // float4 per tile row, tile rows per load pass
  constexpr int RA = BM / RP, RB = BN / RP;
  constexpr int LSB = 40;                                   // B3: bf16 per LDS row (64 B of data + 16 B pad: conflict-free ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * (BM + BN) * LS floats;  B3: 3 * (BM + BN) * LSB bf16
  float* As = lds;
  float* Bs = lds + NB * BM * LS;
  float* tab = B3 ? lds + 3 * (BM + BN) * LSB / 2 : lds + NB * (BM + BN) * LS;   // BNA: [3][C] = mean, a, beta of the producing layer
  __bf16* Ap3 = reinterpret_cast<__bf16*>(lds);             // B3: A planes [3][BM][LSB], then W planes [3][BN][LSB]
  __bf16* Bp3 = Ap3 + 3 * BM * LSB;

  const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  int wg = xcd_remap(blockIdx.x, nbm * nbn * (TM * TN == 1 ? p.splits : 1));
  int split = 0;
  if constexpr (TM * TN == 1) {   // a tile's slices are consecutive: same XCD (speed only)
    if (p.splits > 1) { split = wg % p.splits; wg /= p.splits; }
  }
  int tm = wg / nbn, tn = wg - tm * nbn;
  if (p.panel > 0) {   // block-uniform; nbn % panel == 0 (the launcher's condition)
    const int per_panel = nbm * p.panel;
    const int pn = wg / per_panel, rem = wg - pn * per_panel;
    tm = rem / p.panel;
    tn = pn * p.panel + (rem - tm * p.panel);
  }
  const int tid = threadIdx.x, kq = tid % KQ, r0 = tid / KQ;
  const ConvGeom g = p.g;

  int a_n[RA], a_oy[RA], a_ox[RA];
  bool a_ok[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = tm * BM + r0 + RP * i;
    a_ok[i] = m < p.M;
    if constexpr (PLAIN) {
      a_n[i] = a_ok[i] ? m : 0; a_oy[i] = 0; a_ox[i] = 0;   // a_n doubles as the row index
    } else {
      int mm = a_ok[i] ? m : 0;
      int ohw = g.OH * g.OW;
      int n = mm / ohw, rem = mm - n * ohw;
      int oy = rem / g.OW;
      a_n[i] = n; a_oy[i] = oy; a_ox[i] = rem - oy * g.OW;
    }
  }
  const float* b_ptr[RB];
  bool b_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = tn * BN + r0 + RP * i;
    b_ok[i] = n < p.N;
    b_ptr[i] = p.Bw + (long)(b_ok[i] ? n : 0) * p.ldb + kq * 4;
  }

  // K is walked tap by tap: the per-row source pointer and validity are recomputed only when the tap changes
  // (every C/BK iterations); inside a tap an iteration costs one pointer add per load.
  float4 ra[RA], rb[RB];
  const float* a_src[RA];
  bool a_val[RA];
  bool ra_ok[RA];   // BNA: which of the staged A quads are real data (padding / out-of-range rows must stay zero)
  int ra_c = 0;     // BNA: channel of the staged quad's first element
  if constexpr (BNA) {
    const int C = p.g.C;
    for (int i = threadIdx.x; i < 3 * C; i += 256) {
      const int r = i / C, c = i - r * C;
      tab[i] = p.abn[(r == 0 ? 0 : r + 1) * C + c];
    }
    __syncthreads();
  }
  auto set_tap = [&](int tap) {
    if constexpr (PLAIN) {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        a_val[i] = a_ok[i];
        a_src[i] = p.A + (long)a_n[i] * p.lda + kq * 4;
      }
      return;
    }
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy, ix;
      bool ok = a_ok[i];
      if (g.mode == 0) {
        iy = a_oy[i] * g.stride - g.pad_y + ky;
        ix = a_ox[i] * g.stride - g.pad_x + kx;
      } else {
        int ty = a_oy[i] + g.pad_y - ky, tx = a_ox[i] + g.pad_x - kx;
        ok = ok && ty >= 0 && tx >= 0;
        iy = ty / g.stride; ix = tx / g.stride;
        ok = ok && (iy * g.stride == ty) && (ix * g.stride == tx);
      }
      ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
      a_val[i] = ok;
      a_src[i] = ok ? p.A + ((long)(a_n[i] * g.H + iy) * g.W + ix) * p.lda + kq * 4 : p.A;
    }
  };
  // this workgroup's range of k-tiles [kt0, kt0 + nk): everything unless the tile is split over K
  const int per_tap = (g.C + BK - 1) / BK;
  const int nk_all = (PLAIN ? 1 : g.KH * g.KW) * per_tap;
  int kt0 = 0, nk = nk_all;
  if constexpr (TM * TN == 1) {
    if (p.splits > 1) {
      int kpt = (nk_all + p.splits - 1) / p.splits;
      kpt = ((kpt + 64 / BK - 1) / (64 / BK)) * (64 / BK);   // whole partial chains of the two-level accumulation
      kt0 = min(nk_all, split * kpt);
      nk = min(nk_all, kt0 + kpt) - kt0;
    }
  }
  int cur_tap = kt0 / per_tap, c0 = (kt0 - cur_tap * per_tap) * BK, kb = cur_tap * g.C + c0;  // c0: channel offset inside the tap; kb = cur_tap*C + c0 (column of W)
  const int ntaps = PLAIN ? 1 : g.KH * g.KW;
  set_tap(cur_tap < ntaps ? cur_tap : 0);
  auto gload = [&]() {
    const int nval = g.C - (c0 + kq * 4);   // <= 0 only in the K tail of a plain GEMM (C == K, K % BK != 0)
    if constexpr (BNA) ra_c = c0 + kq * 4;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      if (VEC) ra[i] = ld4_sel(a_src[i] + c0, p.A, a_val[i] && nval > 0);
      else ra[i] = a_val[i] ? ld4_guard(a_src[i] + c0, nval, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (BNA) ra_ok[i] = a_val[i] && nval > 0;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (VEC) rb[i] = ld4_sel(b_ptr[i] + kb, p.Bw, b_ok[i] && nval > 0);
      else rb[i] = b_ok[i] ? ld4_guard(b_ptr[i] + kb, nval, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    c0 += BK; kb += BK;
    if (c0 >= g.C) {   // block-uniform
      kb += g.C - c0;  // (plain GEMM tail: keep kb = tap*C)
      c0 = 0;
      if (++cur_tap < ntaps) set_tap(cur_tap);
    }
  };
  auto lstore = [&](int buf) {
    if constexpr (BNA) {   // z = relu((y - mean) * a + beta), exactly as bn_apply_kernel computes it
      const int C = p.g.C;
      const float4 mu = *reinterpret_cast<const float4*>(tab + ra_c);
      const float4 aa = *reinterpret_cast<const float4*>(tab + C + ra_c);
      const float4 bb = *reinterpret_cast<const float4*>(tab + 2 * C + ra_c);
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        float4 v = ra[i];
        v.x = (v.x - mu.x) * aa.x + bb.x; v.y = (v.y - mu.y) * aa.y + bb.y; v.z = (v.z - mu.z) * aa.z + bb.z; v.w = (v.w - mu.w) * aa.w + bb.w;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        ra[i] = ra_ok[i] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if constexpr (B3) {
#pragma unroll
      for (int i = 0; i < RA + RB; ++i) {
        const float4 v = i < RA ? ra[i < RA ? i : 0] : rb[i < RA ? 0 : i - RA];
        const float e[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h1, h2, h3;
#pragma unroll
        for (int q = 0; q < 4; ++q) { __bf16 a1, a2, a3; split3(e[q], a1, a2, a3); h1[q] = a1; h2[q] = a2; h3[q] = a3; }
        __bf16* dst = (i < RA ? Ap3 + (r0 + RP * i) * LSB : Bp3 + (r0 + RP * (i - RA)) * LSB) + kq * 4;
        constexpr int PLA = BM * LSB, PLB = BN * LSB;
        const int PL = i < RA ? PLA : PLB;
        *reinterpret_cast<bf16x4*>(dst) = h1; *reinterpret_cast<bf16x4*>(dst + PL) = h2; *reinterpret_cast<bf16x4*>(dst + 2 * PL) = h3;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *reinterpret_cast<float4*>(&As[buf * BM * LS + (r0 + RP * i) * LS + kq * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i)
      *reinterpret_cast<float4*>(&Bs[buf * BN * LS + (r0 + RP * i) * LS + kq * 4]) = rb[i];
  };

  const int w = tid >> 6, wm = w >> 1, wn = w & 1, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int FLUSH = 64 / BK;   // k-tiles per partial chain (ACC2)

  // Interior fast path (block-uniform): single-tap gathers (plain GEMM, 1x1 conv) whose tile lies fully inside M x N
  // and whose K is a multiple of BK need no predicate at all -> unconditional 16-byte loads, pointer += BK.
  const bool full = VEC && ntaps == 1 && (PLAIN || g.mode == 0) && (g.C % BK == 0) && (tm + 1) * BM <= p.M && (tn + 1) * BN <= p.N;
  auto mainloop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    auto load = [&]() {
      if constexpr (FULL) {
        if constexpr (BNA) {
          ra_c = c0 + kq * 4;
#pragma unroll
          for (int i = 0; i < RA; ++i) ra_ok[i] = true;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] = *reinterpret_cast<const float4*>(a_src[i] + c0);
#pragma unroll
        for (int i = 0; i < RB; ++i) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + c0);
        c0 += BK;
      } else {
        gload();
      }
    };
    // the MFMAs of one k-tile into c; `fresh`: c starts from 0 (inline constant, no zeroed registers)
    auto compute = [&](int buf, f32x16 (&c)[TM][TN], auto fresh_tag) {
      constexpr bool FRESH = decltype(fresh_tag)::value;
      if constexpr (B3) {
        const __bf16* Ab3 = Ap3 + (wm * TM * 32 + li) * LSB + lh * 8;
        const __bf16* Bb3 = Bp3 + (wn * TN * 32 + li) * LSB + lh * 8;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          bf16x8 a[3][TM], b[3][TN];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(Ab3 + pl * BM * LSB + i * 32 * LSB + s_ * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(Bb3 + pl * BN * LSB + j * 32 * LSB + s_ * 16);
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {   // smallest terms first
              if (FRESH && s_ == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], zero, 0, 0, 0);
              } else {
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c[i][j], 0, 0, 0);
              }
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c[i][j], 0, 0, 0);
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c[i][j], 0, 0, 0);
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c[i][j], 0, 0, 0);
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c[i][j], 0, 0, 0);
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c[i][j], 0, 0, 0);
            }
          __builtin_amdgcn_s_setprio(0);
        }
        return;
      }
      const float* Ab = As + buf * BM * LS + (wm * TM * 32 + li) * LS + lh * 4;
      const float* Bb = Bs + buf * BN * LS + (wn * TN * 32 + li) * LS + lh * 4;
#pragma unroll
      for (int t = 0; t < BK / 8; ++t) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LS + t * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LS + t * 8);
        __builtin_amdgcn_s_setprio(1);  // +1-2 % (measured): keeps the MFMA cluster ahead of co-resident waves' VALU/loads
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (FRESH && t == 0) {
              const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, zero, 0, 0, 0);
            } else {
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, c[i][j], 0, 0, 0);
            }
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, c[i][j], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
      }
    };
    if (nk <= 0) return;   // an empty slice of a split tile: zero accumulators (block-uniform)
    load();
    lstore(0);
    __syncthreads();
    if constexpr (!ACC2) {
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & (NB - 1);
        if (kt + 1 < nk) load();
        compute(buf, acc, std::false_type{});
        if constexpr (NB == 1) __syncthreads();   // everybody has read the only buffer
        if (kt + 1 < nk) lstore((kt + 1) & (NB - 1));
        __syncthreads();
      }
    } else {
      int kt = 0;
      for (; kt + FLUSH <= nk; kt += FLUSH) {   // FLUSH is even: the LDS buffer of tile kt + u is u & 1
        f32x16 part[TM][TN];
#pragma unroll
        for (int u = 0; u < FLUSH; ++u) {
          if (kt + u + 1 < nk) load();
          if (u == 0) compute(0, part, std::true_type{}); else compute(u & (NB - 1), part, std::false_type{});
          if constexpr (NB == 1) __syncthreads();
          if (kt + u + 1 < nk) lstore((u + 1) & (NB - 1));
          __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] += part[i][j];
      }
      if (kt < nk) {   // remainder (< FLUSH tiles): one more partial chain
        f32x16 part[TM][TN];
        const int kt0 = kt;
        for (; kt < nk; ++kt) {
          const int buf = kt & (NB - 1);
          if (kt + 1 < nk) load();
          if (kt == kt0) compute(buf, part, std::true_type{}); else compute(buf, part, std::false_type{});
          if constexpr (NB == 1) __syncthreads();
          if (kt + 1 < nk) lstore((kt + 1) & (NB - 1));
          __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] += part[i][j];
      }
    }
  };
  if (full) mainloop(std::true_type{}); else mainloop(std::false_type{});

  if constexpr (TM * TN == 1) {
    if (p.splits > 1) {
      // The hand-off of cdna_hip_programming.md (split-K, counter form) with WRITE-THROUGH slabs: every slice stores its accumulators
      // with 16-byte sc1 stores (they leave the XCD's L2: no release fence, which would write back every dirty line of that L2 --
      // the other workgroups' output tiles -- once per slice: measured 40 us on a 45 us product of 784 tiles cut in two) -> every
      // wave drains its stores -> workgroup barrier -> ONE lane draws a relaxed agent-scope ticket.  The workgroup that draws the
      // last ticket reads all slabs with sc1 loads (every load of them: they bypass this CU's L1, so no acquire either).
      const long tile = wg;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const float* tbase = p.split_slabs + tile * p.splits * 4096;    // block-uniform
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tbase);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)tbase >> 32));
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, p.splits * 16384, 0x00020000);
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // lane tid holds floats [q][tid][0..3] of its slice's slab
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[0][0][q * 4 + e]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (split * 4096 + q * 1024 + tid * 4) * 4, 0, 16);   // aux 16 = sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = reinterpret_cast<int*>(lds);    // the operand buffers are free; the ONE LDS array of the kernel
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.split_tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == p.splits - 1;
        if (last) __hip_atomic_store(p.split_tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody has arrived: ready for the next launch
        flag[0] = last;
      }
      __syncthreads();
      const int last = flag[0];
      __syncthreads();          // the flag word is operand / epilogue scratch again
      if (!last) return;        // block-uniform
      // slice order, in DOUBLE, rounded to fp32 once: the same sum on every run, and each slice's partial enters exactly (the
      // split product carries less rounding noise than the unsplit chain of K / 64 partial sums it replaces)
      double dsum[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dsum[r] = 0.0;
      for (int s_ = 0; s_ < p.splits; ++s_) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (s_ * 4096 + q * 1024 + tid * 4) * 4, 0, 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) dsum[q * 4 + e] += (double)__uint_as_float(v[e]);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] = (float)dsum[r];
    }
  }

  gemm_nt_epilogue<TM, TN, PLAIN>(p, acc, lds, tm, tn, wm, wn, li, lh);
}

// ---------------------------------------------------------------------------------------
// gemm_nt_big_kernel: the 128x128 block tile for PLAIN products (1x1 / stride-1 convolutions, nn.Linear) as a LEAN kernel of its
// own -- 256 threads = 2x2 waves x (2x2 tiles of 32x32), 32-deep k-tiles in one LDS buffer, interior tiles only (M, N % 128 == 0,
// K % 32 == 0, 16-byte aligned operands), ~220 registers = two waves per SIMD.  Why: two co-running launches of 128x128 tiles
// reach 118-125 TF/s in aggregate where two of 64x64 tiles reach 109-117 (tests/tools/gemm_lab.hip `pair`, profiles/
// r05_gemm_lab_pairs.txt) -- the other stream fills the big tile's tail and its steady state is better -- but the generic
// kernel's 2x2 instantiation takes 212 + 64 registers (ONE wave per SIMD) and lost on the step for that reason.
// Same k order and two-level sums as gemm_nt_kernel: the products are bit-identical to the 64x64 path's.  The per-tile statistics
// keep their 64-row granularity (a wave row of the block IS one 64-row statistics tile: its sums need no LDS and no barrier; the
// association of those sums differs from the 64x64 kernel's, the statistics agree to rounding).  Epilogue: every lane handles four
// consecutive columns of a row (16-byte accesses) in four rounds, one 32x32 accumulator of each wave per round.
template <bool ACC2, bool BNA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_big_kernel(GemmNTArgs p) {
  constexpr int BM = 128, BN = 128, BK = 32, LS = BK + 4, KQ = BK / 4, RP = 256 / KQ, RA = BM / RP, RB = BN / RP, TS = 68;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;
  float* Bs = lds + BM * LS;
  float* tab = lds + (BM + BN) * LS;   // BNA: [3][C] = mean, a, beta of the producing layer
  const int nbn = p.N / BN, nbm = p.M / BM;
  const int wg = xcd_remap(blockIdx.x, nbm * nbn);
  int tm = wg / nbn, tn = wg - tm * nbn;
  if (p.panel > 0) {   // column panels of p.panel 128-column tiles (see GemmNTArgs::panel)
    const int per_panel = nbm * p.panel;
    const int pn = wg / per_panel, rem = wg - pn * per_panel;
    tm = rem / p.panel;
    tn = pn * p.panel + (rem - tm * p.panel);
  }
  const int tid = threadIdx.x, kq = tid % KQ, r0 = tid / KQ;
  const int w = tid >> 6, wm = w >> 1, wn = w & 1, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* a_src[RA];
  const float* b_src[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) a_src[i] = p.A + (long)(tm * BM + r0 + RP * i) * p.lda + kq * 4;
#pragma unroll
  for (int i = 0; i < RB; ++i) b_src[i] = p.Bw + (long)(tn * BN + r0 + RP * i) * p.ldb + kq * 4;
  if constexpr (BNA) {
    const int C = p.g.C;
    for (int i = tid; i < 3 * C; i += 256) {
      const int r = i / C, c = i - r * C;
      tab[i] = p.abn[(r == 0 ? 0 : r + 1) * C + c];
    }
    __syncthreads();
  }
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v ra[RA], rb[RB];
  int koff = 0, ra_c = 0;
  auto load = [&]() __attribute__((always_inline)) {
    if constexpr (BNA) ra_c = koff + kq * 4;
#pragma unroll
    for (int i = 0; i < RA; ++i) ra[i] = *reinterpret_cast<const f32x4v*>(a_src[i] + koff);
#pragma unroll
    for (int i = 0; i < RB; ++i) rb[i] = *reinterpret_cast<const f32x4v*>(b_src[i] + koff);
    koff += BK;
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    if constexpr (BNA) {   // z = relu((y - mean) * a + beta), exactly as bn_apply_kernel / gemm_nt_kernel compute it
      const int C = p.g.C;
      const f32x4v mu = *reinterpret_cast<const f32x4v*>(tab + ra_c);
      const f32x4v aa = *reinterpret_cast<const f32x4v*>(tab + C + ra_c);
      const f32x4v bb = *reinterpret_cast<const f32x4v*>(tab + 2 * C + ra_c);
#pragma unroll
      for (int i = 0; i < RA; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[i][e] = fmaxf((ra[i][e] - mu[e]) * aa[e] + bb[e], 0.f);
    }
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4v*>(&As[(r0 + RP * i) * LS + kq * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4v*>(&Bs[(r0 + RP * i) * LS + kq * 4]) = rb[i];
  };
  auto compute = [&](f32x16 (&c)[2][2], auto fresh_tag) __attribute__((always_inline)) {
    constexpr bool FRESH = decltype(fresh_tag)::value;
    const float* Ab = As + (wm * 64 + li) * LS + lh * 4;
    const float* Bb = Bs + (wn * 64 + li) * LS + lh * 4;
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
      f32x4v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4v*>(Ab + i * 32 * LS + t * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4v*>(Bb + j * 32 * LS + t * 8);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (FRESH && t == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][0], b[j][0], zero, 0, 0, 0);
          } else {
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][0], b[j][0], c[i][j], 0, 0, 0);
          }
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][1], b[j][1], c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][2], b[j][2], c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][3], b[j][3], c[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = p.K / BK;
  load();
  lstore();
  __syncthreads();
  if constexpr (!ACC2) {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load();
      compute(acc, std::false_type{});
      __syncthreads();
      if (kt + 1 < nk) lstore();
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; kt += 2) {   // chains of 64 products (two k-tiles) into `part`, then into the master: gemm_nt_kernel's order
      f32x16 part[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (kt + u < nk) {
          if (kt + u + 1 < nk) load();
          if (u == 0) compute(part, std::true_type{}); else compute(part, std::false_type{});
          __syncthreads();
          if (kt + u + 1 < nk) lstore();
          __syncthreads();
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] += part[i][j];
    }
  }

  // ---- per-tile column statistics of the raw product (training BatchNorm): wave row wm IS the 64-row statistics tile 2 tm + wm
  if (p.colstats) {
    const int ts = tm * 2 + wm;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r] * p.alpha;
      sacc += __shfl_xor(sacc, 32, 64);
      const float mean = sacc / 64.f;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] * p.alpha - mean; q += d * d; }
      q += __shfl_xor(q, 32, 64);
      if (lh == 0) {
        const int col = tn * BN + wn * 64 + j * 32 + li;
        p.colstats[((long)ts * 2) * p.N + col] = mean;
        p.colstats[((long)ts * 2 + 1) * p.N + col] = q;
      }
    }
  }

  // ---- epilogue in four rounds: round (i, j) gathers acc[i][j] of the four waves into a virtual 64x64 tile in LDS
  float* T = lds;
  const int cq = (tid & 15) * 4, rt = tid >> 4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = tn * BN + (cq >> 5) * 64 + j * 32 + (cq & 31);
    const float4 bia = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : zero4;
    const float4 sb = p.sub ? *reinterpret_cast<const float4*>(p.sub + col) : zero4;
    float4 sc = p.scale ? *reinterpret_cast<const float4*>(p.scale + col) : one4;
    if (p.var) {
      const float4 vv = *reinterpret_cast<const float4*>(p.var + col);
      sc.x = sc.x / sqrtf(vv.x + p.var_eps); sc.y = sc.y / sqrtf(vv.y + p.var_eps);
      sc.z = sc.z / sqrtf(vv.z + p.var_eps); sc.w = sc.w / sqrtf(vv.w + p.var_eps);
    }
    const float4 sh = p.shift ? *reinterpret_cast<const float4*>(p.shift + col) : zero4;
    float4 bs1[2], bs2[2];   // BatchNorm-backward sums of this thread's rows, per 64-row statistics tile (half 0: rows < 64 of the block)
    bs1[0] = bs1[1] = bs2[0] = bs2[1] = zero4;
    float4 bmu = zero4, brs = zero4, bap = zero4, bbp = zero4;
    if (p.bstats) {
      bmu = *reinterpret_cast<const float4*>(p.bs_stats + col);
      brs = *reinterpret_cast<const float4*>(p.bs_stats + p.N + col);
      bap = *reinterpret_cast<const float4*>(p.bs_stats + 2 * p.N + col);
      bbp = *reinterpret_cast<const float4*>(p.bs_stats + 3 * p.N + col);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) T[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TS + wn * 32 + li] = acc[i][j][r];
      __syncthreads();
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int rl = ps * 16 + rt;
        const long row = (long)tm * BM + (rl >> 5) * 64 + i * 32 + (rl & 31);
        float4 v = *reinterpret_cast<const float4*>(T + rl * TS + cq);
        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
        v.x = (v.x - sb.x) * sc.x + sh.x + bia.x; v.y = (v.y - sb.y) * sc.y + sh.y + bia.y;
        v.z = (v.z - sb.z) * sc.z + sh.z + bia.z; v.w = (v.w - sb.w) * sc.w + sh.w + bia.w;
        if (p.resid) {
          const float4 rr = *reinterpret_cast<const float4*>(p.resid + row * p.ldr + col);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.pre) *reinterpret_cast<float4*>(p.pre + row * p.ldc + col) = v;
        if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (p.act == 2) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
        if (p.rmask) {
          const float4 mk = *reinterpret_cast<const float4*>(p.rmask + row * p.ldm + col);
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<float4*>(p.C + row * p.ldc + col) = v;
        if (p.bstats) {   // block-uniform
          const float4 yy = *reinterpret_cast<const float4*>(p.bs_y + row * p.ldbs + col);
          float4 gg = v;
          if (p.bs_relu == 2) {
            gg.x = ((yy.x - bmu.x) * bap.x + bbp.x) > 0.f ? gg.x : 0.f; gg.y = ((yy.y - bmu.y) * bap.y + bbp.y) > 0.f ? gg.y : 0.f;
            gg.z = ((yy.z - bmu.z) * bap.z + bbp.z) > 0.f ? gg.z : 0.f; gg.w = ((yy.w - bmu.w) * bap.w + bbp.w) > 0.f ? gg.w : 0.f;
          }
          const int hf = ps >> 1;   // rl < 32: statistics tile 2 tm, else 2 tm + 1 (compile-time after unrolling)
          bs1[hf].x += gg.x; bs2[hf].x += gg.x * ((yy.x - bmu.x) * brs.x);
          bs1[hf].y += gg.y; bs2[hf].y += gg.y * ((yy.y - bmu.y) * brs.y);
          bs1[hf].z += gg.z; bs2[hf].z += gg.z * ((yy.z - bmu.z) * brs.z);
          bs1[hf].w += gg.w; bs2[hf].w += gg.w * ((yy.w - bmu.w) * brs.w);
        }
      }
    }
    if (p.bstats) {   // 16 row-threads per column quad -> LDS -> one fixed-order sum per column, statistics tile and sum (deterministic)
      __syncthreads();
      float* R = lds;   // [4 = (half, which)][16][64]
      *reinterpret_cast<float4*>(R + (0 * 16 + rt) * 64 + cq) = bs1[0];
      *reinterpret_cast<float4*>(R + (1 * 16 + rt) * 64 + cq) = bs2[0];
      *reinterpret_cast<float4*>(R + (2 * 16 + rt) * 64 + cq) = bs1[1];
      *reinterpret_cast<float4*>(R + (3 * 16 + rt) * 64 + cq) = bs2[1];
      __syncthreads();
      const int c = tid & 63, q4 = tid >> 6, hf = q4 >> 1, which = q4 & 1;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += R[(q4 * 16 + r) * 64 + c];
      const int gc = tn * BN + (c >> 5) * 64 + j * 32 + (c & 31);
      p.bstats[((long)(p.bs_tile0 + tm * 2 + hf) * 2 + which) * p.N + gc] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------
struct GemmTNArgs {
  const float* Y;  // [M, N], row stride ldy
  const float* X;  // gather source (NHWC, row stride ldx) -- geometry g, mode 0
  float* P;        // [splits][N][K]
  int M, N, K;
  int ldy, ldx;
  ConvGeom g;
  int m_per_split;
  int vecY, vecX;
  const float* xbn;   // optional: X is the raw output y of a training conv+BN+ReLU layer; relu((y - mean) * a + beta) is applied
                      // while X is staged (statistics block [4][C] of that layer), see GemmNTArgs::abn
};

template <int TNn, int TK, bool VEC, bool XBN = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs p) {
  constexpr int BN = 64 * TNn, BKo = 64 * TK, BMr = 16;
  constexpr int Y4 = BN / 4, X4 = BKo / 4;            // float4 per row
  constexpr int YR = 256 / Y4, XR = 256 / X4;         // rows per pass
  constexpr int YP = BMr / YR, XP = BMr / XR;         // passes
  __shared__ __attribute__((aligned(16))) float lds[2 * BMr * (BN + BKo)];
  float* Ys = lds;
  float* Xs = lds + 2 * BMr * BN;

  const int nbk = (p.K + BKo - 1) / BKo, nbn = (p.N + BN - 1) / BN;
  // 1-D grid over (m-slab z, tile): an XCD gets a contiguous range, so the tiles that re-read one slab of dY / X share an L2
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int z = lin / (nbn * nbk), wg = lin - z * (nbn * nbk);
  const int tn = wg / nbk, tk = wg - tn * nbk;
  const int m_begin = z * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int tid = threadIdx.x;
  const ConvGeom g = p.g;

  const int yc = tid % Y4, yr = tid / Y4;
  const int xc = tid % X4, xr = tid / X4;
  const int ycol = tn * BN + yc * 4;
  const int xk = tk * BKo + xc * 4;           // this thread's k (fixed over the loop)
  const int xtap = xk / g.C, xcch = xk - xtap * g.C;
  const int xky = xtap / g.KW, xkx = xtap - xky * g.KW;
  const int ohw = g.OH * g.OW;

  float4 ry[YP], rx[XP];
  bool rx_ok[XP];
  float4 xmu, xa, xb;   // XBN: this thread's four channels never change (its k column is fixed over the m loop)
  if constexpr (XBN) {
    const int cc = xk < p.K ? xcch : 0;
    xmu = *reinterpret_cast<const float4*>(p.xbn + cc);
    xa = *reinterpret_cast<const float4*>(p.xbn + 2 * g.C + cc);
    xb = *reinterpret_cast<const float4*>(p.xbn + 3 * g.C + cc);
  }
  // row decode (m -> n, oy, ox) is done once and then advanced incrementally by BMr rows per iteration
  const bool plain = (g.KH * g.KW == 1) && g.stride == 1 && g.pad_y == 0 && g.pad_x == 0 && g.OH == g.H && g.OW == g.W;
  int x_n[XP], x_oy[XP], x_ox[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int m = m_begin + xr + XR * i;
    int n = m / ohw, rem = m - n * ohw;
    x_n[i] = n; x_oy[i] = rem / g.OW; x_ox[i] = rem - x_oy[i] * g.OW;
  }
  auto gload = [&](int mt) {
    const int m0 = m_begin + mt * BMr;
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      int m = m0 + yr + YR * i;
      if (VEC) ry[i] = ld4_sel(p.Y + (long)m * p.ldy + ycol, p.Y, m < m_end && ycol < p.N);
      else ry[i] = (m < m_end) ? ld4_guard(p.Y + (long)m * p.ldy + ycol, p.N - ycol, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int m = m0 + xr + XR * i;
      bool ok = m < m_end && xk < p.K;
      const float* src;
      if (plain) {
        src = p.X + (long)m * p.ldx + xcch;
      } else {
        int iy = x_oy[i] * g.stride - g.pad_y + xky, ix = x_ox[i] * g.stride - g.pad_x + xkx;
        ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        src = p.X + ((long)(x_n[i] * g.H + iy) * g.W + ix) * p.ldx + xcch;
        x_ox[i] += BMr;
        while (x_ox[i] >= g.OW) { x_ox[i] -= g.OW; if (++x_oy[i] == g.OH) { x_oy[i] = 0; ++x_n[i]; } }
      }
      if (VEC) rx[i] = ld4_sel(src, p.X, ok);
      else rx[i] = ok ? ld4_guard(src, p.K - xk, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (XBN) rx_ok[i] = ok;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < YP; ++i)
      *reinterpret_cast<float4*>(&Ys[buf * BMr * BN + (yr + YR * i) * BN + yc * 4]) = ry[i];
    if constexpr (XBN) {   // z = relu((y - mean) * a + beta) as bn_apply_kernel computes it; padding stays zero
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        float4 v = rx[i];
        v.x = (v.x - xmu.x) * xa.x + xb.x; v.y = (v.y - xmu.y) * xa.y + xb.y; v.z = (v.z - xmu.z) * xa.z + xb.z; v.w = (v.w - xmu.w) * xa.w + xb.w;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        rx[i] = rx_ok[i] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < XP; ++i)
      *reinterpret_cast<float4*>(&Xs[buf * BMr * BKo + (xr + XR * i) * BKo + xc * 4]) = rx[i];
  };

  const int w = tid >> 6, wn = w >> 1, wk = w & 1, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  f32x16 acc[TNn][TK];
#pragma unroll
  for (int i = 0; i < TNn; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nm = (m_end - m_begin + BMr - 1) / BMr;
  if (nm > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int mt = 0; mt < nm; ++mt) {
    const int buf = mt & 1;
    if (mt + 1 < nm) gload(mt + 1);
    const float* Yb = Ys + buf * BMr * BN + wn * TNn * 32 + li;
    const float* Xb = Xs + buf * BMr * BKo + wk * TK * 32 + li;
#pragma unroll
    for (int s = 0; s < BMr / 2; ++s) {
      float a[TNn], b[TK];
#pragma unroll
      for (int i = 0; i < TNn; ++i) a[i] = Yb[(2 * s + lh) * BN + i * 32];
#pragma unroll
      for (int j = 0; j < TK; ++j) b[j] = Xb[(2 * s + lh) * BKo + j * 32];
#pragma unroll
      for (int i = 0; i < TNn; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (mt + 1 < nm) lstore(buf ^ 1);
    __syncthreads();
  }

  float* P = p.P + (long)z * p.N * p.K;
#pragma unroll
  for (int j = 0; j < TK; ++j) {
    const int col = tk * BKo + wk * TK * 32 + j * 32 + li;
    if (col >= p.K) continue;
#pragma unroll
    for (int i = 0; i < TNn; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tn * BN + wn * TNn * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.N) P[(long)row * p.K + col] = acc[i][j][r];
      }
  }
}

// dW[n, dst(k)] (+)= sum_z P[z][n][k];  conv: k = tap*Cin + c  ->  dst = c*taps + tap  (Cout,Cin,KH,KW)
__global__ void wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ dW, int splits, int N, int K,
                                    int Cin, int taps, int Kdst, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * K;
  if (idx >= total) return;
  int n = (int)(idx / K), k = (int)(idx - (long)n * K);
  float s = 0.f;
  for (int zz = 0; zz < splits; ++zz) s += P[(long)zz * total + idx];
  int dst;
  if (taps > 1) {
    int tap = k / Cin, c = k - tap * Cin;
    dst = c * taps + tap;
  } else {
    dst = k;
  }
  if (dst >= Kdst) return;
  long o = (long)n * Kdst + dst;
  dW[o] = accumulate ? dW[o] + s : s;
}
// Four consecutive k per thread (K % 4 == 0, Cin % 4 == 0 for convolutions, P 16-byte aligned) and the split loop dealt over
// 8 lanes of the workgroup: thread (quad q = tid & 31, lane l = tid >> 5) adds the slabs l, l + 8, ... (one dwordx4 each),
// the 8 lane sums are added in lane order through LDS.  A fixed order (deterministic); the dependent chain per output is
// splits / 8 loads long instead of `splits` -- the kernel was bound by that chain (23 us for 93 launches per step).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ P, float* __restrict__ dW, int splits, int N, int K,
                                                             int Cin, int taps, int Kdst, int accumulate) {
  __shared__ float4 sh[8 * 32];
  const int ql = threadIdx.x & 31, l = threadIdx.x >> 5;
  const long total = (long)N * K;
  const long idx = ((long)blockIdx.x * 32 + ql) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < total)
    for (int zz = l; zz < splits; zz += 8) {
      const float4 v = *reinterpret_cast<const float4*>(P + (long)zz * total + idx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  sh[l * 32 + ql] = s;
  __syncthreads();
  if (l != 0 || idx >= total) return;
  for (int j = 1; j < 8; ++j) { const float4 v = sh[j * 32 + ql]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const int n = (int)(idx / K), k = (int)(idx - (long)n * K);
  const float sv[4] = {s.x, s.y, s.z, s.w};
  if (taps > 1) {
    const int tap = k / Cin, c = k - tap * Cin;   // c .. c + 3 stay inside the tap (Cin % 4 == 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dst = (c + j) * taps + tap;
      if (dst < Kdst) { const long o = (long)n * Kdst + dst; dW[o] = accumulate ? dW[o] + sv[j] : sv[j]; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dst = k + j;
      if (dst < Kdst) { const long o = (long)n * Kdst + dst; dW[o] = accumulate ? dW[o] + sv[j] : sv[j]; }
    }
  }
}

// ---------------------------------------------------------------------------------------
// column sums  out[c] (+)= sum_m X[m, c]   (bias grads).  Stage 1: per row-chunk partials.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, float* __restrict__ part, int M, int N, int ld,
                                      int rows_per_chunk) {
  __shared__ float sh[4 * 64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;  // 64 columns x 4 row lanes
  const int c = blockIdx.x * 64 + cl;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s0 = 0.f, s1 = 0.f;
  if (c < N) {
    int m = m0 + rl;
    for (; m + 4 < m1; m += 8) { s0 += X[(long)m * ld + c]; s1 += X[(long)(m + 4) * ld + c]; }
    for (; m < m1; m += 4) s0 += X[(long)m * ld + c];
  }
  sh[rl * 64 + cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && c < N) part[(long)blockIdx.y * N + c] = sh[cl] + sh[64 + cl] + sh[128 + cl] + sh[192 + cl];
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nchunks, int N,
                                    int accumulate, float scale) {
  __shared__ float sh[8 * 32];
  int c;
  float tot[1];
  if (!chunk_reduce<8, 1>(part, nchunks, N, sh, c, tot)) return;
  const float s = tot[0] * scale;
  out[c] = accumulate ? out[c] + s : s;
}

// ---------------------------------------------------------------------------------------
// DEFERRED reductions (small per-GPU batches under hipGraph replay, where the step is ~1000 dependent launches of ~5 us each):
// stil_wgrad_tn_partial / stil_colsum_partial leave only their slab partials; the caller collects one StilReduceJob per product
// and stil_reduce_jobs finishes ALL of them in ceil(n / 48) launches (the jobs travel as kernel arguments: no device table, nothing
// to keep alive under graph replay) instead of one wgrad_reduce / colsum_final launch per product (137 of a step's 1055 launches).
// Same arithmetic, bit for bit, as the immediate kernels: the vector body is wgrad_reduce4_kernel's (= colsum_final_kernel's
// chunk_reduce<8>: lane l adds slabs l, l + 8, ..., the 8 lane sums are added in lane order), the scalar body wgrad_reduce_kernel's.
#define STIL_REDUCE_BATCH 48
struct ReduceJob {             // include/stil_hip.h StilReduceJob
  const float* P;              // [splits][N][K] partials
  float* dst;                  // dW: [N, Kdst] (taps == 1) or (N, Cin, KH, KW) (taps > 1)
  int splits, N, K, Cin, taps, Kdst, accumulate;
  float scale;
};
struct ReduceBatch {
  int n;
  int first_block[STIL_REDUCE_BATCH + 1];
  ReduceJob job[STIL_REDUCE_BATCH];
};
static inline bool reduce_job_vec(const ReduceJob& j) {
  return j.K % 4 == 0 && (j.taps == 1 || j.Cin % 4 == 0) && ((uintptr_t)j.P % 16) == 0;
}
static inline int reduce_job_blocks(const ReduceJob& j) {
  const long total = (long)j.N * j.K;
  return reduce_job_vec(j) ? cdiv(total / 4, 256) : cdiv(total, 256);
}
// Vector body: ONE thread per four consecutive k with eight running sums t[l] = P[l] + P[l + 8] + ... added in order at the end --
// exactly the association of wgrad_reduce4_kernel's eight row lanes (and of chunk_reduce<8>), so the bits agree, but with eight
// independent loads in flight per thread and no LDS: the deferred slabs are read cold from HBM (0.5-1.5 GB per step), where the
// lane form (one load in flight in 1/8 of the threads when a product has few slabs) managed 0.5 TB/s.
__global__ __launch_bounds__(256) void reduce_jobs_kernel(ReduceBatch b) {
  int ji = 0;
  while (ji + 1 < b.n && (int)blockIdx.x >= b.first_block[ji + 1]) ++ji;     // block-uniform (scalar loads of kernel arguments)
  const ReduceJob& j = b.job[ji];
  const int lb = blockIdx.x - b.first_block[ji];
  const long total = (long)j.N * j.K;
  const bool vec = j.K % 4 == 0 && (j.taps == 1 || j.Cin % 4 == 0) && ((uintptr_t)j.P % 16) == 0;
  if (vec) {
    const long idx = ((long)lb * 256 + threadIdx.x) * 4;
    if (idx >= total) return;
    float4 t[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) t[l] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = j.P + idx;
    int zz = 0;
    for (; zz + 8 <= j.splits; zz += 8) {
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)(zz + l) * total);
        t[l].x += v.x; t[l].y += v.y; t[l].z += v.z; t[l].w += v.w;
      }
    }
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (zz + l < j.splits) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)(zz + l) * total);
        t[l].x += v.x; t[l].y += v.y; t[l].z += v.z; t[l].w += v.w;
      }
    float4 s = t[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) { s.x += t[q].x; s.y += t[q].y; s.z += t[q].z; s.w += t[q].w; }
    const int n = (int)(idx / j.K), k = (int)(idx - (long)n * j.K);
    const float sv[4] = {s.x * j.scale, s.y * j.scale, s.z * j.scale, s.w * j.scale};
    if (j.taps == 1 && k + 3 < j.Kdst && j.Kdst % 4 == 0 && ((uintptr_t)j.dst % 16) == 0) {    // whole quad inside the row, 16-byte aligned
      float4* o = reinterpret_cast<float4*>(j.dst + (long)n * j.Kdst + k);
      float4 r = make_float4(sv[0], sv[1], sv[2], sv[3]);
      if (j.accumulate) { const float4 c = *o; r.x = c.x + sv[0]; r.y = c.y + sv[1]; r.z = c.z + sv[2]; r.w = c.w + sv[3]; }
      *o = r;
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int dst = k + e;
      if (j.taps > 1) { const int tap = k / j.Cin, c = k - tap * j.Cin; dst = (c + e) * j.taps + tap; }   // c .. c + 3 stay inside the tap
      if (dst < j.Kdst) { const long o = (long)n * j.Kdst + dst; j.dst[o] = j.accumulate ? j.dst[o] + sv[e] : sv[e]; }
    }
  } else {
    const long idx = (long)lb * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / j.K), k = (int)(idx - (long)n * j.K);
    float s = 0.f;
    for (int zz = 0; zz < j.splits; ++zz) s += j.P[(long)zz * total + idx];
    s *= j.scale;
    int dst = k;
    if (j.taps > 1) { const int tap = k / j.Cin, c = k - tap * j.Cin; dst = c * j.taps + tap; }
    if (dst >= j.Kdst) return;
    const long o = (long)n * j.Kdst + dst;
    j.dst[o] = j.accumulate ? j.dst[o] + s : s;
  }
}
extern "C" int stil_reduce_job_bytes(void) { return (int)sizeof(ReduceJob); }
extern "C" int stil_reduce_jobs(const void* jobs_host, int njobs, void* stream) {
  STIL_REQUIRE(jobs_host && njobs > 0, "stil_reduce_jobs: null pointer or no jobs");
  const ReduceJob* jobs = (const ReduceJob*)jobs_host;
  for (int j0 = 0; j0 < njobs; j0 += STIL_REDUCE_BATCH) {
    ReduceBatch b;
    b.n = njobs - j0 < STIL_REDUCE_BATCH ? njobs - j0 : STIL_REDUCE_BATCH;
    int nb = 0;
    for (int i = 0; i < b.n; ++i) {
      const ReduceJob& j = jobs[j0 + i];
      STIL_REQUIRE(j.P && j.dst && j.splits > 0 && j.N > 0 && j.K > 0 && j.taps > 0 && (j.taps == 1 || j.Cin * j.taps == j.K),
                   "stil_reduce_jobs: bad job %d", j0 + i);
      b.job[i] = j;
      b.first_block[i] = nb;
      nb += reduce_job_blocks(j);
    }
    for (int i = b.n; i <= STIL_REDUCE_BATCH; ++i) b.first_block[i] = nb;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, b);
    STIL_LAUNCH_CHECK();
  }
  return STIL_OK;
}

// ---------------------------------------------------------------------------------------
static int check_geom(const ConvGeom& g, int vec_required) {
  if (g.KH * g.KW > 1 && (g.C % 16) != 0) {
    stil_set_error("implicit-GEMM conv needs Cin %% 16 == 0 (got %d)", g.C);
    return STIL_EUNSUPPORTED;
  }
  (void)vec_required;
  return STIL_OK;
}

static inline int is_vec(const void* p, int ld) { return (((uintptr_t)p) % 16 == 0) && (ld % 4 == 0); }

static bool g_nt_attr = false;
static int gemm_nt_attr() {
  if (g_nt_attr) return STIL_OK;
  hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 2, 32, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 2, 32, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 2, 32, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 2, 32, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) { stil_set_error("gemm_nt: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return STIL_EHIP; }
  g_nt_attr = true;
  return STIL_OK;
}

// The lean 128x128 kernel (gemm_nt_big_kernel) takes a PLAIN product when: it is asked for (tune % 100 == 44, or STIL_GEMM_BIG=1 with
// the tile left automatic), every tile is interior (M, N % 128 == 0), K % 32 == 0, operands are 16-byte aligned, and the grid is
// large enough to fill the chip (>= `min_tiles` tiles of 128x128).  The launcher additionally needs the 16-byte epilogue's conditions.
static bool nt_big_wanted(int M, int N, int K, bool plain, bool vec, int srcC, int tune) {
  static const int env_big = [] { const char* e = getenv("STIL_GEMM_BIG"); return e ? atoi(e) : 0; }();   // 0 off, n > 0: on from n tiles
  static const int env_mink = [] { const char* e = getenv("STIL_GEMM_BIG_MINK"); return e ? atoi(e) : 0; }();   // automatic mode: only K >= this
  const int forced = tune % 100;
  if (forced != 44 && !(forced == 0 && env_big > 0)) return false;
  if (!plain || !vec || (M % 128) || (N % 128) || (K % 32) || srcC != K) return false;
  const long tiles = (long)(M / 128) * (N / 128);
  return forced == 44 || (tiles >= env_big && K >= env_mink);
}

// `tune` (per call, 0 = automatic; for A/B measurements): variant + 100 * bk32 + 1000 * acc2
//   variant: block tile 22 = 128x128, 21 = 128x64, 11 = 64x64;  bk32: 1 = 32-deep LDS tiles (default 16);
//   acc2: 1 = single-chain accumulation, 2 = two-level accumulation, 0 = two-level for K >= 512
// which tile variant stil_gemm_nt launches for an [M,N] output
extern "C" int stil_gemm_nt_variant(int M, int N, int tune) {
  tune %= 10000;   // + 10000 = scalar epilogue: no bearing on the tile
  const int forced = tune % 100;
  if (forced == 11 || forced == 21 || forced == 22 || forced == 12) return forced;
  // 44 (the lean 128x128 kernel for plain products) keeps the 64-row statistics granularity and falls back to 64x64 where it does not apply
  // measured (tests/tools/gemm_bench.py, profiles/r02t-u): with 6 waves per SIMD the 64x64 tile is at least as fast as
  // 128x64 / 128x128 on every shape of the step (occupancy + tile quantisation over 256 CUs outweigh operand reuse).
  (void)M; (void)N;
  return 11;
}

// may a launch with these output-side operands carry `bstats`?  (the 16-byte epilogue's conditions, as stil_gemm_nt derives them)
extern "C" int stil_gemm_nt_bstats_ok(const float* C, int N, int ldc, const float* resid, int ldr, const float* relu_mask, int ldm, const float* bs_y) {
  auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
  return (N % 4 == 0) && (ldc % 4 == 0) && (!resid || ldr % 4 == 0) && (!relu_mask || ldm % 4 == 0) && al16(relu_mask) && al16(C) && al16(resid) &&
         al16(bs_y) && bs_y != nullptr ? 1 : 0;
}

// rows per output tile of the variant stil_gemm_nt picks for [M,N] (the granularity of `colstats`)
extern "C" int stil_gemm_nt_tile_rows(int M, int N, int tune) { const int v = stil_gemm_nt_variant(M, N, tune); return (v == 11 || v == 12) ? 64 : 128; }

// The kernel instantiation stil_gemm_nt launches for these operands, as variant + 100 * bk32 + 1000 * acc2 + 10000 * vec +
// 100000 * plain + 1000000 * a_bn (bench bookkeeping: names the rocprofv3 row of a launch).  `plain`: 1x1 / stride 1 / no padding / identity
// output map, i.e. A is a plain row-major matrix.  The launcher below calls the same function.
static int nt_config(const float* A, const float* W, int M, int N, int K, int lda, int ldb, int srcC, int KH, int KW,
                     int plain, int a_bn, int tune, bool allow_big) {
  const bool want_b3 = tune >= 100000;   // the opt-in split-precision mode: reported as + 10000000 where the launch qualifies
  tune %= 10000;
  const int variant = stil_gemm_nt_variant(M, N, tune);
  // 32-deep k-tiles: every staged row is one whole 128-byte line (16-deep tiles take half a line per k-tile and count on L1 for
  // the other half), in ONE LDS buffer (two barriers per k-tile, the LDS of a 16-deep pair: as many workgroups per CU), wherever the
  // reduction allows it: K % 32 == 0 and whole taps (1x1, or Cin % 32 == 0); the operand-staging BatchNorm keeps 16-deep tiles.
  // Round 3 used them for plain products only (two buffers from K = 256 on); round 4 measured the whole step, paired runs on one
  // box: everything single-buffered 32-deep +1.0 % (8 of 8 pairs), plain K >= 256 alone -0.5 %, conv gathers alone +-0 --
  // the mix of LDS footprints of the two streams' kernels matters, not one kernel's own rate (profiles/r04_experiments.txt 12).
  const int bkd = (tune / 100) % 10;   // 0 automatic (= 3 where possible), 1 on (two LDS buffers), 2 off, 3 on with ONE LDS buffer
  const bool bk32 = (bkd == 1 || bkd == 3 || (bkd == 0 && !a_bn)) && (K % 32 == 0) && (KH * KW == 1 || srcC % 32 == 0);
  const bool single = bk32 && (bkd == 3 || bkd == 0);
  const bool acc2 = tune / 1000 == 2 || (tune / 1000 == 0 && K >= 512);
  const bool vec = is_vec(A, lda) && (srcC % 4 == 0) && is_vec(W, ldb) && (K % 4 == 0);  // every 16-byte load aligned and entirely in or out
  if (allow_big && !want_b3 && nt_big_wanted(M, N, K, plain && KH * KW == 1, vec, srcC, tune) && (long)cdiv(M, 64) * cdiv(N, 64) >= 256)
    return 44 + 200 + 1000 * (acc2 ? 1 : 0) + 10000 + 100000 + (a_bn ? 1000000 : 0);
  const bool b3 = want_b3 && vec && variant == 11 && K % 32 == 0 && (KH * KW == 1 || srcC % 32 == 0) && (!a_bn || srcC % 32 == 0);
  if (b3)     // 64x64, 32-deep, one buffer, two-level sums (+ operand-staging BatchNorm)
    return 11 + 200 + 1000 + 10000 + 100000 * ((plain && KH * KW == 1) ? 1 : 0) + (a_bn ? 1000000 : 0) + 10000000;
  if (a_bn)   // operand-staging BatchNorm: 64x64 tiles, BK = 16 (see stil_gemm_nt)
    return 11 + 1000 * (acc2 ? 1 : 0) + 10000 * (vec ? 1 : 0) + 100000 * ((plain && KH * KW == 1) ? 1 : 0) + 1000000;
  return variant + 100 * ((vec && bk32) ? (single ? 2 : 1) : 0) + 1000 * (acc2 ? 1 : 0) + 10000 * (vec ? 1 : 0) + 100000 * ((plain && KH * KW == 1) ? 1 : 0);
}

extern "C" int stil_gemm_nt_config(const float* A, const float* W, int M, int N, int K, int lda, int ldb, int srcC, int KH, int KW,
                                   int plain, int a_bn, int tune) {
  return nt_config(A, W, M, N, K, lda, ldb, srcC, KH, KW, plain, a_bn, tune, true);
}

// split-K policy (64x64 tiles only), from tests/tools/split_sweep.py's table of what a per-GPU batch of 16-32 samples launches
// (profiles/r05_split_sweep.txt; with write-through slabs a slice costs ~1 us, so the question is only whether the grid leaves CUs
// idle or a last partial round of workgroups).  Grids below one workgroup per CU: slices of >= 4 k-tiles (of 32) aiming at >= 512
// workgroups (>= 1536 when the reduction is long, >= 64 k-tiles).  Grids of 256-1535 tiles: only long reductions (>= 36 k-tiles),
// slices of >= 12 k-tiles aiming at >= 1536 workgroups (one and a half rounds of the chip's 1024 resident workgroups).  At most
// 8 slices: the last arrival reads them all.  B = 256 launches none of these shapes except the heads' M = 256 products.
// stil_gemm_nt_force_splits(s): measurement hook (tests/tools/split_sweep.py), s > 0 forces s slices wherever a product can be
// split at all (64x64 tiles, K >= 256, at most NT_SPLIT_MAX_TILES tiles, >= 2 k-tiles per slice); s <= 0 = the policy.
#define NT_SPLIT_MAX_TILES 4096
static int g_force_splits = 0;
extern "C" int stil_gemm_nt_force_splits(int s) { const int old = g_force_splits; g_force_splits = s; return old; }
static int nt_splits(int M, int N, int K, int tune) {
  if (stil_gemm_nt_variant(M, N, tune) != 11) return 1;
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
  if (g_force_splits > 0) {
    if (tiles > NT_SPLIT_MAX_TILES || K < 256) return 1;
    int s = g_force_splits;
    if (s > K / 64) s = K / 64;
    return s < 2 ? 1 : s;
  }
  static const int max_tiles = [] { const char* e = getenv("STIL_GEMM_SPLIT_MAXTILES"); return e ? atoi(e) : 1536; }();   // A/B knobs
  static const int big_target = [] { const char* e = getenv("STIL_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 1536; }();
  if (K < 256 || tiles >= max_tiles) return 1;
  const int kt = K / 32;
  int maxs, target;
  if (tiles < 256) { maxs = kt / 4; target = kt >= 64 ? 1536 : 512; }
  else { if (kt < 36) return 1; maxs = kt / 12; target = big_target; }
  int s = cdiv(target, tiles);
  if (s > maxs) s = maxs;
  if (s > 8) s = 8;
  return s < 2 ? 1 : s;
}
// column-panel width (in 64-column tiles) of the tile order, 0 = row-tile major.  Panels pay when W (N x K floats) does not fit an
// XCD's 4 MB L2 beside the streaming A panels: then every group of row tiles in flight re-fetches all of W.  With panels of <= 1 MB
// of W the price is A read once per panel instead of once: worth it while A x (panels - 1) is below what W's re-fetches cost, i.e.
// for wide, short products (N >= 1024, K <= 1024).  STIL_GEMM_PANEL = -1 (automatic, default) | 0 (off) | n (forced, A/B).
static int nt_panel(int M, int N, int K) {
  static const int forced = [] { const char* e = getenv("STIL_GEMM_PANEL"); return e ? atoi(e) : -1; }();
  const int nbn = cdiv(N, 64);
  int gw = forced;
  if (forced < 0) {
    if (N < 1024 || K > 1024 || (long)N * K * 4 <= (2l << 20)) return 0;
    gw = (int)((1l << 20) / (64l * K * 4));      // <= 1 MB of W per panel
    if (gw < 4) gw = 4;
  }
  if (gw <= 0 || gw >= nbn) return 0;
  while (gw > 1 && nbn % gw) --gw;                // whole panels only
  (void)M;
  return gw > 1 ? gw : 0;
}

// bytes of the split-K workspace stil_gemm_nt wants for this product (0: the product is not split).  Layout: a FIXED ticket
// region of NT_SPLIT_TICKET_BYTES (one int per tile; split products have at most NT_SPLIT_MAX_TILES tiles) then [tiles][splits][64*64] float
// slabs -- fixed, so that one workspace can serve products of different shapes: no product's slabs ever overlap another's tickets.
// The caller zeroes it ONCE (the tickets); every launch leaves the tickets zero again.  One workspace per stream: launches that
// may run concurrently must not share one.
#define NT_SPLIT_TICKET_BYTES (4 * NT_SPLIT_MAX_TILES)
extern "C" size_t stil_gemm_nt_split_workspace_bytes(int M, int N, int K, int tune) {
  const int s = nt_splits(M, N, K, tune % 10000);
  if (s <= 1) return 0;
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
  return (size_t)NT_SPLIT_TICKET_BYTES + (size_t)tiles * s * 4096 * sizeof(float);
}

extern "C" int stil_gemm_nt(const float* A, const float* W, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad_y,
                            int pad_x, int mode, int out_stride, int out_py, int out_px, int out_OH, int out_OW,
                            const float* bias, const float* sub, const float* scale,
                            const float* shift, const float* resid, int ldr, float* pre, int act, float alpha,
                            float* colstats, const float* a_bn, const float* relu_mask, int ldm,
                            const float* bs_y, const float* bs_stats, float* bstats, int bs_relu, int bs_tile0,
                            const float* scale_var, float var_eps, void* split_ws, size_t split_ws_bytes, int tune_arg, void* stream) {
  const bool want_b3 = tune_arg >= 100000;           // + 100000: the opt-in split-precision mode (bf16x3) wherever a launch qualifies
  if (want_b3) tune_arg -= 100000;
  const int tune = tune_arg % 10000;                 // + 10000: scalar (one dword per lane) epilogue, for A/B measurements and tests
  const bool scalar_epilogue = tune_arg / 10000 == 1;
  STIL_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "stil_gemm_nt: null pointer or empty shape (M=%d N=%d K=%d)", M, N, K);
  STIL_REQUIRE(tune_arg >= 0 && tune_arg < 20000 && tune < 3000 && (tune % 100 == 0 || tune % 100 == 11 || tune % 100 == 21 || tune % 100 == 22 || tune % 100 == 12 || tune % 100 == 44) && (tune / 100) % 10 <= 3,
               "stil_gemm_nt: bad tune %d", tune);
  STIL_REQUIRE(KH * KW * srcC == K, "stil_gemm_nt: K=%d != KH*KW*C=%d", K, KH * KW * srcC);
  STIL_REQUIRE(M % (OH * OW) == 0, "stil_gemm_nt: M=%d not a multiple of OH*OW=%d", M, OH * OW);
  GemmNTArgs p;
  p.A = A; p.Bw = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.g = ConvGeom{srcH, srcW, srcC, OH, OW, KH, KW, stride, pad_y, pad_x, mode};
  p.os = out_stride < 1 ? 1 : out_stride; p.opy = out_py; p.opx = out_px; p.oOH = out_OH; p.oOW = out_OW;
  p.var = scale_var; p.var_eps = var_eps;
  p.splits = 1; p.split_slabs = nullptr; p.split_tickets = nullptr;
  p.panel = 0;
  {
    const size_t need = stil_gemm_nt_split_workspace_bytes(M, N, K, tune_arg);
    if (split_ws && need > 0 && split_ws_bytes >= need && ((uintptr_t)split_ws % 256) == 0) {
      const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
      p.splits = nt_splits(M, N, K, tune_arg % 10000);
      p.split_tickets = (int*)split_ws;
      p.split_slabs = (float*)((char*)split_ws + NT_SPLIT_TICKET_BYTES);
      (void)tiles;
    }
  }
  STIL_REQUIRE(!scale_var || scale, "stil_gemm_nt: scale_var divides `scale`, which is missing");
  p.bias = bias; p.sub = sub; p.scale = scale; p.shift = shift; p.resid = resid; p.ldr = ldr; p.pre = pre; p.act = act;
  p.alpha = alpha;
  p.colstats = colstats;
  p.abn = a_bn;
  p.rmask = relu_mask; p.ldm = ldm;
  STIL_REQUIRE(!relu_mask || !colstats, "stil_gemm_nt: relu_mask and colstats exclude each other");
  {
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
    p.wide = (N % 4 == 0) && (ldc % 4 == 0) && (!resid || ldr % 4 == 0) && (!relu_mask || ldm % 4 == 0) && al16(relu_mask) && al16(C) && al16(resid) && al16(pre) && al16(bias) && al16(sub) &&
             al16(scale) && al16(scale_var) && al16(shift) && !scalar_epilogue ? 1 : 0;
  }
  p.bs_y = bs_y; p.bs_stats = bs_stats; p.bstats = bstats; p.ldbs = ldc; p.bs_relu = bs_relu; p.bs_tile0 = bs_tile0;
  STIL_REQUIRE(!bstats || (bs_y && bs_stats && (bs_relu == 0 || bs_relu == 2) && bs_tile0 >= 0 && !colstats && !a_bn && p.wide &&
                           ((uintptr_t)bs_y % 16) == 0 && ((uintptr_t)bs_stats % 16) == 0 && (tune % 100 == 0 || tune % 100 == 11 || tune % 100 == 44)),
               "stil_gemm_nt: bstats needs bs_y / bs_stats, bs_relu 0 or 2, the 16-byte epilogue on 64x64 tiles (stil_gemm_nt_bstats_ok) "
               "and excludes colstats / a_bn");
  STIL_REQUIRE(!colstats || (p.os == 1 && !bias && !sub && !scale && !shift && !resid && act == 0),
               "stil_gemm_nt: colstats describes the raw product (no bias / affine / residual / activation / output map)");
  p.vecA = is_vec(A, lda) && (srcC % 4 == 0);
  p.vecB = is_vec(W, ldb);
  int rc = check_geom(p.g, 0);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  // plain row-major A and identity output map: the geometry code is compiled out (gemm_nt_kernel<..., PLAIN>)
  const bool plain = KH * KW == 1 && stride == 1 && pad_y == 0 && pad_x == 0 && mode == 0 && p.os == 1 && srcH == OH && srcW == OW;
  const int cfg = nt_config(A, W, M, N, K, lda, ldb, srcC, KH, KW, plain ? 1 : 0, a_bn ? 1 : 0, tune, false);   // (the lean 128x128 kernel is decided below)
  const int variant = cfg % 100;
  const bool bk32 = (cfg / 100) % 10 >= 1;   // BK = 32: whole 128-byte lines per staged row; needs whole taps and vector loads
  const bool single = (cfg / 100) % 10 == 2; // ... with one LDS buffer (64x64 tiles only)
  const bool acc2 = (cfg / 1000) % 10 == 1;  // long reductions: two-level accumulation (see gemm_nt_kernel)
  const bool vec = (cfg / 10000) % 10 == 1;
  rc = gemm_nt_attr();
  if (rc) return rc;
  if (variant == 11 && p.splits == 1) p.panel = nt_panel(M, N, K);
  if (!want_b3 && variant == 11 && p.splits == 1 && p.wide && !scalar_epilogue && mode == 0 && (!a_bn || srcC <= 2048) &&
      nt_big_wanted(M, N, K, plain, vec, srcC, tune) && (long)cdiv(M, 64) * cdiv(N, 64) >= 256) {
    p.panel = (p.panel > 0 && p.panel % 2 == 0 && (N / 128) % (p.panel / 2) == 0) ? p.panel / 2 : 0;   // panels in 128-column tiles
    const dim3 grid_((M / 128) * (N / 128));
    const size_t lds_ = (size_t)256 * 36 * sizeof(float) + (a_bn ? (size_t)3 * srcC * sizeof(float) : 0);
    static bool attr_ = false;
    if (!attr_) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_ = true;
    }
    if (acc2 && a_bn) hipLaunchKernelGGL((gemm_nt_big_kernel<true, true>), grid_, dim3(256), lds_, s, p);
    else if (acc2) hipLaunchKernelGGL((gemm_nt_big_kernel<true, false>), grid_, dim3(256), lds_, s, p);
    else if (a_bn) hipLaunchKernelGGL((gemm_nt_big_kernel<false, true>), grid_, dim3(256), lds_, s, p);
    else hipLaunchKernelGGL((gemm_nt_big_kernel<false, false>), grid_, dim3(256), lds_, s, p);
    STIL_LAUNCH_CHECK();
    return STIL_OK;
  }
  if (a_bn) {   // BatchNorm + ReLU of the producing layer applied while A is staged: 64x64 tiles, vector loads, BK = 16 only
    STIL_REQUIRE(vec && !bk32 && mode == 0 && srcC % 16 == 0 && srcC <= 2048 && (tune % 100 == 0 || tune % 100 == 11 || tune % 100 == 44),
                 "stil_gemm_nt: a_bn runs 64x64 tiles only (tune %% 100 must be 0 or 11: the caller sizes colstats by "
                 "stil_gemm_nt_tile_rows) and needs 16-byte aligned operands, a forward gather, Cin %% 16 == 0 and Cin <= 2048 (Cin=%d)", srcC);
    const dim3 grid_(cdiv(M, 64) * cdiv(N, 64) * p.splits);
    if (want_b3 && srcC % 32 == 0 && K % 32 == 0 && !scalar_epilogue) {   // split-precision mode: 32-deep k-tiles, two-level sums
      const size_t lds3_ = (size_t)3 * 128 * 40 * 2 + (size_t)3 * srcC * sizeof(float);
      if (plain) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, true, true, true, 1, true>), grid_, dim3(256), lds3_, s, p);
      else hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, true, false, true, 1, true>), grid_, dim3(256), lds3_, s, p);
      STIL_LAUNCH_CHECK();
      return STIL_OK;
    }
    const size_t lds_ = (size_t)2 * 64 * 2 * 20 * sizeof(float) + (size_t)3 * srcC * sizeof(float);
    if (acc2 && plain) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 16, true, true, true, true>), grid_, dim3(256), lds_, s, p);
    else if (acc2) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 16, true, true, false, true>), grid_, dim3(256), lds_, s, p);
    else if (plain) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 16, true, false, true, true>), grid_, dim3(256), lds_, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 16, true, false, false, true>), grid_, dim3(256), lds_, s, p);
    STIL_LAUNCH_CHECK();
    return STIL_OK;
  }
#define LAUNCH_NT(TM_, TN_, BK_, V_)                                                                          \
  do {                                                                                                        \
    const dim3 grid_(cdiv(M, 64 * TM_) * cdiv(N, 64 * TN_) * ((TM_) * (TN_) == 1 ? p.splits : 1));              \
    const size_t lds_ = (size_t)2 * 64 * (TM_ + TN_) * (BK_ + 4) * sizeof(float);                             \
    if (acc2 && plain) hipLaunchKernelGGL((gemm_nt_kernel<TM_, TN_, BK_, V_, true, true>), grid_, dim3(256), lds_, s, p);   \
    else if (acc2) hipLaunchKernelGGL((gemm_nt_kernel<TM_, TN_, BK_, V_, true, false>), grid_, dim3(256), lds_, s, p);      \
    else if (plain) hipLaunchKernelGGL((gemm_nt_kernel<TM_, TN_, BK_, V_, false, true>), grid_, dim3(256), lds_, s, p);     \
    else hipLaunchKernelGGL((gemm_nt_kernel<TM_, TN_, BK_, V_, false, false>), grid_, dim3(256), lds_, s, p);               \
  } while (0)
  if (want_b3 && vec && variant == 11 && K % 32 == 0 && (KH * KW == 1 || srcC % 32 == 0) && !scalar_epilogue) {
    const dim3 grid_(cdiv(M, 64) * cdiv(N, 64) * p.splits);
    const size_t lds3_ = (size_t)3 * 128 * 40 * 2;
    if (plain) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, true, true, false, 1, true>), grid_, dim3(256), lds3_, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, true, false, false, 1, true>), grid_, dim3(256), lds3_, s, p);
    STIL_LAUNCH_CHECK();
    return STIL_OK;
  }
  if (!vec) {  // unaligned / ragged operands (K = 286 classifier gradients ...): scalar guarded loads
    if (variant == 11) LAUNCH_NT(1, 1, 16, false); else if (variant == 21) LAUNCH_NT(2, 1, 16, false); else if (variant == 12) LAUNCH_NT(1, 2, 16, false); else LAUNCH_NT(2, 2, 16, false);
  } else if (variant == 11 && single) {
    const dim3 grid_(cdiv(M, 64) * cdiv(N, 64) * p.splits);
    const size_t lds_ = (size_t)64 * 2 * 36 * sizeof(float);
    if (acc2 && plain) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, true, true, false, 1>), grid_, dim3(256), lds_, s, p);
    else if (acc2) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, true, false, false, 1>), grid_, dim3(256), lds_, s, p);
    else if (plain) hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, false, true, false, 1>), grid_, dim3(256), lds_, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<1, 1, 32, true, false, false, false, 1>), grid_, dim3(256), lds_, s, p);
  } else if (variant == 11) { if (bk32) LAUNCH_NT(1, 1, 32, true); else LAUNCH_NT(1, 1, 16, true); }
  else if (variant == 12) { LAUNCH_NT(1, 2, 16, true); }
  else if (variant == 21) { if (bk32) LAUNCH_NT(2, 1, 32, true); else LAUNCH_NT(2, 1, 16, true); }
  else { if (bk32) LAUNCH_NT(2, 2, 32, true); else LAUNCH_NT(2, 2, 16, true); }
#undef LAUNCH_NT
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

// block tile of the weight-gradient kernel: 22 = 128x128 (64x128 for N <= 64), 11 = 64x64; `tune` forces one (0 = automatic).
// measured in situ (bench.py --breakdown, cold caches): 64x64 tiles win for N <= 256 (more blocks, fewer M-splits) except the
// 1x1 convolutions with 128..256 outputs and K in 256..1024, where 128x128 halves the operand re-reads
// Small products (M < 8192 rows, or M < 32768 with fewer than 256 K outputs: what a per-GPU batch of 16-64 samples launches from
// layer2 on, and the heads at any batch): 64x64 -- the 128x128 tile's few workgroups take a flat ~30 us there (tests/tools/
// split_sweep.py tn, profiles/r05_split_sweep.txt: cardiac share of 16, 618 -> 424 us over the twenty products; B = 32: 1111 ->
// 1031).  The B = 256 step keeps 128x128 on its M = 12544 ... 50176 products: moving the row bound to 16384 / 32768 / 65536 costs
// it 0.7 / 1.2 / 1.6 % (paired, profiles/r05_experiments.txt 14).  STIL_WGRAD_SMALLM: the row bound (A/B knob).
static inline int tn_variant(int M, int N, int K, int tune) {
  static const int small_m = [] { const char* e = getenv("STIL_WGRAD_SMALLM"); return e ? atoi(e) : 8192; }();
  if (tune == 11 || tune == 22) return tune;
  if (M < small_m || (M < 4 * small_m && (long)N * K < 262144)) return 11;
  if (N > 256) return 22;
  return (N >= 128 && K >= 256 && K <= 1024) ? 22 : 11;
}
// stil_wgrad_force_splits(s): measurement hook (tests/tools/split_sweep.py tn), s > 0 forces s slabs (slabs of >= 16 rows)
static int g_force_wsplits = 0;
extern "C" int stil_wgrad_force_splits(int s) { const int old = g_force_wsplits; g_force_wsplits = s; return old; }
static int wgrad_splits(int M, int N, int K, int tune) {
  if (g_force_wsplits > 0) { const int m = M / 16 > 0 ? M / 16 : 1; const int s = g_force_wsplits < m ? g_force_wsplits : m; return s > 256 ? 256 : s; }
  const int tv = tn_variant(M, N, K, tune);
  int tiles = tv == 11 ? cdiv(N, 64) * cdiv(K, 64) : cdiv(N, N <= 64 ? 64 : 128) * cdiv(K, 128);
  int want = cdiv(tv == 11 ? 2304 : 768, tiles);
  int maxs = M / 256 > 0 ? M / 256 : 1;
  int s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return s;
}

extern "C" size_t stil_wgrad_workspace_bytes(int M, int N, int K, int tune) {
  return (size_t)wgrad_splits(M, N, K, tune) * N * K * sizeof(float);
}

// dW (+)= dY^T . Xgather ; dW laid out [N, Kdst] (taps==1) or (N, Cin, KH, KW) (taps>1)
extern "C" int stil_wgrad_splits(int M, int N, int K, int tune) { return wgrad_splits(M, N, K, tune); }
static int wgrad_tn_impl(const float* dY, const float* X, float* dW, int M, int N, int K, int ldy, int ldx,
                             int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad,
                             int Kdst, int accumulate, const float* x_bn, float* workspace, size_t workspace_bytes, int tune, void* stream) {
  STIL_REQUIRE(dY && X && workspace, "stil_wgrad_tn: null pointer");
  STIL_REQUIRE(tune == 0 || tune == 11 || tune == 22, "stil_wgrad_tn: bad tune %d", tune);
  STIL_REQUIRE(KH * KW * srcC == K, "stil_wgrad_tn: K=%d != KH*KW*C=%d", K, KH * KW * srcC);
  STIL_REQUIRE(M % (OH * OW) == 0, "stil_wgrad_tn: M=%d not a multiple of OH*OW", M);
  STIL_REQUIRE(srcC % 4 == 0 || KH * KW == 1, "stil_wgrad_tn: conv needs Cin %% 4 == 0");
  int splits = wgrad_splits(M, N, K, tune);
  STIL_REQUIRE(workspace_bytes >= (size_t)splits * N * K * sizeof(float), "stil_wgrad_tn: workspace too small");
  GemmTNArgs p;
  p.Y = dY; p.X = X; p.P = workspace; p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.ldx = ldx;
  p.g = ConvGeom{srcH, srcW, srcC, OH, OW, KH, KW, stride, pad, pad, 0};
  int mps = cdiv(M, splits);
  mps = ((mps + 15) / 16) * 16;
  p.m_per_split = mps;
  p.vecY = is_vec(dY, ldy);
  p.vecX = is_vec(X, ldx) && (srcC % 4 == 0);
  p.xbn = x_bn;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = p.vecY && p.vecX && (N % 4 == 0) && (K % 4 == 0);
  STIL_REQUIRE(!x_bn || (vec && srcC % 4 == 0), "stil_wgrad_tn: x_bn needs 16-byte aligned operands and Cin %% 4 == 0");
  if (tn_variant(M, N, K, tune) == 11) {
    dim3 grid(cdiv(N, 64) * cdiv(K, 64) * splits);
    if (x_bn) hipLaunchKernelGGL((gemm_tn_kernel<1, 1, true, true>), grid, dim3(256), 0, s, p);
    else if (vec) hipLaunchKernelGGL((gemm_tn_kernel<1, 1, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<1, 1, false>), grid, dim3(256), 0, s, p);
  } else if (N <= 64) {
    dim3 grid(cdiv(N, 64) * cdiv(K, 128) * splits);
    if (x_bn) hipLaunchKernelGGL((gemm_tn_kernel<1, 2, true, true>), grid, dim3(256), 0, s, p);
    else if (vec) hipLaunchKernelGGL((gemm_tn_kernel<1, 2, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<1, 2, false>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(cdiv(N, 128) * cdiv(K, 128) * splits);
    if (x_bn) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, true, true>), grid, dim3(256), 0, s, p);
    else if (vec) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, false>), grid, dim3(256), 0, s, p);
  }
  STIL_LAUNCH_CHECK();
  if (!dW) return STIL_OK;     // stil_wgrad_tn_partial: the caller reduces the slabs later (stil_reduce_jobs)
  long total = (long)N * K;
  int taps = KH * KW;
  if (K % 4 == 0 && (taps == 1 || srcC % 4 == 0) && ((uintptr_t)workspace % 16) == 0)
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(cdiv(total / 4, 32)), dim3(256), 0, s, workspace, dW, splits, N, K, srcC,
                       taps, Kdst, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, workspace, dW, splits, N, K, srcC,
                       taps, Kdst, accumulate);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_wgrad_tn(const float* dY, const float* X, float* dW, int M, int N, int K, int ldy, int ldx,
                             int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad,
                             int Kdst, int accumulate, const float* x_bn, float* workspace, size_t workspace_bytes, int tune, void* stream) {
  STIL_REQUIRE(dW, "stil_wgrad_tn: null pointer");
  return wgrad_tn_impl(dY, X, dW, M, N, K, ldy, ldx, srcH, srcW, srcC, OH, OW, KH, KW, stride, pad, Kdst, accumulate, x_bn, workspace,
                       workspace_bytes, tune, stream);
}
// the slab partials only: workspace = [stil_wgrad_splits(M, N, K, tune)][N][K]; finish with a StilReduceJob (stil_reduce_jobs)
extern "C" int stil_wgrad_tn_partial(const float* dY, const float* X, int M, int N, int K, int ldy, int ldx,
                                     int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad,
                                     const float* x_bn, float* workspace, size_t workspace_bytes, int tune, void* stream) {
  return wgrad_tn_impl(dY, X, nullptr, M, N, K, ldy, ldx, srcH, srcW, srcC, OH, OW, KH, KW, stride, pad, K, 0, x_bn, workspace,
                       workspace_bytes, tune, stream);
}

extern "C" size_t stil_colsum_workspace_bytes(int M, int N) {
  int rpc = 128;
  return (size_t)cdiv(M, rpc) * N * sizeof(float);
}

static int colsum_impl(const float* X, float* out, int M, int N, int ld, int accumulate, float scale,
                       float* workspace, size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(X && workspace && M > 0 && N > 0, "stil_colsum: null pointer or empty shape");
  int rpc = 128, nch = cdiv(M, rpc);
  STIL_REQUIRE(workspace_bytes >= (size_t)nch * N * sizeof(float), "stil_colsum: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), nch), dim3(256), 0, s, X, workspace, M, N, ld, rpc);
  STIL_LAUNCH_CHECK();
  if (!out) return STIL_OK;    // stil_colsum_partial
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 32)), dim3(256), 0, s, workspace, out, nch, N, accumulate, scale);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_colsum(const float* X, float* out, int M, int N, int ld, int accumulate, float scale,
                           float* workspace, size_t workspace_bytes, void* stream) {
  STIL_REQUIRE(out, "stil_colsum: null pointer or empty shape");
  return colsum_impl(X, out, M, N, ld, accumulate, scale, workspace, workspace_bytes, stream);
}
// the per-chunk partials only: workspace = [stil_colsum_chunks(M)][N]; finish with a StilReduceJob {splits = chunks, N = 1, K = N}
extern "C" int stil_colsum_chunks(int M) { return cdiv(M, 128); }
extern "C" int stil_colsum_partial(const float* X, int M, int N, int ld, float* workspace, size_t workspace_bytes, void* stream) {
  return colsum_impl(X, nullptr, M, N, ld, 0, 1.f, workspace, workspace_bytes, stream);
}
