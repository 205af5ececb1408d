// Per-step weight re-layouts in ONE launch.  The convolution / Linear weights live in the reference layout
// (Cout,Cin,KH,KW) / (N,K) inside the flat parameter slab (state_dict compatible); the NT GEMM wants
//   kind 0  [Cout][tap][Cin]            forward operand of a k > 1 convolution
//   kind 1  [Cin][tap][Cout]            input-gradient operand (k = 1 or Linear: the plain transpose [K][N])
//   kind 2  [Cin][ky'][kx'][Cout]       one output phase of a strided input-gradient (conv_w_phase_kernel's map)
// of every layer, once per optimisation step (weights change only in Adam / the EMA update).  Round 3 issued one small
// kernel per layer and use: 136 launches per step (76 transposes, 45 layouts, 15 phases) of 4-6 us each.  Here a job table
// (built once per model on the host) drives one grid; block b works on job blk2job[b] (its (b - first_block)-th block).
#include "common.h"

struct LayoutJob {
  long long src, dst;          // float offsets into the source slab / the layout slab
  int kind, Cout, Cin, KH, KW; // source tensor (Cout, Cin, KH, KW)
  int stride, ky0, kx0, KHs, KWs;  // kind 2
  int first_block, pad_;       // first grid block of this job
};

// Blocks per job (the host builds blk2job with the same rule, stil_weight_layout_job_blocks):
//   kind 1: one 32x32 tile of the [Cout] x [Cin*taps] -> [Cin*taps] x [Cout] transpose per block (through LDS: both sides coalesced);
//   kind 0: one (output channel, 128-input-channel chunk) per block: its 128*taps source floats are contiguous, staged in LDS and
//           written back as `taps` runs of 128 contiguous floats;
//   kind 2: 1024 destination elements per block (a gather over the few strided convolutions: 4 of 53 layers).
__global__ __launch_bounds__(256) void weight_layouts_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                              const LayoutJob* __restrict__ jobs, const int* __restrict__ blk2job) {
  __shared__ float buf[1152];      // kind 0: 128 x 9; kind 1: 32 x 33
  const LayoutJob j = jobs[blk2job[blockIdx.x]];
  const float* __restrict__ w = src_base + j.src;
  float* __restrict__ out = dst_base + j.dst;
  const int taps = j.KH * j.KW;
  const int lb = blockIdx.x - j.first_block, tid = threadIdx.x;
  if (j.kind == 1) {               // out[c * R + r] = w[r * C + c],  R = Cout, C = Cin * taps
    const int R = j.Cout, C = j.Cin * taps, tiles_c = (C + 31) / 32;
    const int tr = lb / tiles_c, tc = lb - tr * tiles_c, tx = tid & 31, ty = tid >> 5;
    for (int k = ty; k < 32; k += 8) {
      const int r = tr * 32 + k, c = tc * 32 + tx;
      buf[k * 33 + tx] = (r < R && c < C) ? w[(long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int c = tc * 32 + k, r = tr * 32 + tx;
      if (r < R && c < C) out[(long)c * R + r] = buf[tx * 33 + k];
    }
  } else if (j.kind == 0) {        // out[(co*taps + tap)*Cin + ci] = w[(co*Cin + ci)*taps + tap]
    const int chunks = (j.Cin + 127) / 128;
    const int co = lb / chunks, ci0 = (lb - co * chunks) * 128, n = min(128, j.Cin - ci0);
    const float* s = w + ((long)co * j.Cin + ci0) * taps;
    for (int e = tid; e < n * taps; e += 256) buf[e] = s[e];
    __syncthreads();
    for (int e = tid; e < n * taps; e += 256) {
      const int tap = e / n, c = e - tap * n;
      out[((long)co * taps + tap) * j.Cin + ci0 + c] = buf[c * taps + tap];
    }
  } else {                         // out[((ci*KHs + ky)*KWs + kx)*Cout + co] = w[co][ci][ky0 + s*(KHs-1-ky)][kx0 + s*(KWs-1-kx)]
    const long total = (long)j.Cin * j.KHs * j.KWs * j.Cout;
    const long i0 = (long)lb * 1024 + tid;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = i0 + u * 256;
      if (i >= total) break;
      const int co = (int)(i % j.Cout);
      long t = i / j.Cout;
      const int kx = (int)(t % j.KWs); t /= j.KWs;
      const int ky = (int)(t % j.KHs);
      const int ci = (int)(t / j.KHs);
      const int sy = j.ky0 + j.stride * (j.KHs - 1 - ky), sx = j.kx0 + j.stride * (j.KWs - 1 - kx);
      out[i] = w[(((long)co * j.Cin + ci) * j.KH + sy) * j.KW + sx];
    }
  }
}

// blocks of one job (the rule above), for the host's blk2job table
extern "C" int stil_weight_layout_job_blocks(int kind, int Cout, int Cin, int KH, int KW, int KHs, int KWs) {
  const int taps = KH * KW;
  if (taps > 9 && kind == 0) return -1;     // the LDS stage of kind 0 holds 128 x 9 floats
  if (kind == 1) return cdiv(Cout, 32) * cdiv((long)Cin * taps, 32);
  if (kind == 0) return Cout * cdiv(Cin, 128);
  return cdiv((long)Cin * KHs * KWs * Cout, 1024);
}

extern "C" int stil_weight_layouts(const float* src_base, float* dst_base, const void* jobs, const int* blk2job, int n_blocks,
                                   void* stream) {
  STIL_REQUIRE(src_base && dst_base && jobs && blk2job && n_blocks > 0, "stil_weight_layouts: null pointer or empty plan");
  hipLaunchKernelGGL(weight_layouts_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, src_base, dst_base,
                     (const LayoutJob*)jobs, blk2job);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
extern "C" int stil_weight_layout_job_bytes(void) { return (int)sizeof(LayoutJob); }
