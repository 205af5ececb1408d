// Per-step weight re-layouts in ONE launch.  The convolution / Linear weights live in the reference layout
// (Cout,Cin,KH,KW) / (N,K) inside the flat parameter slab (state_dict compatible); the NT GEMM wants
//   kind 0  [Cout][tap][Cin]            forward operand of a k > 1 convolution
//   kind 1  [Cin][tap][Cout]            input-gradient operand (k = 1 or Linear: the plain transpose [K][N])
//   kind 2  [Cin][ky'][kx'][Cout]       one output phase of a strided input-gradient (conv_w_phase_kernel's map)
// of every layer, once per optimisation step (weights change only in Adam / the EMA update).  Round 3 issued one small
// kernel per layer and use: 136 launches per step (76 transposes, 45 layouts, 15 phases) of 4-6 us each.  Here a job table
// (built once per model on the host) drives one grid: block b copies 1024 destination elements of job blk2job[b]
// (destination-linear: coalesced stores, gathered loads from a tensor that is L2-resident for its whole job).
#include "common.h"

struct LayoutJob {
  long long src, dst;          // float offsets into the source slab / the layout slab
  int kind, Cout, Cin, KH, KW; // source tensor (Cout, Cin, KH, KW)
  int stride, ky0, kx0, KHs, KWs;  // kind 2
  int first_block, pad_;       // first grid block of this job
};

__global__ __launch_bounds__(256) void weight_layouts_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                              const LayoutJob* __restrict__ jobs, const int* __restrict__ blk2job) {
  const LayoutJob j = jobs[blk2job[blockIdx.x]];
  const float* __restrict__ w = src_base + j.src;
  float* __restrict__ out = dst_base + j.dst;
  const int taps = j.KH * j.KW;
  const long total = j.kind == 2 ? (long)j.Cin * j.KHs * j.KWs * j.Cout : (long)j.Cout * j.Cin * taps;
  const long i0 = ((long)blockIdx.x - j.first_block) * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = i0 + u * 256;
    if (i >= total) break;
    long s;
    if (j.kind == 0) {          // out[(co*taps + tap)*Cin + ci]
      const int ci = (int)(i % j.Cin);
      const long t = i / j.Cin;
      const int tap = (int)(t % taps), co = (int)(t / taps);
      s = ((long)co * j.Cin + ci) * taps + tap;
    } else if (j.kind == 1) {   // out[(ci*taps + tap)*Cout + co]
      const int co = (int)(i % j.Cout);
      const long t = i / j.Cout;
      const int tap = (int)(t % taps), ci = (int)(t / taps);
      s = ((long)co * j.Cin + ci) * taps + tap;
    } else {                    // out[((ci*KHs + ky)*KWs + kx)*Cout + co] = w[co][ci][ky0 + s*(KHs-1-ky)][kx0 + s*(KWs-1-kx)]
      const int co = (int)(i % j.Cout);
      long t = i / j.Cout;
      const int kx = (int)(t % j.KWs); t /= j.KWs;
      const int ky = (int)(t % j.KHs);
      const int ci = (int)(t / j.KHs);
      const int sy = j.ky0 + j.stride * (j.KHs - 1 - ky), sx = j.kx0 + j.stride * (j.KWs - 1 - kx);
      s = (((long)co * j.Cin + ci) * j.KH + sy) * j.KW + sx;
    }
    out[i] = w[s];
  }
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
