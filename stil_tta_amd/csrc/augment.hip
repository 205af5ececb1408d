// Device side of the input pipeline: what the reference's DataLoader workers do per sample on
// the CPU -- datasets/ContrastiveImagingAndTabularDataset.py:146-158 (marginal-distribution tabular corruption),
// :177-198 + utils/utils.py:46-70 (RandomResizedCrop / Resize, HorizontalFlip, ColorJitter, ToGray, convert_to_ts) --
// as batch kernels.  All of it is HBM-bound byte/float shuffling: coalesced channel-plane writes, one launch per batch.
// Random draws are ARGUMENTS (boxes, flips, factors, column indices, table rows): the host wrapper draws them (or the
// tests inject them), so results are reproducible and the kernels are pure functions.
#include "common.h"

// ---------------------------------------------------------------- tabular corruption
// out[b, :] = clean[b, :];  out[b, idx[b, j]] = marginal[idx[b, j], pos[b, j]]  for j < k
// (marginal = the training table transposed: [n_cols, n_rows], ContrastiveImagingAndTabularDataset.py:121-127).
__global__ __launch_bounds__(64) void tab_corrupt_kernel(const float* __restrict__ clean, const float* __restrict__ marginal,
                                                          const int* __restrict__ idx, const int* __restrict__ pos,
                                                          float* __restrict__ out, int n, int N, int k) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < n; c += 64) out[(long)b * n + c] = clean[(long)b * n + c];
  __syncthreads();   // same block wrote the row: the scatter below lands after the copy
  for (int j = threadIdx.x; j < k; j += 64) {
    const int c = idx[(long)b * k + j];
    out[(long)b * n + c] = marginal[(long)c * N + pos[(long)b * k + j]];
  }
}

// counter-based hash RNG (same family as rng_mask_kernel): uniform 32-bit word for (seed, counter)
__device__ __forceinline__ unsigned int aug_hash(unsigned long long seed, unsigned long long ctr) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (ctr + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (unsigned int)(z >> 16);
}

// Draws of `corrupt` on the device: per row k DISTINCT columns (partial Fisher-Yates: random.sample semantics) and k
// uniform table rows (np.random.choice with replacement).  One wave per row, lane 0 walks the permutation in LDS.
__global__ __launch_bounds__(64) void tab_corrupt_draw_kernel(int* __restrict__ idx, int* __restrict__ pos, int n, int N, int k,
                                                               unsigned long long seed, unsigned long long offset,
                                                               const unsigned long long* __restrict__ step) {
  extern __shared__ int perm[];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < n; c += 64) perm[c] = c;
  __syncthreads();
  const unsigned long long base = offset + (step ? step[0] : 0ULL) * 0x100000000ULL + (unsigned long long)b * (2ULL * k);
  if (threadIdx.x == 0) {
    for (int j = 0; j < k; ++j) {
      const int r = j + (int)(((unsigned long long)aug_hash(seed, base + j) * (unsigned long long)(n - j)) >> 32);
      const int t = perm[j]; perm[j] = perm[r]; perm[r] = t;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 64) {
    idx[(long)b * k + j] = perm[j];
    pos[(long)b * k + j] = (int)(((unsigned long long)aug_hash(seed, base + k + j) * (unsigned long long)N) >> 32);
  }
}

// ---------------------------------------------------------------- crop + resize + flip + (colour) + to-tensor
// One output pixel per thread, all 3 channels.  Source: uint8 HWC [B, H, W, 3] (the `.npy` images the reference loads
// with augmentation_speedup) or float CHW [B, 3, H, W].  box[b] = (top, left, height, width) of the crop in source
// pixels (RandomResizedCrop.get_params; the whole image for the plain Resize of default_transform), flip[b] != 0
// mirrors the crop horizontally.  Bilinear with half-pixel centres, no antialias (torchvision 0.12 tensor resize ==
// F.interpolate(mode="bilinear", align_corners=False)): src = (dst + 0.5) * in/out - 0.5, clamped at 0; the right /
// bottom neighbour is clamped to the crop.  Colour (optional, per sample, utils/utils.py:50-52): jitter[b] =
// (brightness, contrast, saturation, gray) applied as torchvision's float ops in the fixed order brightness ->
// contrast -> saturation -> grayscale; contrast needs the mean grey level of the (brightness-adjusted) source image,
// which the host passes as gmean[b] (stil_aug_gray_mean).  The pointwise colour ops run on the resized crop (the
// reference jitters before it crops; the two orders differ only where the [0, 1] clamp is active).  Output float CHW [B, 3, P, P], values scaled by `scale` (1/255 for
// uint8 sources: convert_to_ts) and clipped to [0, 1] like the reference's np.clip.
struct AugArgs {
  const unsigned char* src_u8; const float* src_f32;
  const int* box; const unsigned char* flip; const float* jitter; const float* gmean;
  float* out; int H, W, P; float scale;
};

__device__ __forceinline__ void aug_fetch(const AugArgs& p, int b, int y, int x, float (&v)[3]) {
  if (p.src_u8) {
    const unsigned char* q = p.src_u8 + (((long)b * p.H + y) * p.W + x) * 3;
    v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2];
  } else {
    const long plane = (long)p.H * p.W;
    const float* q = p.src_f32 + (long)b * 3 * plane + (long)y * p.W + x;
    v[0] = q[0]; v[1] = q[plane]; v[2] = q[2 * plane];
  }
}

__device__ __forceinline__ void aug_sample(const AugArgs& p, int b, int oy, int ox, float (&rgb)[3]) {
  // the box is clamped to the image: a bad caller-supplied box can change the picture, never read outside it
  const int top = min(max(p.box[b * 4 + 0], 0), p.H - 1), left = min(max(p.box[b * 4 + 1], 0), p.W - 1);
  const int ch = min(max(p.box[b * 4 + 2], 1), p.H - top), cw = min(max(p.box[b * 4 + 3], 1), p.W - left);
  if (p.flip && p.flip[b]) ox = p.P - 1 - ox;   // flipping the output == flipping the crop before the resize
  const float sy = (float)ch / (float)p.P, sx = (float)cw / (float)p.P;
  float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int y0 = min((int)fy, ch - 1), x0 = min((int)fx, cw - 1);
  const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  float a[3], bb[3], c[3], d[3];
  aug_fetch(p, b, top + y0, left + x0, a); aug_fetch(p, b, top + y0, left + x1, bb);
  aug_fetch(p, b, top + y1, left + x0, c); aug_fetch(p, b, top + y1, left + x1, d);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t0 = a[k] + lx * (bb[k] - a[k]), t1 = c[k] + lx * (d[k] - c[k]);
    rgb[k] = (t0 + ly * (t1 - t0)) * p.scale;
  }
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(const float (&v)[3]) { return 0.2989f * v[0] + 0.587f * v[1] + 0.114f * v[2]; }

__global__ __launch_bounds__(256) void aug_resize_kernel(AugArgs p) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.P * p.P) return;
  const int oy = pix / p.P, ox = pix - oy * p.P;
  float v[3];
  aug_sample(p, b, oy, ox, v);
  if (p.jitter) {
    const float br = p.jitter[b * 4 + 0], ct = p.jitter[b * 4 + 1], sa = p.jitter[b * 4 + 2], gr = p.jitter[b * 4 + 3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = clamp01(v[k] * br);                                  // adjust_brightness: blend with 0
    const float m = p.gmean ? p.gmean[b] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = clamp01(ct * v[k] + (1.f - ct) * m);                 // adjust_contrast: blend with the mean grey
    const float g = gray_of(v);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = clamp01(sa * v[k] + (1.f - sa) * g);                 // adjust_saturation: blend with grey
    if (gr != 0.f) { const float g2 = gray_of(v); v[0] = v[1] = v[2] = g2; }                // RandomGrayscale / ToGray
  }
  const long plane = (long)p.P * p.P;
  float* o = p.out + (long)b * 3 * plane + pix;
  o[0] = clamp01(v[0]); o[plane] = clamp01(v[1]); o[2 * plane] = clamp01(v[2]);
}

// mean grey level of the brightness-adjusted SOURCE image, per sample (the constant adjust_contrast blends with; the
// reference jitters the whole image before it crops): one block per sample, fixed-order reduction (deterministic)
__global__ __launch_bounds__(256) void aug_gray_mean_kernel(AugArgs p, float* __restrict__ gmean) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float br = p.jitter ? p.jitter[b * 4 + 0] : 1.f;
  float s = 0.f;
  for (int pix = threadIdx.x; pix < p.H * p.W; pix += 256) {
    float v[3];
    aug_fetch(p, b, pix / p.W, pix % p.W, v);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = clamp01(v[k] * p.scale * br);
    s += gray_of(v);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) gmean[b] = s / (float)(p.H * p.W);
}

// ---------------------------------------------------------------- Gaussian blur (utils/utils.py:52, transforms.GaussianBlur(29, (0.1, 2.0)))
// Separable: one launch per direction.  Weights as torchvision's _get_gaussian_kernel1d: exp(-0.5 (x / sigma)^2) over
// x = -(k-1)/2 .. (k-1)/2, normalised; borders are mirrored without repeating the edge pixel ("reflect").  sigma[b] <= 0
// copies sample b unchanged (the transform is applied with probability 0.5).  Pass 0 reads the source (uint8 HWC or float
// CHW, scaled by `scale`) along rows, pass 1 reads the float CHW intermediate along columns.
__global__ __launch_bounds__(256) void aug_blur_kernel(AugArgs p, const float* __restrict__ sigma, const float* __restrict__ mid,
                                                        float* __restrict__ dst, int ksize, int vertical) {
  __shared__ float wgt[64];
  const int b = blockIdx.y, half = ksize >> 1;
  const float sg = sigma[b];
  if (threadIdx.x < ksize) {
    const float x = (float)((int)threadIdx.x - half);
    wgt[threadIdx.x] = sg > 0.f ? expf(-0.5f * (x / sg) * (x / sg)) : (threadIdx.x == half ? 1.f : 0.f);
  }
  __syncthreads();
  float wsum = 0.f;
  for (int t = 0; t < ksize; ++t) wsum += wgt[t];   // fixed order: every thread gets the same normaliser
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.H * p.W) return;
  const int y = pix / p.W, x0 = pix - y * p.W;
  const long plane = (long)p.H * p.W;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int t = 0; t < ksize; ++t) {
    int yy = y, xx = x0;
    if (vertical) { yy = y + t - half; if (yy < 0) yy = -yy; if (yy >= p.H) yy = 2 * p.H - 2 - yy; }
    else { xx = x0 + t - half; if (xx < 0) xx = -xx; if (xx >= p.W) xx = 2 * p.W - 2 - xx; }
    float v[3];
    if (vertical) {
      const float* q = mid + (long)b * 3 * plane + (long)yy * p.W + xx;
      v[0] = q[0]; v[1] = q[plane]; v[2] = q[2 * plane];
    } else {
      aug_fetch(p, b, yy, xx, v);
      v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale;
    }
    const float wt = wgt[t] / wsum;
    acc[0] += wt * v[0]; acc[1] += wt * v[1]; acc[2] += wt * v[2];
  }
  float* o = dst + (long)b * 3 * plane + pix;
  o[0] = acc[0]; o[plane] = acc[1]; o[2 * plane] = acc[2];
}

// ---------------------------------------------------------------- rotation (utils/utils.py:82,97,112,170: A.Rotate(limit) / RandomRotation)
// Whole source image rotated by angle[b] degrees (counter-clockwise, like cv2.getRotationMatrix2D) about its centre
// ((W-1)/2, (H-1)/2), same size: every output pixel samples the source bilinearly at its back-rotated position; positions
// outside the image are mirrored without repeating the edge pixel (cv2.BORDER_REFLECT_101, albumentations' default).
// angle 0 copies.  -> float CHW [B,3,H,W] scaled by `scale`.
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * n - 2;
  i = i % period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}

__global__ __launch_bounds__(256) void aug_rotate_kernel(AugArgs p, const float* __restrict__ angle, float* __restrict__ dst) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.H * p.W) return;
  const int y = pix / p.W, x = pix - y * p.W;
  const float th = angle[b] * 0.017453292519943295f;
  const float ca = cosf(th), sa = sinf(th);
  const float cx = 0.5f * (float)(p.W - 1), cy = 0.5f * (float)(p.H - 1);
  const float dx = (float)x - cx, dy = (float)y - cy;
  const float fx = ca * dx - sa * dy + cx, fy = sa * dx + ca * dy + cy;   // inverse of M = [[ca, sa], [-sa, ca]] about the centre
  const float flx = floorf(fx), fly = floorf(fy);
  const float lx = fx - flx, ly = fy - fly;
  const int x0 = reflect101((int)flx, p.W), x1 = reflect101((int)flx + 1, p.W);
  const int y0 = reflect101((int)fly, p.H), y1 = reflect101((int)fly + 1, p.H);
  float a[3], bb[3], c[3], d[3];
  aug_fetch(p, b, y0, x0, a); aug_fetch(p, b, y0, x1, bb); aug_fetch(p, b, y1, x0, c); aug_fetch(p, b, y1, x1, d);
  const long plane = (long)p.H * p.W;
  float* o = dst + (long)b * 3 * plane + pix;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t0 = a[k] + lx * (bb[k] - a[k]), t1 = c[k] + lx * (d[k] - c[k]);
    o[k * plane] = (t0 + ly * (t1 - t0)) * p.scale;
  }
}

// ---------------------------------------------------------------- hue (+ grayscale after it): ColorJitter(hue=0.1) of the strong views
// torchvision's float adjust_hue on CHW images in [0, 1], in place: RGB -> HSV, h <- (h + hue[b]) mod 1, HSV -> RGB; then
// (gray[b] != 0) the 3-channel grey image (RandomGrayscale / ToGray come after the colour jitter, utils/utils.py:225-226).
__global__ __launch_bounds__(256) void aug_hue_kernel(float* __restrict__ img, const float* __restrict__ hue, const float* __restrict__ gray,
                                                       int H, int W) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const long plane = (long)H * W;
  float* q = img + (long)b * 3 * plane + pix;
  float v[3] = {q[0], q[plane], q[2 * plane]};
  const float hf = hue ? hue[b] : 0.f;
  if (hf != 0.f) {
    const float r = v[0], g = v[1], bl = v[2];
    const float maxc = fmaxf(r, fmaxf(g, bl)), minc = fminf(r, fminf(g, bl));
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eq ? 1.f : maxc), dv = eq ? 1.f : cr;
    const float rc = (maxc - r) / dv, gc = (maxc - g) / dv, bc = (maxc - bl) / dv;
    float h = maxc == r ? (bc - gc) : (maxc == g ? 2.f + rc - bc : 4.f + gc - rc);
    h = fmodf(h / 6.f + 1.f, 1.f);
    h = h + hf;
    h = h - floorf(h);                                   // python's % 1.0
    const float h6 = h * 6.f;
    const float fi = floorf(h6), f = h6 - fi;
    const int i = ((int)fi) % 6;
    const float pp = clamp01(maxc * (1.f - s)), qq = clamp01(maxc * (1.f - f * s)), tt = clamp01(maxc * (1.f - (1.f - f) * s));
    switch (i) {
      case 0: v[0] = maxc; v[1] = tt; v[2] = pp; break;
      case 1: v[0] = qq; v[1] = maxc; v[2] = pp; break;
      case 2: v[0] = pp; v[1] = maxc; v[2] = tt; break;
      case 3: v[0] = pp; v[1] = qq; v[2] = maxc; break;
      case 4: v[0] = tt; v[1] = pp; v[2] = maxc; break;
      default: v[0] = maxc; v[1] = pp; v[2] = qq; break;
    }
  }
  if (gray && gray[b] != 0.f) { const float g2 = gray_of(v); v[0] = v[1] = v[2] = g2; }
  q[0] = v[0]; q[plane] = v[1]; q[2 * plane] = v[2];
}

// ---------------------------------------------------------------- C ABI
extern "C" int stil_tab_corrupt(const float* clean, const float* marginal, const int* idx, const int* pos, float* out,
                                int B, int n_cols, int n_rows, int k, void* stream) {
  STIL_REQUIRE(clean && marginal && out && B > 0 && n_cols > 0 && n_rows > 0, "stil_tab_corrupt: null pointer or empty shape");
  STIL_REQUIRE(k >= 0 && k <= n_cols && (k == 0 || (idx && pos)), "stil_tab_corrupt: k=%d outside [0, %d] or missing draws", k, n_cols);
  hipLaunchKernelGGL(tab_corrupt_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, clean, marginal, idx, pos, out, n_cols, n_rows, k);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_tab_corrupt_draw(int* idx, int* pos, int B, int n_cols, int n_rows, int k, unsigned long long seed,
                                     unsigned long long offset, const unsigned long long* step, void* stream) {
  STIL_REQUIRE(idx && pos && B > 0 && k > 0 && k <= n_cols && n_rows > 0, "stil_tab_corrupt_draw: bad arguments (k=%d, n_cols=%d)", k, n_cols);
  STIL_REQUIRE(n_cols <= 8192, "stil_tab_corrupt_draw: n_cols=%d > 8192", n_cols);
  hipLaunchKernelGGL(tab_corrupt_draw_kernel, dim3(B), dim3(64), n_cols * sizeof(int), (hipStream_t)stream, idx, pos, n_cols, n_rows, k,
                     seed, offset, step);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

static int aug_check(const AugArgs& p, int B) {
  STIL_REQUIRE((p.src_u8 != nullptr) != (p.src_f32 != nullptr), "stil_aug_resize: exactly one of src_u8 / src_f32");
  STIL_REQUIRE(p.box && p.out && B > 0 && p.H > 0 && p.W > 0 && p.P > 0, "stil_aug_resize: null pointer or empty shape");
  return STIL_OK;
}

extern "C" int stil_aug_gray_mean(const unsigned char* src_u8, const float* src_f32, const float* jitter, float* gmean, int B, int H,
                                  int W, float scale, void* stream) {
  AugArgs p{src_u8, src_f32, nullptr, nullptr, jitter, nullptr, nullptr, H, W, 1, scale};
  STIL_REQUIRE((src_u8 != nullptr) != (src_f32 != nullptr) && gmean && B > 0 && H > 0 && W > 0, "stil_aug_gray_mean: bad arguments");
  hipLaunchKernelGGL(aug_gray_mean_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p, gmean);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_aug_resize(const unsigned char* src_u8, const float* src_f32, const int* box, const unsigned char* flip,
                               const float* jitter, const float* gmean, float* out, int B, int H, int W, int P, float scale,
                               void* stream) {
  AugArgs p{src_u8, src_f32, box, flip, jitter, gmean, out, H, W, P, scale};
  int rc = aug_check(p, B);
  if (rc) return rc;
  STIL_REQUIRE(!jitter || gmean, "stil_aug_resize: colour jitter needs gmean (stil_aug_gray_mean)");
  hipLaunchKernelGGL(aug_resize_kernel, dim3(cdiv((long)P * P, 256), B), dim3(256), 0, (hipStream_t)stream, p);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_aug_blur(const unsigned char* src_u8, const float* src_f32, const float* sigma, float* tmp, float* out, int B, int H,
                             int W, int ksize, float scale, void* stream) {
  STIL_REQUIRE((src_u8 != nullptr) != (src_f32 != nullptr) && sigma && tmp && out && B > 0, "stil_aug_blur: bad arguments");
  STIL_REQUIRE(ksize % 2 == 1 && ksize >= 3 && ksize <= 63 && ksize / 2 < H && ksize / 2 < W,
               "stil_aug_blur: kernel size %d must be odd, in [3, 63] and its radius smaller than the image (%d x %d)", ksize, H, W);
  AugArgs p{src_u8, src_f32, nullptr, nullptr, nullptr, nullptr, nullptr, H, W, 1, scale};
  const dim3 grid(cdiv((long)H * W, 256), B);
  hipLaunchKernelGGL(aug_blur_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, sigma, (const float*)nullptr, tmp, ksize, 0);
  STIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(aug_blur_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, sigma, (const float*)tmp, out, ksize, 1);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_aug_rotate(const unsigned char* src_u8, const float* src_f32, const float* angle, float* out, int B, int H, int W,
                               float scale, void* stream) {
  STIL_REQUIRE((src_u8 != nullptr) != (src_f32 != nullptr) && angle && out && B > 0 && H > 0 && W > 0, "stil_aug_rotate: bad arguments");
  AugArgs p{src_u8, src_f32, nullptr, nullptr, nullptr, nullptr, nullptr, H, W, 1, scale};
  hipLaunchKernelGGL(aug_rotate_kernel, dim3(cdiv((long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, p, angle, out);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}

extern "C" int stil_aug_hue(float* img, const float* hue, const float* gray, int B, int H, int W, void* stream) {
  STIL_REQUIRE(img && (hue || gray) && B > 0 && H > 0 && W > 0, "stil_aug_hue: bad arguments");
  hipLaunchKernelGGL(aug_hue_kernel, dim3(cdiv((long)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, img, hue, gray, H, W);
  STIL_LAUNCH_CHECK();
  return STIL_OK;
}
