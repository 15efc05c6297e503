// Element-wise / pooling / perceptual-distance kernels of the frozen convolutional networks that sit beside the denoiser in
// the distillation step: the VGG16 feature stack of LPIPS (ReLU mask, 2x2 max pooling, the
// channel-normalised feature distance of lpips.LPIPS.forward, lpips==0.1.4, setup.py:40), the T2I-adapter's pixel unshuffle
// and average pooling, and the bf16 <-> fp32 staging of the one wide-head attention of the VAE decoder's mid block.
// Every kernel is templated on the storage type: bf16 (the measured path) and float (the fp32 validation plans).
// HBM-bound: 16-byte (bf16) accesses where the layout allows, one wave per pixel row for the distance kernels.
#include "ops.h"

namespace {

#define NGRID_STRIDE(i, total) \
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (total); i += (int64_t)gridDim.x * 256)
static inline int nblk(int64_t total, int cap = 16384) {
  int64_t b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
__device__ __forceinline__ float ldv(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ldv(const float* p) { return *p; }
__device__ __forceinline__ void stv(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }

// dy <- dy * (y > 0): the ReLU behind a convolution whose epilogue applied it (y is the stored, post-ReLU output)
template <typename TT>
__global__ __launch_bounds__(256) void relu_mask_kernel(const TT* y, TT* dy, int64_t n) {
  NGRID_STRIDE(i, n) {
    if (!(ldv(y + i) > 0.f)) stv(dy + i, 0.f);
  }
}

// 2x2 / stride-2 max pooling on NHWC (H, W even): y[b, h, w, c] = max over the window
template <typename TT>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const TT* x, TT* y, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  NGRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const TT* s = x + (((int64_t)b * H + 2 * ho) * W + 2 * wo) * C + c;
    const float v = fmaxf(fmaxf(ldv(s), ldv(s + C)), fmaxf(ldv(s + (int64_t)W * C), ldv(s + (int64_t)W * C + C)));
    stv(y + i, v);
  }
}
// its input gradient: dy goes to the FIRST window element (row-major) that equals the maximum -- torch's choice on ties;
// the other three positions receive zero (accumulate: dx += ...)
template <typename TT>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const TT* x, const TT* dy, TT* dx, int B, int H, int W, int C,
                                                           int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  NGRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int64_t base = (((int64_t)b * H + 2 * ho) * W + 2 * wo) * C + c;
    const int64_t off[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ldv(x + base + off[k]);
    int am = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (v[k] > v[am]) am = k;
    const float g = ldv(dy + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float add = k == am ? g : 0.f;
      stv(dx + base + off[k], accumulate ? ldv(dx + base + off[k]) + add : add);
    }
  }
}

// 2x2 average pooling forward (the T2I-adapter's Downsample2D(use_conv=False) = AvgPool2d(2))
template <typename TT>
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const TT* x, TT* y, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  NGRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const TT* s = x + (((int64_t)b * H + 2 * ho) * W + 2 * wo) * C + c;
    stv(y + i, 0.25f * (ldv(s) + ldv(s + C) + ldv(s + (int64_t)W * C) + ldv(s + (int64_t)W * C + C)));
  }
}

// NCHW f32 image -> NHWC [B, H/f, W/f, Cpad] with torch.nn.PixelUnshuffle(f) channel order (c * f * f + dy * f + dx), zero padded
template <typename TT>
__global__ __launch_bounds__(256) void pixel_unshuffle_kernel(const float* x, TT* y, int B, int C, int H, int W, int f, int Cpad) {
  const int Ho = H / f, Wo = W / f;
  const int64_t total = (int64_t)B * Ho * Wo * Cpad;
  NGRID_STRIDE(i, total) {
    const int cc = (int)(i % Cpad);
    int64_t p = i / Cpad;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float v = 0.f;
    if (cc < C * f * f) {
      const int c = cc / (f * f), r = cc - c * f * f, dy = r / f, dx = r - dy * f;
      v = x[(((int64_t)b * C + c) * H + ho * f + dy) * W + wo * f + dx];
    }
    stv(y + i, v);
  }
}

// NHWC [rows][C] (storage type) -> NCHW f32 WITHOUT the bf16 plan's layout assumptions (any C): the adapter's feature maps
template <typename TT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_any_kernel(const TT* x, float* y, int B, int C, int HW) {
  const int64_t total = (int64_t)B * C * HW;
  NGRID_STRIDE(i, total) {
    const int s = (int)(i % HW);
    int64_t p = i / HW;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    y[i] = ldv(x + ((int64_t)b * HW + s) * C + c);
  }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* x, float* y, int64_t n) {
  NGRID_STRIDE(i, n) y[i] = bf2f(x[i]);
}

// ---- LPIPS level (lpips.LPIPS.forward, lpips=True, spatial=False): per pixel row the two feature vectors are divided by
// (their channel L2 norm + 1e-10), the squared difference is weighted by the non-negative 1x1 "lin" layer and averaged over
// the pixels of the sample:  out[b] += (1 / HW) * sum_rows sum_c w[c] * (n0[c] - n1[c])^2.
// One wave per row; lane l owns channels l, l + 64, ... (C <= 512: at most 8 per lane, kept in registers).
// (row term shared by the production kernel and its deterministic twin: the wave's sum over channels of w (n0 - n1)^2)
template <typename TT>
__device__ __forceinline__ float lpips_row_term(const TT* f0, const TT* f1, const float* w, int64_t row, int C, int lane) {
  float a[8], b[8];
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    a[k] = c < C ? ldv(f0 + row * C + c) : 0.f;
    b[k] = c < C ? ldv(f1 + row * C + c) : 0.f;
    sa = fmaf(a[k], a[k], sa);
    sb = fmaf(b[k], b[k], sb);
  }
  sa = wave_sum(sa);
  sb = wave_sum(sb);
  const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    if (c < C) {
      const float d = a[k] * ia - b[k] * ib;
      acc = fmaf(w[c] * d, d, acc);
    }
  }
  return wave_sum(acc);
}
template <typename TT>
__global__ __launch_bounds__(256) void lpips_level_fwd_kernel(const TT* f0, const TT* f1, const float* w, float* out, int64_t rows,
                                                              int HW, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float acc = lpips_row_term(f0, f1, w, row, C, lane);
  if (lane == 0) atomicAdd(out + row / HW, acc / (float)HW);
}
// deterministic twin (fdmi_det(), common.h): one block per sample, wave v takes rows v, v + 4, ... in order, the four waves' sums are
// added in wave order by the sample's only writer
template <typename TT>
__global__ __launch_bounds__(256) void lpips_level_fwd_det_kernel(const TT* f0, const TT* f1, const float* w, float* out, int HW, int C) {
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  float tot = 0.f;
  for (int r = wave; r < HW; r += 4) tot += lpips_row_term(f0, f1, w, (int64_t)b * HW + r, C, lane) / (float)HW;
  if (lane == 0) part[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) out[b] += ((part[0] + part[1]) + part[2]) + part[3];
}
// gradient wrt f0 (the student's features; f1 is the no-grad teacher side), accumulated into df0 (the feature tensor also feeds
// the next VGG stage): with n0 = f0 / (s + eps), s = |f0|, g = gout[b] / HW * 2 w (n0 - n1):
//   df0 = g / (s + eps) - f0 * (f0 . g) / (s (s + eps)^2)
template <typename TT>
__global__ __launch_bounds__(256) void lpips_level_bwd_kernel(const TT* f0, const TT* f1, const float* w, const float* gout, TT* df0,
                                                              int64_t rows, int HW, int C, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float a[8], b[8], g[8];
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    a[k] = c < C ? ldv(f0 + row * C + c) : 0.f;
    b[k] = c < C ? ldv(f1 + row * C + c) : 0.f;
    sa = fmaf(a[k], a[k], sa);
    sb = fmaf(b[k], b[k], sb);
  }
  sa = wave_sum(sa);
  sb = wave_sum(sb);
  const float s = sqrtf(sa), ia = 1.f / (s + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
  const float go = gout[row / HW] / (float)HW;
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    g[k] = c < C ? go * 2.f * w[c] * (a[k] * ia - b[k] * ib) : 0.f;
    dot = fmaf(a[k], g[k], dot);
  }
  dot = wave_sum(dot);
  const float k2 = s > 0.f ? dot * ia * ia / s : 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    if (c < C) {
      const float d = g[k] * ia - a[k] * k2;
      stv(df0 + row * C + c, accumulate ? ldv(df0 + row * C + c) + d : d);
    }
  }
}

// LPIPS ScalingLayer on the way in: NCHW f32 image in [-1, 1] -> NHWC (3 channels padded to Cpad), (x - shift[c]) / scale[c]
template <typename TT>
__global__ __launch_bounds__(256) void lpips_input_kernel(const float* x, TT* y, int B, int HW, int Cpad, float s0, float s1, float s2,
                                                          float q0, float q1, float q2) {
  const int64_t total = (int64_t)B * HW * Cpad;
  NGRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const int64_t p = i / Cpad;
    const int b = (int)(p / HW);
    const int s = (int)(p - (int64_t)b * HW);
    float v = 0.f;
    if (c < 3) {
      const float sh = c == 0 ? s0 : (c == 1 ? s1 : s2), sc = c == 0 ? q0 : (c == 1 ? q1 : q2);
      v = (x[((int64_t)b * 3 + c) * HW + s] - sh) / sc;
    }
    stv(y + i, v);
  }
}
// ... and its gradient: dx[b, c, s] = dy[b, s, c] / scale[c]
template <typename TT>
__global__ __launch_bounds__(256) void lpips_input_bwd_kernel(const TT* dy, float* dx, int B, int HW, int Cpad, float q0, float q1, float q2) {
  const int64_t total = (int64_t)B * 3 * HW;
  NGRID_STRIDE(i, total) {
    const int s = (int)(i % HW);
    int64_t p = i / HW;
    const int c = (int)(p % 3);
    const int b = (int)(p / 3);
    const float sc = c == 0 ? q0 : (c == 1 ? q1 : q2);
    dx[i] = ldv(dy + ((int64_t)b * HW + s) * Cpad + c) / sc;
  }
}

// T5LayerNorm (transformers T5LayerNorm, the T5 text encoder of the PixArt / SD3 conditioners): y = x * rsqrt(mean(x^2) + eps) * w --
// no mean subtraction, no bias; statistics in fp32.  One wave per row.
template <typename TT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const TT* x, const float* w, TT* y, int64_t rows, int C, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const TT* xr = x + row * C;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = ldv(xr + c);
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)C + eps);
  TT* yr = y + row * C;
  for (int c = lane; c < C; c += 64) stv(yr + c, ldv(xr + c) * r * w[c]);
}
// y = a * b (the gate of T5's gated feed-forward: gelu_new(x Wi0^T) * (x Wi1^T))
template <typename TT>
__global__ __launch_bounds__(256) void ewise_mul_kernel(const TT* a, const TT* b, TT* y, int64_t n) {
  NGRID_STRIDE(i, n) stv(y + i, ldv(a + i) * ldv(b + i));
}

}  // namespace

#define NLAUNCH(kernel, total, ...)                                                     \
  do {                                                                                  \
    hipLaunchKernelGGL(kernel, dim3(nblk(total)), dim3(256), 0, st, __VA_ARGS__);       \
    FDMI_HIP(hipGetLastError());                                                        \
    return 0;                                                                           \
  } while (0)

int launch_relu_mask(const bf16_t* y, bf16_t* dy, int64_t n, hipStream_t st) { NLAUNCH(relu_mask_kernel<bf16_t>, n, y, dy, n); }
int launch_relu_mask32(const float* y, float* dy, int64_t n, hipStream_t st) { NLAUNCH(relu_mask_kernel<float>, n, y, dy, n); }
int launch_maxpool2_fwd(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, hipStream_t st) {
  FDMI_CHECK((H & 1) == 0 && (W & 1) == 0, "maxpool2: even H, W");
  NLAUNCH(maxpool2_fwd_kernel<bf16_t>, (int64_t)B * (H / 2) * (W / 2) * C, x, y, B, H, W, C);
}
int launch_maxpool2_fwd32(const float* x, float* y, int B, int H, int W, int C, hipStream_t st) {
  FDMI_CHECK((H & 1) == 0 && (W & 1) == 0, "maxpool2: even H, W");
  NLAUNCH(maxpool2_fwd_kernel<float>, (int64_t)B * (H / 2) * (W / 2) * C, x, y, B, H, W, C);
}
int launch_maxpool2_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C, int accumulate, hipStream_t st) {
  NLAUNCH(maxpool2_bwd_kernel<bf16_t>, (int64_t)B * (H / 2) * (W / 2) * C, x, dy, dx, B, H, W, C, accumulate);
}
int launch_maxpool2_bwd32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int accumulate, hipStream_t st) {
  NLAUNCH(maxpool2_bwd_kernel<float>, (int64_t)B * (H / 2) * (W / 2) * C, x, dy, dx, B, H, W, C, accumulate);
}
int launch_avgpool2_fwd(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, hipStream_t st) {
  NLAUNCH(avgpool2_fwd_kernel<bf16_t>, (int64_t)B * (H / 2) * (W / 2) * C, x, y, B, H, W, C);
}
int launch_avgpool2_fwd32(const float* x, float* y, int B, int H, int W, int C, hipStream_t st) {
  NLAUNCH(avgpool2_fwd_kernel<float>, (int64_t)B * (H / 2) * (W / 2) * C, x, y, B, H, W, C);
}
int launch_pixel_unshuffle(const float* x, bf16_t* y, int B, int C, int H, int W, int f, int Cpad, hipStream_t st) {
  FDMI_CHECK(f > 0 && H % f == 0 && W % f == 0 && Cpad >= C * f * f, "pixel_unshuffle: H, W must be multiples of the factor");
  NLAUNCH(pixel_unshuffle_kernel<bf16_t>, (int64_t)B * (H / f) * (W / f) * Cpad, x, y, B, C, H, W, f, Cpad);
}
int launch_pixel_unshuffle32(const float* x, float* y, int B, int C, int H, int W, int f, int Cpad, hipStream_t st) {
  FDMI_CHECK(f > 0 && H % f == 0 && W % f == 0 && Cpad >= C * f * f, "pixel_unshuffle: H, W must be multiples of the factor");
  NLAUNCH(pixel_unshuffle_kernel<float>, (int64_t)B * (H / f) * (W / f) * Cpad, x, y, B, C, H, W, f, Cpad);
}
int launch_nhwc_to_nchw_any(const bf16_t* x, float* y, int B, int C, int HW, hipStream_t st) {
  NLAUNCH(nhwc_to_nchw_any_kernel<bf16_t>, (int64_t)B * C * HW, x, y, B, C, HW);
}
int launch_nhwc_to_nchw_any32(const float* x, float* y, int B, int C, int HW, hipStream_t st) {
  NLAUNCH(nhwc_to_nchw_any_kernel<float>, (int64_t)B * C * HW, x, y, B, C, HW);
}
int launch_bf16_to_f32(const bf16_t* x, float* y, int64_t n, hipStream_t st) { NLAUNCH(bf16_to_f32_kernel, n, x, y, n); }

int launch_lpips_level_fwd(const bf16_t* f0, const bf16_t* f1, const float* w, float* out, int64_t rows, int HW, int C, hipStream_t st) {
  FDMI_CHECK(C <= 512, "lpips_level: at most 512 channels");
  if (fdmi_det()) hipLaunchKernelGGL(lpips_level_fwd_det_kernel<bf16_t>, dim3((unsigned)(rows / HW)), dim3(256), 0, st, f0, f1, w, out, HW, C);
  else hipLaunchKernelGGL(lpips_level_fwd_kernel<bf16_t>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, f0, f1, w, out, rows, HW, C);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_lpips_level_fwd32(const float* f0, const float* f1, const float* w, float* out, int64_t rows, int HW, int C, hipStream_t st) {
  FDMI_CHECK(C <= 512, "lpips_level: at most 512 channels");
  if (fdmi_det()) hipLaunchKernelGGL(lpips_level_fwd_det_kernel<float>, dim3((unsigned)(rows / HW)), dim3(256), 0, st, f0, f1, w, out, HW, C);
  else hipLaunchKernelGGL(lpips_level_fwd_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, f0, f1, w, out, rows, HW, C);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_lpips_level_bwd(const bf16_t* f0, const bf16_t* f1, const float* w, const float* gout, bf16_t* df0, int64_t rows, int HW,
                           int C, int accumulate, hipStream_t st) {
  FDMI_CHECK(C <= 512, "lpips_level: at most 512 channels");
  hipLaunchKernelGGL(lpips_level_bwd_kernel<bf16_t>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, f0, f1, w, gout, df0, rows, HW,
                     C, accumulate);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_lpips_level_bwd32(const float* f0, const float* f1, const float* w, const float* gout, float* df0, int64_t rows, int HW,
                             int C, int accumulate, hipStream_t st) {
  FDMI_CHECK(C <= 512, "lpips_level: at most 512 channels");
  hipLaunchKernelGGL(lpips_level_bwd_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, f0, f1, w, gout, df0, rows, HW,
                     C, accumulate);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_lpips_input(const float* x, bf16_t* y, int B, int HW, int Cpad, const float* shift, const float* scale, hipStream_t st) {
  NLAUNCH(lpips_input_kernel<bf16_t>, (int64_t)B * HW * Cpad, x, y, B, HW, Cpad, shift[0], shift[1], shift[2], scale[0], scale[1], scale[2]);
}
int launch_lpips_input32(const float* x, float* y, int B, int HW, int Cpad, const float* shift, const float* scale, hipStream_t st) {
  NLAUNCH(lpips_input_kernel<float>, (int64_t)B * HW * Cpad, x, y, B, HW, Cpad, shift[0], shift[1], shift[2], scale[0], scale[1], scale[2]);
}
int launch_lpips_input_bwd(const bf16_t* dy, float* dx, int B, int HW, int Cpad, const float* scale, hipStream_t st) {
  NLAUNCH(lpips_input_bwd_kernel<bf16_t>, (int64_t)B * 3 * HW, dy, dx, B, HW, Cpad, scale[0], scale[1], scale[2]);
}
int launch_lpips_input_bwd32(const float* dy, float* dx, int B, int HW, int Cpad, const float* scale, hipStream_t st) {
  NLAUNCH(lpips_input_bwd_kernel<float>, (int64_t)B * 3 * HW, dy, dx, B, HW, Cpad, scale[0], scale[1], scale[2]);
}
int launch_rmsnorm(const bf16_t* x, const float* w, bf16_t* y, int64_t rows, int C, float eps, hipStream_t st) {
  hipLaunchKernelGGL(rmsnorm_kernel<bf16_t>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, w, y, rows, C, eps);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_rmsnorm32(const float* x, const float* w, float* y, int64_t rows, int C, float eps, hipStream_t st) {
  hipLaunchKernelGGL(rmsnorm_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, w, y, rows, C, eps);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_ewise_mul(const bf16_t* a, const bf16_t* b, bf16_t* y, int64_t n, hipStream_t st) { NLAUNCH(ewise_mul_kernel<bf16_t>, n, a, b, y, n); }
int launch_ewise_mul32(const float* a, const float* b, float* y, int64_t n, hipStream_t st) { NLAUNCH(ewise_mul_kernel<float>, n, a, b, y, n); }
