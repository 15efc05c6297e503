// HBM-bound kernels of the adaLN-single DiT path (Pixart-alpha) that the UNet path does not have:
// the gated residual, tanh-GELU and its derivative, and the per-sample column sums that give the gradients of the
// adaLN shift / scale / gate vectors.  All bf16 token-major [rows][C], 16-byte accesses, fp32 arithmetic.
#include "ops.h"
#include "dit_ops.h"

static inline int nblocks(int64_t total, int cap = 8192) {
  int64_t b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// y[m][c] = res[m][c] + gate[m / rows_per_batch][c] * x[m][c]   (res may be null: the backward's dy = gate * dout)
// res32 / y32 (round 5): the residual's fp32 master is read instead of `res`, and the fp32 result is stored beside the bf16 one --
// the transformer denoisers' residual stream stays in fp32 (gemm.h, GemmArgs::residual32)
__global__ __launch_bounds__(256) void gate_residual_kernel(const bf16_t* x, const bf16_t* gate, int64_t gate_ld,
                                                            const bf16_t* res, bf16_t* y, int64_t rows, int C,
                                                            int rows_per_batch, const float* res32, float* y32) {
  const int CPR = C >> 3;
  const int64_t total = rows * CPR;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / CPR;
    const int c0 = (int)(i - m * CPR) * 8;
    const u16x8 xv = *(const u16x8*)(x + m * C + c0);
    const u16x8 gv = *(const u16x8*)(gate + (m / rows_per_batch) * gate_ld + c0);
    u16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0}, o;
    if (res32) {   // (uniform)
      const float4 r0 = *(const float4*)(res32 + m * C + c0), r1 = *(const float4*)(res32 + m * C + c0 + 4);
      const float rf[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = fmaf(bf2f(gv[e]), bf2f(xv[e]), rf[e]);
        o[e] = f2bf(v[e]);
      }
      if (y32) {
        *(float4*)(y32 + m * C + c0) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(y32 + m * C + c0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      *(u16x8*)(y + m * C + c0) = o;
      continue;
    }
    if (res) rv = *(const u16x8*)(res + m * C + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaf(bf2f(gv[e]), bf2f(xv[e]), bf2f(rv[e])));
    *(u16x8*)(y + m * C + c0) = o;
  }
}

// tanh-approximated GELU (torch.nn.GELU(approximate="tanh")): 0.5 x (1 + tanh(k (x + 0.044715 x^3))), k = sqrt(2/pi)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
  return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float dgelu_tanh_f(float x) {
  const float x2 = x * x;
  const float t = tanhf(0.7978845608028654f * fmaf(0.044715f * x2, x, x));
  const float du = 0.7978845608028654f * fmaf(3.f * 0.044715f, x2, 1.f);
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
}
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_tanh_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u16x8 xv = *(const u16x8*)(x + i * 8);
    u16x8 o, dv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (BWD) dv = *(const u16x8*)(dy + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = bf2f(xv[e]);
      o[e] = f2bf(BWD ? bf2f(dv[e]) * dgelu_tanh_f(v) : gelu_tanh_f(v));
    }
    *(u16x8*)(out + i * 8) = o;
  }
}

// Per-sample column sums over the rows of each batch entry (outputs pre-zeroed, fp32 atomics):
//   out1[b][c] += sum_r dy[r][c]                    (d shift)
//   out0[b][c] += sum_r dy[r][c] * f(x[r][c])       (d scale with f = LayerNorm normalisation from stats[r] = (mean, rstd);
//                                                    d gate with f = identity when stats is null)
// grid (row chunks of RCH, B, ceil(C/8/256)); a thread owns one 8-channel chunk and walks the rows of its chunk.
constexpr int BCS_RCH = 64;
__global__ __launch_bounds__(256) void batch_colsum_kernel(const bf16_t* dy, const bf16_t* x, const float* stats,
                                                           float* out0, float* out1, int rows_per_batch, int C, int rch) {
  const int ch = blockIdx.z * 256 + threadIdx.x;
  if (ch >= (C >> 3)) return;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rch;
  const int r1 = min(r0 + rch, rows_per_batch);
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  for (int r = r0; r < r1; ++r) {
    const int64_t row = (int64_t)b * rows_per_batch + r;
    const u16x8 dv = *(const u16x8*)(dy + row * C + ch * 8);
    float mean = 0.f, rstd = 1.f;
    if (stats) {
      mean = stats[row * 2];
      rstd = stats[row * 2 + 1];
    }
    if (x) {
      const u16x8 xv = *(const u16x8*)(x + row * C + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s0[e] = fmaf(bf2f(dv[e]), (bf2f(xv[e]) - mean) * rstd, s0[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] += bf2f(dv[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (out0) atomicAdd(out0 + (int64_t)b * C + ch * 8 + e, s0[e]);
    if (out1) atomicAdd(out1 + (int64_t)b * C + ch * 8 + e, s1[e]);
  }
}

int launch_gate_residual(const bf16_t* x, const bf16_t* gate, int64_t gate_ld, const bf16_t* res, bf16_t* y,
                         int64_t rows, int C, int rows_per_batch, hipStream_t st, const float* res32, float* y32) {
  FDMI_CHECK(C % 8 == 0 && gate_ld % 8 == 0 && rows_per_batch > 0, "gate_residual: C and gate_ld must be multiples of 8");
  FDMI_CHECK(!y32 || res32, "gate_residual: an fp32 output needs the fp32 residual");
  hipLaunchKernelGGL(gate_residual_kernel, dim3(nblocks(rows * (C >> 3))), dim3(256), 0, st, x, gate, gate_ld, res, y,
                     rows, C, rows_per_batch, res32, y32);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_gelu_tanh(const bf16_t* x, bf16_t* y, int64_t n, hipStream_t st) {
  FDMI_CHECK(n % 8 == 0, "gelu_tanh: element count must be a multiple of 8");
  hipLaunchKernelGGL(gelu_tanh_kernel<false>, dim3(nblocks(n >> 3)), dim3(256), 0, st, x, (const bf16_t*)nullptr, y,
                     n >> 3);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_gelu_tanh_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n, hipStream_t st) {
  FDMI_CHECK(n % 8 == 0, "gelu_tanh_bwd: element count must be a multiple of 8");
  hipLaunchKernelGGL(gelu_tanh_kernel<true>, dim3(nblocks(n >> 3)), dim3(256), 0, st, x, dy, dx, n >> 3);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_batch_colsum(const bf16_t* dy, const bf16_t* x, const float* stats, float* out0, float* out1, int B,
                        int rows_per_batch, int C, hipStream_t st) {
  FDMI_CHECK(C % 8 == 0 && B > 0 && rows_per_batch > 0, "batch_colsum: C must be a multiple of 8");
  FDMI_CHECK(!out0 || x, "batch_colsum: out0 needs x");
  if (out0) FDMI_HIP(hipMemsetAsync(out0, 0, (size_t)B * C * sizeof(float), st));
  if (out1) FDMI_HIP(hipMemsetAsync(out1, 0, (size_t)B * C * sizeof(float), st));
  const int rch = fdmi_det() ? rows_per_batch : BCS_RCH;   // deterministic mode: one block per (sample, column chunk) = one contributor
  hipLaunchKernelGGL(batch_colsum_kernel, dim3(cdiv(rows_per_batch, rch), B, cdiv(C >> 3, 256)), dim3(256), 0, st, dy,
                     x, stats, out0, out1, rows_per_batch, C, rch);
  FDMI_HIP(hipGetLastError());
  return 0;
}

// ---- layout / per-sample-vector kernels of the transformer-denoiser PLANS (unet.hip: fdmi_dit_*) ----------------------------
// TO = bf16_t (the measured plans) or float (the fp32 validation plans); small tensors or pure data movement, one element per lane.
namespace {
template <typename TO> __device__ __forceinline__ TO cvt_out(float v);
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
__device__ __forceinline__ float cvt_in(bf16_t v) { return bf2f(v); }
__device__ __forceinline__ float cvt_in(float v) { return v; }

// x [B][C][H][W] f32 <-> patches [B * h * w][C * p * p], column = (c, py, px): the patch-embedding convolution (kernel = stride
// = p) as a linear map.  BWD: gx = the same map read backwards (every input element belongs to exactly one patch).
template <typename TO, bool BWD>
__global__ __launch_bounds__(256) void patchify_kernel(const float* x, TO* pt, float* gx, int B, int C, int H, int W, int p) {
  const int h = H / p, w = W / p, cols = C * p * p;
  const int64_t total = (int64_t)B * h * w * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % cols);
    const int64_t tok = i / cols;
    const int px = col % p, py = (col / p) % p, c = col / (p * p);
    const int wx = (int)(tok % w), hy = (int)((tok / w) % h), b = (int)(tok / ((int64_t)h * w));
    const int64_t xi = (((int64_t)b * C + c) * H + hy * p + py) * W + wx * p + px;
    if (BWD) gx[xi] = cvt_in(pt[i]);
    else pt[i] = cvt_out<TO>(x[xi]);
  }
}
// tokens [B * h * w][p * p * oc], column = (py, px, c)  <->  images [B][keep][H][W] f32 (the first `keep` of the oc channels:
// the wrappers drop a learned-variance half, tranformers.py:91 / :154).  BWD: d tokens from d images, zero in the dropped channels.
template <typename TO, bool BWD>
__global__ __launch_bounds__(256) void unpatchify_kernel(TO* tk, float* img, int B, int oc, int keep, int H, int W, int p) {
  const int h = H / p, w = W / p, cols = p * p * oc;
  const int64_t total = (int64_t)B * h * w * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % cols);
    const int64_t tok = i / cols;
    const int c = col % oc, px = (col / oc) % p, py = col / (oc * p);
    const int wx = (int)(tok % w), hy = (int)((tok / w) % h), b = (int)(tok / ((int64_t)h * w));
    const int64_t xi = (((int64_t)b * keep + c) * H + hy * p + py) * W + wx * p + px;
    if (BWD) tk[i] = cvt_out<TO>(c < keep ? img[xi] : 0.f);
    else if (c < keep) img[xi] = cvt_in(tk[i]);
  }
}
// dst[b][i] = table[i] + src[b][i % src_cols]   (a scale-shift table plus the per-sample modulation vectors, in fp32, one rounding)
template <typename TO>
__global__ __launch_bounds__(256) void add_table_kernel(const TO* src, int src_cols, const float* table, TO* dst, int B, int n) {
  const int64_t total = (int64_t)B * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % n);
    const int64_t b = i / n;
    dst[i] = cvt_out<TO>(table[c] + cvt_in(src[b * src_cols + c % src_cols]));
  }
}
// dst[b][col0 + c] (+)= src[b][c]   (fp32 column sums into the gradient of a modulation tensor)
template <typename TO>
__global__ __launch_bounds__(256) void vec_grad_add_kernel(const float* src, TO* dst, int64_t ldd, int col0, int B, int C, int accumulate) {
  const int64_t total = (int64_t)B * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t b = i / C;
    TO* d = dst + b * ldd + col0 + c;
    *d = cvt_out<TO>(src[i] + (accumulate ? cvt_in(*d) : 0.f));
  }
}
// KV[b * (T + L) + s][0 .. 2D) = columns [D, 3D) of yx[b * T + s] (s < T) or of yc[b * L + s - T]: the keys and values of the MMDiT's
// joint [latent | text] sequence out of the two streams' fused [q | k | v] projections (the queries stay where they are)
__global__ __launch_bounds__(256) void kv_join_kernel(const bf16_t* yx, const bf16_t* yc, bf16_t* kv, int B, int T, int L, int D) {
  const int S = T + L, cpr = (2 * D) >> 3;
  const int64_t total = (int64_t)B * S * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cpr) * 8;
    const int64_t row = i / cpr;
    const int s = (int)(row % S);
    const int64_t b = row / S;
    const bf16_t* src = s < T ? yx + (b * T + s) * 3 * D : yc + (b * L + (s - T)) * 3 * D;
    *(u16x8*)(kv + row * 2 * D + c) = *(const u16x8*)(src + D + c);
  }
}
}  // namespace

#define DIT_LAUNCH(kernel, total, ...)                                                              \
  hipLaunchKernelGGL(kernel, dim3(nblocks(total)), dim3(256), 0, st, __VA_ARGS__);                  \
  FDMI_HIP(hipGetLastError());                                                                      \
  return 0;

int launch_patchify(const float* x, void* patches, int B, int C, int H, int W, int p, int f32, hipStream_t st) {
  FDMI_CHECK(p > 0 && H % p == 0 && W % p == 0, "patchify: H, W must be multiples of the patch size");
  const int64_t total = (int64_t)B * C * H * W;
  if (f32) { DIT_LAUNCH((patchify_kernel<float, false>), total, x, (float*)patches, (float*)nullptr, B, C, H, W, p) }
  DIT_LAUNCH((patchify_kernel<bf16_t, false>), total, x, (bf16_t*)patches, (float*)nullptr, B, C, H, W, p)
}
int launch_patchify_bwd(const void* dpatches, float* gx, int B, int C, int H, int W, int p, int f32, hipStream_t st) {
  const int64_t total = (int64_t)B * C * H * W;
  if (f32) { DIT_LAUNCH((patchify_kernel<float, true>), total, (const float*)nullptr, (float*)dpatches, gx, B, C, H, W, p) }
  DIT_LAUNCH((patchify_kernel<bf16_t, true>), total, (const float*)nullptr, (bf16_t*)dpatches, gx, B, C, H, W, p)
}
int launch_unpatchify(const void* tokens, float* img, int B, int oc, int keep, int H, int W, int p, int f32, hipStream_t st) {
  FDMI_CHECK(p > 0 && H % p == 0 && W % p == 0 && keep > 0 && keep <= oc, "unpatchify: bad geometry");
  const int64_t total = (int64_t)B * oc * H * W;
  if (f32) { DIT_LAUNCH((unpatchify_kernel<float, false>), total, (float*)tokens, img, B, oc, keep, H, W, p) }
  DIT_LAUNCH((unpatchify_kernel<bf16_t, false>), total, (bf16_t*)tokens, img, B, oc, keep, H, W, p)
}
int launch_unpatchify_bwd(const float* gimg, void* dtokens, int B, int oc, int keep, int H, int W, int p, int f32, hipStream_t st) {
  const int64_t total = (int64_t)B * oc * H * W;
  if (f32) { DIT_LAUNCH((unpatchify_kernel<float, true>), total, (float*)dtokens, (float*)gimg, B, oc, keep, H, W, p) }
  DIT_LAUNCH((unpatchify_kernel<bf16_t, true>), total, (bf16_t*)dtokens, (float*)gimg, B, oc, keep, H, W, p)
}
int launch_add_table(const void* src, int src_cols, const float* table, void* dst, int B, int n, int f32, hipStream_t st) {
  FDMI_CHECK(src_cols > 0 && n % src_cols == 0, "add_table: the table width must be a multiple of the vector width");
  if (f32) { DIT_LAUNCH(add_table_kernel<float>, (int64_t)B * n, (const float*)src, src_cols, table, (float*)dst, B, n) }
  DIT_LAUNCH(add_table_kernel<bf16_t>, (int64_t)B * n, (const bf16_t*)src, src_cols, table, (bf16_t*)dst, B, n)
}
int launch_vec_grad_add(const float* src, void* dst, int64_t ldd, int col0, int B, int C, int accumulate, int f32, hipStream_t st) {
  if (f32) { DIT_LAUNCH(vec_grad_add_kernel<float>, (int64_t)B * C, src, (float*)dst, ldd, col0, B, C, accumulate) }
  DIT_LAUNCH(vec_grad_add_kernel<bf16_t>, (int64_t)B * C, src, (bf16_t*)dst, ldd, col0, B, C, accumulate)
}
int launch_kv_join(const bf16_t* yx, const bf16_t* yc, bf16_t* kv, int B, int T, int L, int D, hipStream_t st) {
  FDMI_CHECK(D % 8 == 0, "kv_join: D must be a multiple of 8");
  DIT_LAUNCH(kv_join_kernel, (int64_t)B * (T + L) * (2 * D / 8), yx, yc, kv, B, T, L, D)
}
