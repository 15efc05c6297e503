// HBM-bound element-wise / layout kernels around the MFMA ops (16-byte vector accesses where the
// layout allows), the GEGLU backward, the LoRA refresh (cast + transpose) and fused AdamW.
#include "ops.h"

#define GRID_STRIDE(i, total) \
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (total); i += (int64_t)gridDim.x * 256)

static inline int nblocks(int64_t total, int cap = 8192) {
  int64_t b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// latents: NCHW f32 -> NHWC bf16 with the channel dim zero-padded to Cpad (multiple of 8)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, bf16_t* y, int B, int C, int HW,
                                                           int Cpad) {
  const int64_t total = (int64_t)B * HW * Cpad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const int64_t p = i / Cpad;
    const int b = (int)(p / HW);
    const int s = (int)(p - (int64_t)b * HW);
    y[i] = c < C ? f2bf(x[((int64_t)b * C + c) * HW + s]) : (bf16_t)0;
  }
}
// NHWC bf16 (row stride ldx) -> NCHW f32
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16_t* x, int64_t ldx, float* y, int B, int C,
                                                           int HW, int accumulate) {
  const int64_t total = (int64_t)B * C * HW;
  GRID_STRIDE(i, total) {
    const int s = (int)(i % HW);
    const int64_t bc = i / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const float v = bf2f(x[((int64_t)b * HW + s) * ldx + c]);
    y[i] = accumulate ? y[i] + v : v;
  }
}
// NCHW f32 gradient -> NHWC bf16 rows of width ldy (columns >= C zeroed)
__global__ __launch_bounds__(256) void nchw_grad_to_nhwc_kernel(const float* g, bf16_t* y, int64_t ldy, int B,
                                                                int C, int HW) {
  const int64_t total = (int64_t)B * HW * ldy;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % ldy);
    const int64_t p = i / ldy;
    const int b = (int)(p / HW);
    const int s = (int)(p - (int64_t)b * HW);
    y[i] = c < C ? f2bf(g[((int64_t)b * C + c) * HW + s]) : (bf16_t)0;
  }
}
// y[NHWC bf16, width C] += scale * r[NCHW f32]   (T2I-adapter residuals, UW:100-106 / FD:208-218)
__global__ __launch_bounds__(256) void add_nchw_to_nhwc_kernel(const float* r, float scale, bf16_t* y, int B, int C, int HW) {
  const int64_t total = (int64_t)B * HW * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int b = (int)(p / HW);
    const int s = (int)(p - (int64_t)b * HW);
    y[i] = f2bf(fmaf(scale, r[((int64_t)b * C + c) * HW + s], bf2f(y[i])));
  }
}

// sinusoidal timestep embedding (diffusers Timesteps: [sin | cos], flipped to [cos | sin])
__global__ __launch_bounds__(256) void timestep_embed_kernel(const float* t, bf16_t* out, int B, int dim,
                                                             int flip, float shift) {
  const int half = dim / 2;
  const int64_t total = (int64_t)B * half;
  GRID_STRIDE(i, total) {
    const int k = (int)(i % half);
    const int b = (int)(i / half);
    const float freq = expf(-9.210340371976184f * (float)k / ((float)half - shift));
    const float arg = t[b] * freq;
    const float sn = sinf(arg), cs = cosf(arg);
    bf16_t* o = out + (int64_t)b * dim;
    if (flip) {
      o[k] = f2bf(cs);
      o[half + k] = f2bf(sn);
    } else {
      o[k] = f2bf(sn);
      o[half + k] = f2bf(cs);
    }
  }
}

__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* x, bf16_t* y, int64_t n) {
  GRID_STRIDE(i, n) y[i] = f2bf(silu_f(bf2f(x[i])));
}
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* x, bf16_t* y, int64_t n) {
  GRID_STRIDE(i, n) y[i] = f2bf(x[i]);
}

// strided 2-D copy / accumulate in 8-channel chunks
__global__ __launch_bounds__(256) void copy2d_kernel(const bf16_t* src, int64_t lds, int sc0, bf16_t* dst,
                                                     int64_t ldd, int dc0, int64_t rows, int cols,
                                                     int accumulate) {
  const int cpr = cols >> 3;
  const int64_t total = rows * cpr;
  GRID_STRIDE(i, total) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    u16x8 v = *(const u16x8*)(src + r * lds + sc0 + c);
    bf16_t* d = dst + r * ldd + dc0 + c;
    if (accumulate) {
      const u16x8 o = *(const u16x8*)d;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(o[e]));
    }
    *(u16x8*)d = v;
  }
}

__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const bf16_t* dy, bf16_t* dx, int B, int H, int W,
                                                          int C, int accumulate) {
  const int cpr = C >> 3;
  const int64_t total = (int64_t)B * H * W * cpr;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cpr) * 8;
    const int64_t p = i / cpr;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int b = (int)(p / ((int64_t)W * H));
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
      for (int dx_ = 0; dx_ < 2; ++dx_) {
        const u16x8 v =
            *(const u16x8*)(dy + (((int64_t)b * 2 * H + 2 * y + dy_) * 2 * W + 2 * x + dx_) * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
      }
    bf16_t* d = dx + p * C + c;
    u16x8 o;
    if (accumulate) {
      const u16x8 prev = *(const u16x8*)d;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(prev[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    *(u16x8*)d = o;
  }
}

// GEGLU backward on the 16-wide (value | gate) interleaved pre-activation layout.
//   out = val * gelu(gate);  dval = dout * gelu(gate);  dgate = dout * val * gelu'(gate)
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* pre, const bf16_t* dout, bf16_t* dpre,
                                                        int64_t M, int F) {
  const int cpr = F >> 3;  // 8 output columns per thread (half a 16-block)
  const int64_t total = M * cpr;
  GRID_STRIDE(i, total) {
    const int64_t m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;           // output column
    const int pc = (c >> 4) * 32 + (c & 15);        // value column in the interleaved layout
    const u16x8 v = *(const u16x8*)(pre + m * 2 * F + pc);
    const u16x8 g = *(const u16x8*)(pre + m * 2 * F + pc + 16);
    const u16x8 d = *(const u16x8*)(dout + m * F + c);
    u16x8 dv, dg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gf = bf2f(g[e]), vf = bf2f(v[e]), df = bf2f(d[e]);
      dv[e] = f2bf(df * gelu_f(gf));
      dg[e] = f2bf(df * vf * dgelu_f(gf));
    }
    *(u16x8*)(dpre + m * 2 * F + pc) = dv;
    *(u16x8*)(dpre + m * 2 * F + pc + 16) = dg;
  }
}

// bf16 2-D transpose through LDS: out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose2d_kernel(const bf16_t* in, int64_t ldi, bf16_t* out, int64_t ldo,
                                                          int64_t rows, int cols, int64_t rows_pad) {
  __shared__ bf16_t tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(r0 + r) * ldi + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows_pad) out[(int64_t)(c0 + c) * ldo + r0 + r] = tile[r][c];
  }
}

__global__ __launch_bounds__(256) void pad_cols_kernel(const bf16_t* src, int cols, bf16_t* dst, int cols_pad,
                                                       int64_t rows) {
  const int64_t total = rows * cols_pad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cols_pad);
    const int64_t r = i / cols_pad;
    dst[i] = c < cols ? src[r * cols + c] : (bf16_t)0;
  }
}

// LoRA refresh: f32 master w[rows][cols] -> bf16 wb[rows][cols] and bf16 wtb[cols][rows]
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* w, bf16_t* wb, bf16_t* wtb, int rows,
                                                             int cols) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    bf16_t v = 0;
    if (r0 + r < rows && c0 + c < cols) {
      v = f2bf(w[(int64_t)(r0 + r) * cols + c0 + c]);
      wb[(int64_t)(r0 + r) * cols + c0 + c] = v;
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows) wtb[(int64_t)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}

// the same for EVERY LoRA matrix of a plan in one launch: one block per 64x64 tile of the job table
__global__ __launch_bounds__(256) void cast_transpose_jobs_kernel(const CastJob* jobs) {
  __shared__ bf16_t tile[64][66];
  const CastJob jb = jobs[blockIdx.x];
  const int r0 = jb.r0, c0 = jb.c0, rows = jb.rows, cols = jb.cols;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    bf16_t v = 0;
    if (r0 + r < rows && c0 + c < cols) {
      v = f2bf(jb.src[(int64_t)(r0 + r) * cols + c0 + c]);
      jb.dst[(int64_t)(r0 + r) * cols + c0 + c] = v;
      if (jb.dst2) jb.dst2[(int64_t)(r0 + r) * jb.ld2 + c0 + c] = v;
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows) {
      jb.dstT[(int64_t)(c0 + c) * (jb.ldT ? jb.ldT : rows) + r0 + r] = tile[r][c];
      if (jb.dstT2) jb.dstT2[(int64_t)(c0 + c) * jb.ldT2 + r0 + r] = tile[r][c];
    }
  }
}

// fused AdamW (torch.optim.AdamW semantics, decoupled weight decay, bias correction)
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2, float grad_scale) {
  GRID_STRIDE(i, n) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * mi / denom;
    p[i] = pi;
  }
}

// fused scheduler / CFG element-wise helpers (fp32 latents, [B,4,H,W])
__global__ __launch_bounds__(256) void add_noise_kernel(const float* z, const float* noise, const float* sa,
                                                        const float* sb, float* out, int B, int64_t per) {
  const int64_t total = (int64_t)B * per;
  GRID_STRIDE(i, total) {
    const int b = (int)(i / per);
    out[i] = sa[b] * z[i] + sb[b] * noise[i];
  }
}
__global__ __launch_bounds__(256) void axpby4_kernel(const float* x0, float c0, const float* x1, float c1,
                                                     const float* x2, float c2, const float* x3, float c3,
                                                     float* out, int64_t n) {
  GRID_STRIDE(i, n) {
    float v = c0 * x0[i];
    if (x1) v += c1 * x1[i];
    if (x2) v += c2 * x2[i];
    if (x3) v += c3 * x3[i];
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
#define LAUNCH(kern, total, ...)                                                     \
  hipLaunchKernelGGL(kern, dim3(nblocks(total)), dim3(256), 0, st, __VA_ARGS__);     \
  FDMI_HIP(hipGetLastError());                                                       \
  return 0;

int launch_nchw_to_nhwc(const float* x, bf16_t* y, int B, int C, int HW, int Cpad, hipStream_t st) {
  LAUNCH(nchw_to_nhwc_kernel, (int64_t)B * HW * Cpad, x, y, B, C, HW, Cpad)
}
int launch_nhwc_to_nchw(const bf16_t* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate,
                        hipStream_t st) {
  LAUNCH(nhwc_to_nchw_kernel, (int64_t)B * C * HW, x, ldx, y, B, C, HW, accumulate)
}
int launch_add_nchw_to_nhwc(const float* r, float scale, bf16_t* y, int B, int C, int HW, hipStream_t st) {
  LAUNCH(add_nchw_to_nhwc_kernel, (int64_t)B * HW * C, r, scale, y, B, C, HW)
}
int launch_nchw_grad_to_nhwc(const float* g, bf16_t* y, int64_t ldy, int B, int C, int HW, hipStream_t st) {
  LAUNCH(nchw_grad_to_nhwc_kernel, (int64_t)B * HW * ldy, g, y, ldy, B, C, HW)
}
int launch_timestep_embed(const float* t, bf16_t* out, int B, int dim, int flip, float shift, hipStream_t st) {
  LAUNCH(timestep_embed_kernel, (int64_t)B * (dim / 2), t, out, B, dim, flip, shift)
}
int launch_silu(const bf16_t* x, bf16_t* y, int64_t n, hipStream_t st) { LAUNCH(silu_kernel, n, x, y, n) }
int launch_f32_to_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t st) {
  LAUNCH(f32_to_bf16_kernel, n, x, y, n)
}
int launch_copy2d(const bf16_t* src, int64_t lds, int sc0, bf16_t* dst, int64_t ldd, int dc0, int64_t rows,
                  int cols, int accumulate, hipStream_t st) {
  FDMI_CHECK(cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && sc0 % 8 == 0 && dc0 % 8 == 0,
             "copy2d: 8-element alignment required");
  LAUNCH(copy2d_kernel, rows * (cols / 8), src, lds, sc0, dst, ldd, dc0, rows, cols, accumulate)
}
int launch_pool2x2_sum(const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C, int accumulate,
                       hipStream_t st) {
  FDMI_CHECK(C % 8 == 0, "pool2x2: C must be a multiple of 8");
  LAUNCH(pool2x2_sum_kernel, (int64_t)B * H * W * (C / 8), dy, dx, B, H, W, C, accumulate)
}
int launch_geglu_bwd(const bf16_t* pre, const bf16_t* dout, bf16_t* dpre, int64_t M, int F, hipStream_t st) {
  FDMI_CHECK(F % 16 == 0, "geglu_bwd: F must be a multiple of 16");
  LAUNCH(geglu_bwd_kernel, M * (F / 8), pre, dout, dpre, M, F)
}
int launch_transpose2d(const bf16_t* in, int64_t ldi, bf16_t* out, int64_t ldo, int64_t rows, int cols,
                       hipStream_t st) {
  return launch_transpose2d_pad(in, ldi, out, ldo, rows, cols, rows, st);
}
int launch_transpose2d_pad(const bf16_t* in, int64_t ldi, bf16_t* out, int64_t ldo, int64_t rows, int cols,
                           int64_t rows_pad, hipStream_t st) {
  hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)((rows_pad + 63) / 64), cdiv(cols, 64)), dim3(256), 0, st,
                     in, ldi, out, ldo, rows, cols, rows_pad);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_pad_cols(const bf16_t* src, int cols, bf16_t* dst, int cols_pad, int64_t rows, hipStream_t st) {
  LAUNCH(pad_cols_kernel, rows * cols_pad, src, cols, dst, cols_pad, rows)
}
int launch_cast_transpose_jobs(const CastJob* jobs, int njobs, hipStream_t st) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(cast_transpose_jobs_kernel, dim3(njobs), dim3(256), 0, st, jobs);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_cast_transpose(const float* w, bf16_t* wb, bf16_t* wtb, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(cast_transpose_kernel, dim3(cdiv(rows, 64), cdiv(cols, 64)), dim3(256), 0, st, w, wb, wtb,
                     rows, cols);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                 float eps, float wd, int step, float grad_scale, hipStream_t st) {
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  LAUNCH(adamw_kernel, n, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2, grad_scale)
}
int launch_add_noise(const float* z, const float* noise, const float* sa, const float* sb, float* out, int B,
                     int64_t per, hipStream_t st) {
  LAUNCH(add_noise_kernel, (int64_t)B * per, z, noise, sa, sb, out, B, per)
}
int launch_axpby4(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                  const float* x3, float c3, float* out, int64_t n, hipStream_t st) {
  LAUNCH(axpby4_kernel, n, x0, c0, x1, c1, x2, c2, x3, c3, out, n)
}

// =============================================================================================
// discriminator support (a15: examples/train_flash_sd.py:225-240) and fused loss kernels (a12, a13)
// =============================================================================================
// explicit im2col (only for the discriminator's weight gradient; the forward / dgrad paths gather implicitly)
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* x, bf16_t* out, int B, int H, int W, int C,
                                                     int Ho, int Wo, int KH, int KW, int stride, int pad) {
  const int cpr = C >> 3;
  const int64_t K8 = (int64_t)KH * KW * cpr;
  const int64_t total = (int64_t)B * Ho * Wo * K8;
  GRID_STRIDE(i, total) {
    const int64_t m = i / K8;
    int r = (int)(i - m * K8);
    const int c = (r % cpr) * 8;
    r /= cpr;
    const int kx = r % KW, ky = r / KW;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((int64_t)Wo * Ho));
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (iy >= 0 && ix >= 0 && iy < H && ix < W) v = *(const u16x8*)(x + (((int64_t)b * H + iy) * W + ix) * C + c);
    *(u16x8*)(out + m * (K8 * 8) + (int64_t)(ky * KW + kx) * C + c) = v;
  }
}
__global__ __launch_bounds__(256) void silu_bwd_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n) {
  GRID_STRIDE(i, n) dx[i] = f2bf(bf2f(dy[i]) * dsilu_f(bf2f(x[i])));
}
// out0[c] += sum_r dy[r][c] ; out1[c] += sum_r dy[r][c] * xhat[r][c]   (xhat from GroupNorm stats when given)
// det (fdmi_det(), common.h): the launcher runs ONE block per 64 columns and the block's four row lanes are combined in lane order
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* dy, const bf16_t* x, const float* stats, float* out0,
                                                     float* out1, int64_t rows, int C, int HW, int G, float eps, int det) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rlane = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const int cpg = stats ? C / G : 1;
    for (int64_t r = (int64_t)blockIdx.y * 4 + rlane; r < rows; r += (int64_t)gridDim.y * 4) {
      const float d = bf2f(dy[r * C + c]);
      s0 += d;
      if (out1) {
        float xh = bf2f(x[r * C + c]);
        if (stats) {
          const int b = (int)(r / HW), gi = c / cpg;
          const float inv_n = 1.f / ((float)HW * cpg);
          const float sm = stats[((int64_t)b * G + gi) * 2] * inv_n, sq = stats[((int64_t)b * G + gi) * 2 + 1] * inv_n;
          xh = (xh - sm) * rsqrtf(fmaxf(sq - sm * sm, 0.f) + eps);
        }
        s1 += d * xh;
      }
    }
    if (!det) {
      atomicAdd(out0 + c, s0);
      if (out1) atomicAdd(out1 + c, s1);
    }
  }
  if (det) {
    __shared__ float cpart[4][64][2];
    cpart[rlane][threadIdx.x & 63][0] = s0;
    cpart[rlane][threadIdx.x & 63][1] = s1;
    __syncthreads();
    if (rlane == 0 && c < C) {
      const int l = threadIdx.x & 63;
      out0[c] += ((cpart[0][l][0] + cpart[1][l][0]) + cpart[2][l][0]) + cpart[3][l][0];
      if (out1) out1[c] += ((cpart[0][l][1] + cpart[1][l][1]) + cpart[2][l][1]) + cpart[3][l][1];
    }
  }
}
// distillation loss (FD:368-382): out += sum |s-t|^p / n  (p = 2: l2, p = 1: l1)
__global__ __launch_bounds__(256) void distill_loss_kernel(const float* s, const float* t, int64_t n, int l1, float* out) {
  float acc = 0.f;
  GRID_STRIDE(i, n) {
    const float d = s[i] - t[i];
    acc += l1 ? fabsf(d) : d * d;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1] + part[2] + part[3]) / (float)n);
}
__global__ __launch_bounds__(256) void distill_grad_kernel(const float* s, const float* t, int64_t n, int l1, float gscale,
                                                           float* ds) {
  GRID_STRIDE(i, n) {
    const float d = s[i] - t[i];
    ds[i] = l1 ? gscale * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * gscale * d;
  }
}
// DMD (FD:459-499), one block per sample for the weight, then the element-wise gradient + loss:
//   x0 = ia*noisy + ma*real ; w = 1/(mean|s - x0| + 1e-5) ; coeff = (real - fake)*kb
//   loss = mean((w*coeff)^2) ; dL/ds = 2*w*coeff/N
__global__ __launch_bounds__(256) void dmd_weight_kernel(const float* s, const float* noisy, const float* real, const float* ia,
                                                         const float* ma, float* w, int64_t per) {
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += 256) {
    const int64_t k = (int64_t)b * per + i;
    acc += fabsf(s[k] - (ia[b] * noisy[k] + ma[b] * real[k]));
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) w[b] = 1.f / ((part[0] + part[1] + part[2] + part[3]) / (float)per + 1e-5f);
}
__global__ __launch_bounds__(256) void dmd_grad_kernel(const float* real, const float* fake, const float* kb, const float* w,
                                                       float* grad, float* loss, int B, int64_t per) {
  const int64_t n = (int64_t)B * per;
  float acc = 0.f;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / per);
    const float wc = w[b] * (real[i] - fake[i]) * kb[b];
    grad[i] = 2.f * wc / (float)n;
    acc += wc * wc;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) / (float)n);
}

int launch_im2col(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride,
                  int pad, hipStream_t st) {
  FDMI_CHECK(C % 8 == 0, "im2col: C must be a multiple of 8");
  LAUNCH(im2col_kernel, (int64_t)B * Ho * Wo * KH * KW * (C / 8), x, out, B, H, W, C, Ho, Wo, KH, KW, stride, pad)
}
int launch_silu_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n, hipStream_t st) {
  LAUNCH(silu_bwd_kernel, n, x, dy, dx, n)
}
int launch_colsum(const bf16_t* dy, const bf16_t* x, const float* stats, float* out0, float* out1, int64_t rows, int C,
                  int HW, int G, float eps, hipStream_t st) {
  int gy = (int)((rows + 63) / 64);
  if (gy > 256) gy = 256;
  if (gy < 1) gy = 1;
  const int det = fdmi_det() ? 1 : 0;
  if (det) gy = 1;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), gy), dim3(256), 0, st, dy, x, stats, out0, out1, rows, C, HW, G, eps, det);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_distill_loss(const float* s, const float* t, int64_t n, int l1, float* out, hipStream_t st) {
  FDMI_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
  hipLaunchKernelGGL(distill_loss_kernel, dim3(fdmi_det() ? 1 : nblocks(n, 256)), dim3(256), 0, st, s, t, n, l1, out);   // (deterministic mode: one block = one contributor)
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_distill_grad(const float* s, const float* t, int64_t n, int l1, float gscale, float* ds, hipStream_t st) {
  LAUNCH(distill_grad_kernel, n, s, t, n, l1, gscale, ds)
}
int launch_dmd_loss(const float* s, const float* noisy, const float* real, const float* fake, const float* ia,
                    const float* ma, const float* kb, float* w, float* grad, float* loss, int B, int64_t per,
                    hipStream_t st) {
  FDMI_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
  hipLaunchKernelGGL(dmd_weight_kernel, dim3(B), dim3(256), 0, st, s, noisy, real, ia, ma, w, per);
  hipLaunchKernelGGL(dmd_grad_kernel, dim3(fdmi_det() ? 1 : nblocks((int64_t)B * per, 256)), dim3(256), 0, st, real, fake, kb, w, grad, loss, B,
                     per);
  FDMI_HIP(hipGetLastError());
  return 0;
}
