// fp32 VALIDATION MODE of the hot path (north_star: "loss parity to 1e-3 rel"; SURVEY.md section 7 "Hard parts" (i)):
// every contraction of the denoiser plans -- linear, implicit-GEMM convolution (forward and dgrad), the attention products,
// the LoRA weight gradients -- on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 with fp32 storage of every
// activation, and fp32 normalisation / element-wise kernels with fp64 statistics.  Same operand conventions as the bf16
// kernels (gemm.h GemmArgs: out[M,N] = A[M,K] W[N,K]^T + epilogue), so a plan built with precision = fp32 walks the same
// graph and differs from the bf16 plan only in storage type and kernel family.  These kernels are written for fidelity,
// not speed (64 x 64 tile, 4 waves, one 32x32 accumulator per wave, single-buffered LDS): the C1 step (3.3 TFLOP) takes
// ~0.1-0.2 s, which is all a parity gate needs.  The measured bf16 path never launches anything in this file.
#include "ops.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GRID_STRIDE32(i, total) \
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (total); i += (int64_t)gridDim.x * 256)

namespace {

inline int nblocks32(int64_t total, int cap = 8192) {
  int64_t b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

__device__ __forceinline__ float gelu32(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu32(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float silu32(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float dsilu32(float x) {
  const float s = 1.f / (1.f + expf(-x));
  return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 256 threads of a block (result valid in every thread)
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution.  Element (m, k) of A sits at A[m * lda + k * a_sk] (ROW) or is gathered from the
// NHWC tensor (CONV, forward or dgrad geometry as in gemm4.hip's retap()); element (n, k) of W at W[n * ldw + k * w_sk].
// A_MFAST / W_NFAST pick the thread mapping of the tile loads so that the unit-stride index runs across the lanes.
// blockIdx.z = (batch b1 * nb2 + b2) * ksplit + k-split; splits > 1 accumulate with atomics (accum_atomic only).
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void gemm32_kernel(const GemmArgs a, int ksplit, int a_mfast, int w_nfast) {
  __shared__ float As[16][68];
  __shared__ float Ws[16][68];
  __shared__ float Cs[64][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int z = blockIdx.z;
  const int batch = z / ksplit, ks = z - batch * ksplit;
  const int b1 = batch / a.nb2, b2 = batch - b1 * a.nb2;
  const float* A = (const float*)a.A + b1 * a.a_b1 + b2 * a.a_b2;
  const float* W = (const float*)a.W + b1 * a.w_b1 + b2 * a.w_b2;
  const int64_t coff = b1 * a.c_b1 + b2 * a.c_b2;
  const int ktiles = (a.K + 15) >> 4;
  const int per = (ktiles + ksplit - 1) / ksplit;
  const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
  const int ar = a_mfast ? (tid & 63) : (tid >> 2), akq = a_mfast ? (tid >> 6) : (tid & 3);
  const int wr = w_nfast ? (tid & 63) : (tid >> 2), wkq = w_nfast ? (tid >> 6) : (tid & 3);
  const bool mok = (m0 + ar) < a.M, nok = (n0 + wr) < a.N;
  int cpix = 0, cy = 0, cx = 0;   // CONV: batch pixel base and the window origin of this thread's output row
  if (MODE == GEMM_CONV && mok) {
    const int m = m0 + ar, hw = a.Hout * a.Wout;
    const int b = m / hw, rem = m - b * hw;
    const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
    cpix = b * a.Hin * a.Win;
    cy = a.dgrad ? (oy + a.pad) : (oy * a.stride - a.pad);
    cx = a.dgrad ? (ox + a.pad) : (ox * a.stride - a.pad);
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int k0 = kt * 16;
    float av[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const int kb = k0 + akq * 4;
      if (MODE == GEMM_ROW) {
        const float* src = A + (int64_t)(m0 + ar) * a.lda + (int64_t)kb * a.a_sk;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (mok && kb + i < a.K) av[i] = src[(int64_t)i * a.a_sk];
      } else if (mok && kb < a.K) {   // the 4 reduction indices share a filter tap (Cin % 4 == 0, checked by the launcher)
        const int tap = kb / a.Cin, c = kb - tap * a.Cin;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        bool ok;
        int sy, sx;
        if (a.dgrad) {
          const int ty = cy - ky, tx = cx - kx;
          ok = ty >= 0 && tx >= 0 && (ty % a.stride) == 0 && (tx % a.stride) == 0;
          sy = ty / a.stride;
          sx = tx / a.stride;
          ok = ok && sy < a.Hin && sx < a.Win;
        } else {
          const int iy = cy + ky, ix = cx + kx;
          ok = iy >= 0 && ix >= 0 && iy < (a.Hin << a.ups) && ix < (a.Win << a.ups);
          sy = iy >> a.ups;
          sx = ix >> a.ups;
        }
        if (ok) {
          const float4 v = *(const float4*)(A + (int64_t)(cpix + sy * a.Win + sx) * a.Cin + c);
          av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
        }
      }
    }
    {
      const int kb = k0 + wkq * 4;
      const float* src = W + (int64_t)(n0 + wr) * a.ldw + (int64_t)kb * a.w_sk;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nok && kb + i < a.K) wv[i] = src[(int64_t)i * a.w_sk];
    }
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[akq * 4 + i][ar] = av[i];
      Ws[wkq * 4 + i][wr] = wv[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      // v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
      const float fa = As[kk * 2 + (lane >> 5)][wm * 32 + (lane & 31)];
      const float fb = Ws[kk * 2 + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
    }
  }
  // accumulator -> LDS: D[i][j], j = lane & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) Cs[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wn * 32 + (lane & 31)] = acc[r];
  __syncthreads();
  const float* bias = a.bias;
  const float* rowvec = (const float*)a.rowvec;
  const float* residual = (const float*)a.residual;
  float* preact = (float*)a.preact;
  float* C = (float*)a.C + coff;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int row = e >> 6, col = e & 63;
    const int64_t m = m0 + row;
    const int n = n0 + col;
    if (m >= a.M || n >= a.N) continue;
    if (a.accum_atomic) {
      atomicAdd(C + m * a.ldc + n, a.alpha * Cs[row][col]);
      continue;
    }
    if (a.act == ACT_GEGLU) {   // packed columns: 16 values | 16 gates per block of 32 (tiles start at multiples of 64)
      if ((n & 31) >= 16) continue;
      float val = a.alpha * Cs[row][col], gate = a.alpha * Cs[row][col + 16];
      if (bias) { val += bias[n]; gate += bias[n + 16]; }
      if (rowvec) {
        const float* rv = rowvec + (m / a.rows_per_batch) * a.rowvec_ld;
        val += rv[n]; gate += rv[n + 16];
      }
      if (preact) { preact[m * a.ldp + n] = val; preact[m * a.ldp + n + 16] = gate; }
      const int no = (n >> 5) * 16 + (n & 15);
      float o = val * gelu32(gate);
      if (residual) o += residual[m * a.ldr + no];
      C[m * a.ldc + no] = o;
      continue;
    }
    float v = a.alpha * Cs[row][col];
    if (bias) v += bias[n];
    if (rowvec) {
      const float rv = rowvec[(m / a.rows_per_batch) * a.rowvec_ld + n];
      v = a.rowvec_mul ? v * rv : v + rv;
    }
    if (residual) v += residual[m * a.ldr + n];
    if (a.act == ACT_SILU) v = silu32(v);
    else if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (a.act == ACT_GELU) v = gelu32(v);
    else if (a.act == ACT_GELU_TANH) v = 0.5f * v * (1.f + tanhf(0.7978845608028654f * fmaf(0.044715f * v * v, v, v)));
    C[m * a.ldc + n] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// attention pieces: row softmax of the materialised scores and its backward (in place)
// ---------------------------------------------------------------------------------------------------------------------
// causal_sq > 0: rows are queries i = row % causal_sq of a causal self-attention; keys j > i get probability 0
// bias [H][Sq][n] (shared by the samples; T5's relative-position bias) and kbias [sample][n] (additive key mask, e.g. 0 / -inf-like)
// are added to the scores first when given (rows = samples * H * Sq, row = (sample * H + h) * Sq + i)
__global__ __launch_bounds__(256) void softmax32_kernel(float* S, int64_t rows, int n, int64_t ld, int causal_sq = 0,
                                                        const float* bias = nullptr, const float* kbias = nullptr, int H = 1, int Sq = 1) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* s = S + row * ld;
  if (bias || kbias) {
    const int64_t hi = row % ((int64_t)H * Sq), smp = row / ((int64_t)H * Sq);
    const float* br = bias ? bias + hi * n : nullptr;
    const float* kr = kbias ? kbias + smp * n : nullptr;
    for (int j = lane; j < n; j += 64) s[j] += (br ? br[j] : 0.f) + (kr ? kr[j] : 0.f);
  }
  if (causal_sq > 0) {
    const int nv = (int)(row % causal_sq) + 1;
    for (int j = nv + lane; j < n; j += 64) s[j] = 0.f;   // (the masked tail; the rest of the kernel sees n = i + 1 keys)
    n = nv < n ? nv : n;
  }
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, s[j]);
  mx = wave_max(mx);
  double sum = 0.0;
  for (int j = lane; j < n; j += 64) sum += (double)expf(s[j] - mx);
  sum = wave_sum_d(sum);
  const float inv = (float)(1.0 / sum);
  for (int j = lane; j < n; j += 64) s[j] = expf(s[j] - mx) * inv;
}
// dS = P o (dP - rowsum(P o dP)), written over dP
__global__ __launch_bounds__(256) void softmax32_bwd_kernel(const float* P, float* dP, int64_t rows, int n, int64_t ld) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = P + row * ld;
  float* d = dP + row * ld;
  double dot = 0.0;
  for (int j = lane; j < n; j += 64) dot += (double)p[j] * (double)d[j];
  const float delta = (float)wave_sum_d(dot);
  for (int j = lane; j < n; j += 64) d[j] = p[j] * (d[j] - delta);
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) on [B][HW][C]: one block per (group, sample); stats[b][g] = (mean, rstd)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn32_fwd_kernel(const float* x, const float* gamma, const float* beta, float* stats,
                                                       float* y, int HW, int C, int G, float eps, int silu) {
  __shared__ double sh[4];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / G;
  const int64_t n = (int64_t)HW * cpg;
  const float* xb = x + (int64_t)b * HW * C + g * cpg;
  float* yb = y + (int64_t)b * HW * C + g * cpg;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t r = i / cpg;
    s += (double)xb[r * C + (i - r * cpg)];
  }
  const double mean = block_sum_d(s, sh) / (double)n;
  double ss = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t r = i / cpg;
    const double d = (double)xb[r * C + (i - r * cpg)] - mean;
    ss += d * d;
  }
  const double var = block_sum_d(ss, sh) / (double)n;
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    stats[((int64_t)b * G + g) * 2] = mu;
    stats[((int64_t)b * G + g) * 2 + 1] = rstd;
  }
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t r = i / cpg;
    const int c = (int)(i - r * cpg);
    float zv = (xb[r * C + c] - mu) * rstd * gamma[g * cpg + c] + beta[g * cpg + c];
    if (silu) zv = silu32(zv);
    yb[r * C + c] = zv;
  }
}
__global__ __launch_bounds__(256) void gn32_bwd_kernel(const float* x, const float* dy, const float* gamma, const float* beta,
                                                       const float* stats, float* dx, int HW, int C, int G, int silu,
                                                       int accumulate) {
  __shared__ double sh[4];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / G;
  const int64_t n = (int64_t)HW * cpg;
  const int64_t base = (int64_t)b * HW * C + g * cpg;
  const float mu = stats[((int64_t)b * G + g) * 2], rstd = stats[((int64_t)b * G + g) * 2 + 1];
  auto dxhat = [&](int64_t r, int c, float& xh) {
    xh = (x[base + r * C + c] - mu) * rstd;
    float dz = dy[base + r * C + c];
    const float gm = gamma[g * cpg + c];
    if (silu) dz *= dsilu32(gm * xh + beta[g * cpg + c]);
    return dz * gm;
  };
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t r = i / cpg;
    float xh;
    const float d = dxhat(r, (int)(i - r * cpg), xh);
    s1 += (double)d;
    s2 += (double)d * (double)xh;
  }
  const float m1 = (float)(block_sum_d(s1, sh) / (double)n);
  const float m2 = (float)(block_sum_d(s2, sh) / (double)n);
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t r = i / cpg;
    const int c = (int)(i - r * cpg);
    float xh;
    const float d = dxhat(r, c, xh);
    float v = rstd * (d - m1 - xh * m2);
    if (accumulate) v += dx[base + r * C + c];
    dx[base + r * C + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm (optional affine, optional adaLN modulate y (1 + scale[b]) + shift[b]): one wavefront per row
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln32_fwd_kernel(const float* x, const float* gamma, const float* beta, const float* shift,
                                                       const float* scale, int64_t mod_ld, int rows_per_batch, float* y,
                                                       int64_t rows, int C, float eps, float* stats) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * C;
  double s = 0.0;
  for (int c = lane; c < C; c += 64) s += (double)xr[c];
  const double mean = wave_sum_d(s) / C;
  double ss = 0.0;
  for (int c = lane; c < C; c += 64) {
    const double d = (double)xr[c] - mean;
    ss += d * d;
  }
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt(wave_sum_d(ss) / C + (double)eps));
  if (stats && lane == 0) {
    stats[row * 2] = mu;
    stats[row * 2 + 1] = rstd;
  }
  const int64_t b = row / rows_per_batch;
  for (int c = lane; c < C; c += 64) {
    float v = (xr[c] - mu) * rstd;
    if (gamma) v = v * gamma[c] + (beta ? beta[c] : 0.f);
    if (scale) v = v * (1.f + scale[b * mod_ld + c]) + shift[b * mod_ld + c];
    y[row * C + c] = v;
  }
}
__global__ __launch_bounds__(256) void ln32_bwd_kernel(const float* x, const float* dy, const float* gamma, const float* scale,
                                                       int64_t mod_ld, int rows_per_batch, float* dx, int64_t rows, int C,
                                                       float eps, int accumulate) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * C;
  const float* dr = dy + row * C;
  double s = 0.0;
  for (int c = lane; c < C; c += 64) s += (double)xr[c];
  const double mean = wave_sum_d(s) / C;
  double ss = 0.0;
  for (int c = lane; c < C; c += 64) {
    const double d = (double)xr[c] - mean;
    ss += d * d;
  }
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt(wave_sum_d(ss) / C + (double)eps));
  const int64_t b = row / rows_per_batch;
  auto dxhat = [&](int c) {
    float d = dr[c];
    if (gamma) d *= gamma[c];
    if (scale) d *= 1.f + scale[b * mod_ld + c];
    return d;
  };
  double s1 = 0.0, s2 = 0.0;
  for (int c = lane; c < C; c += 64) {
    const float d = dxhat(c);
    s1 += (double)d;
    s2 += (double)d * (double)((xr[c] - mu) * rstd);
  }
  const float m1 = (float)(wave_sum_d(s1) / C), m2 = (float)(wave_sum_d(s2) / C);
  for (int c = lane; c < C; c += 64) {
    float v = rstd * (dxhat(c) - m1 - (xr[c] - mu) * rstd * m2);
    if (accumulate) v += dx[row * C + c];
    dx[row * C + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// element-wise / layout kernels (fp32 twins of elem.hip)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc32_kernel(const float* x, float* y, int B, int C, int HW, int Cpad) {
  const int64_t total = (int64_t)B * HW * Cpad;
  GRID_STRIDE32(i, total) {
    const int c = (int)(i % Cpad);
    const int64_t p = i / Cpad;
    const int b = (int)(p / HW);
    const int s = (int)(p - (int64_t)b * HW);
    y[i] = c < C ? x[((int64_t)b * C + c) * HW + s] : 0.f;
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw32_kernel(const float* x, int64_t ldx, float* y, int B, int C, int HW,
                                                             int accumulate) {
  const int64_t total = (int64_t)B * C * HW;
  GRID_STRIDE32(i, total) {
    const int s = (int)(i % HW);
    const int64_t bc = i / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const float v = x[((int64_t)b * HW + s) * ldx + c];
    y[i] = accumulate ? y[i] + v : v;
  }
}
__global__ __launch_bounds__(256) void add_nchw_to_nhwc32_kernel(const float* r, float scale, float* y, int B, int C, int HW) {
  const int64_t total = (int64_t)B * HW * C;
  GRID_STRIDE32(i, total) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int b = (int)(p / HW);
    const int s = (int)(p - (int64_t)b * HW);
    y[i] = fmaf(scale, r[((int64_t)b * C + c) * HW + s], y[i]);
  }
}
__global__ __launch_bounds__(256) void timestep_embed32_kernel(const float* t, float* out, int B, int dim, int flip, float shift) {
  const int half = dim / 2;
  const int64_t total = (int64_t)B * half;
  GRID_STRIDE32(i, total) {
    const int k = (int)(i % half);
    const int b = (int)(i / half);
    const float freq = expf(-9.210340371976184f * (float)k / ((float)half - shift));
    const float arg = t[b] * freq;
    float* o = out + (int64_t)b * dim;
    o[flip ? k : half + k] = cosf(arg);
    o[flip ? half + k : k] = sinf(arg);
  }
}
__global__ __launch_bounds__(256) void silu32_kernel(const float* x, float* y, int64_t n) { GRID_STRIDE32(i, n) y[i] = silu32(x[i]); }
__global__ __launch_bounds__(256) void silu32_bwd_kernel(const float* x, const float* dy, float* dx, int64_t n) {
  GRID_STRIDE32(i, n) dx[i] = dy[i] * dsilu32(x[i]);
}
__global__ __launch_bounds__(256) void gelu_tanh32_kernel(const float* x, float* y, int64_t n) {
  GRID_STRIDE32(i, n) {
    const float v = x[i];
    y[i] = 0.5f * v * (1.f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
  }
}
__global__ __launch_bounds__(256) void gelu_tanh32_bwd_kernel(const float* x, const float* dy, float* dx, int64_t n) {
  GRID_STRIDE32(i, n) {
    const float v = x[i];
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    const float th = tanhf(u);
    const float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * v * v);
    dx[i] = dy[i] * (0.5f * (1.f + th) + 0.5f * v * (1.f - th * th) * du);
  }
}
__global__ __launch_bounds__(256) void copy2d32_kernel(const float* src, int64_t lds, int sc0, float* dst, int64_t ldd, int dc0,
                                                       int64_t rows, int cols, int accumulate) {
  const int64_t total = rows * cols;
  GRID_STRIDE32(i, total) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float v = src[r * lds + sc0 + c];
    float* d = dst + r * ldd + dc0 + c;
    *d = accumulate ? *d + v : v;
  }
}
__global__ __launch_bounds__(256) void pool2x2_sum32_kernel(const float* dy, float* dx, int B, int H, int W, int C, int accumulate) {
  const int64_t total = (int64_t)B * H * W * C;
  GRID_STRIDE32(i, total) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int b = (int)(p / ((int64_t)W * H));
    const float* s = dy + (((int64_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
    float v = s[0] + s[C] + s[(int64_t)2 * W * C] + s[(int64_t)2 * W * C + C];
    if (accumulate) v += dx[i];
    dx[i] = v;
  }
}
// pre[M][2F] in 16-wide (value | gate) interleave
__global__ __launch_bounds__(256) void geglu32_bwd_kernel(const float* pre, const float* dout, float* dpre, int64_t M, int F) {
  const int64_t total = M * F;
  GRID_STRIDE32(i, total) {
    const int64_t m = i / F;
    const int c = (int)(i - m * F);
    const int pc = (c >> 4) * 32 + (c & 15);
    const float v = pre[m * 2 * F + pc], g = pre[m * 2 * F + pc + 16], d = dout[i];
    dpre[m * 2 * F + pc] = d * gelu32(g);
    dpre[m * 2 * F + pc + 16] = d * v * dgelu32(g);
  }
}
__global__ __launch_bounds__(256) void pad_cols32_kernel(const float* src, int cols, float* dst, int cols_pad, int64_t rows) {
  const int64_t total = rows * cols_pad;
  GRID_STRIDE32(i, total) {
    const int c = (int)(i % cols_pad);
    const int64_t r = i / cols_pad;
    dst[i] = c < cols ? src[r * cols + c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void im2col32_kernel(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                                       int KH, int KW, int stride, int pad) {
  const int64_t K = (int64_t)KH * KW * C;
  const int64_t total = (int64_t)B * Ho * Wo * K;
  GRID_STRIDE32(i, total) {
    const int64_t m = i / K;
    int r = (int)(i - m * K);
    const int c = r % C;
    r /= C;
    const int kx = r % KW, ky = r / KW;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((int64_t)Wo * Ho));
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    out[i] = (iy >= 0 && ix >= 0 && iy < H && ix < W) ? x[(((int64_t)b * H + iy) * W + ix) * C + c] : 0.f;
  }
}
// out0[c] += sum_r dy ; out1[c] += sum_r dy * xhat   (xhat = (x - mean) rstd from stats[b][g] = (mean, rstd) when given)
__global__ __launch_bounds__(256) void colsum32_kernel(const float* dy, const float* x, const float* stats, float* out0, float* out1,
                                                       int64_t rows, int C, int HW, int G) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rlane = threadIdx.x >> 6;
  if (c >= C) return;
  const int cpg = stats ? C / G : 1;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t r = (int64_t)blockIdx.y * 4 + rlane; r < rows; r += (int64_t)gridDim.y * 4) {
    const float d = dy[r * C + c];
    s0 += (double)d;
    if (out1) {
      float xh = x[r * C + c];
      if (stats) {
        const int64_t sb = ((r / HW) * G + c / cpg) * 2;
        xh = (xh - stats[sb]) * stats[sb + 1];
      }
      s1 += (double)d * (double)xh;
    }
  }
  atomicAdd(out0 + c, (float)s0);
  if (out1) atomicAdd(out1 + c, (float)s1);
}
__global__ __launch_bounds__(256) void gate_residual32_kernel(const float* x, const float* gate, int64_t gate_ld, const float* res,
                                                              float* y, int64_t rows, int C, int rows_per_batch) {
  const int64_t total = rows * C;
  GRID_STRIDE32(i, total) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float v = gate[(r / rows_per_batch) * gate_ld + c] * x[i];
    y[i] = res ? res[i] + v : v;
  }
}
// out1[b][c] = sum_r dy ; out0[b][c] = sum_r dy * f(x)   (f = LayerNorm normalisation from stats[row] = (mean, rstd))
__global__ __launch_bounds__(256) void batch_colsum32_kernel(const float* dy, const float* x, const float* stats, float* out0,
                                                             float* out1, int rows_per_batch, int C) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int r = 0; r < rows_per_batch; ++r) {
    const int64_t row = (int64_t)b * rows_per_batch + r;
    const float d = dy[row * C + c];
    s1 += (double)d;
    if (out0) {
      float f = x[row * C + c];
      if (stats) f = (f - stats[row * 2]) * stats[row * 2 + 1];
      s0 += (double)d * (double)f;
    }
  }
  if (out1) out1[(int64_t)b * C + c] = (float)s1;
  if (out0) out0[(int64_t)b * C + c] = (float)s0;
}

#define LAUNCH32(kernel, total, ...)                                                       \
  hipLaunchKernelGGL(kernel, dim3(nblocks32(total)), dim3(256), 0, st, __VA_ARGS__);       \
  FDMI_HIP(hipGetLastError());                                                             \
  return 0;

}  // namespace

// =====================================================================================================================
// launchers
// =====================================================================================================================
int launch_gemm32(const GemmArgs& a, hipStream_t st) {
  FDMI_CHECK(a.f32 == 1, "gemm32: not an fp32 problem");
  FDMI_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.A && a.W && a.C, "gemm32: null / empty operand");
  FDMI_CHECK(a.act != ACT_GEGLU || (a.N % 32 == 0 && !a.accum_atomic), "gemm32: GEGLU needs packed N % 32 == 0");
  if (a.mode == GEMM_CONV) {
    FDMI_CHECK(a.Cin % 4 == 0 && a.K == a.KH * a.KW * a.Cin && a.stride >= 1, "gemm32: conv needs Cin % 4 == 0, K = KH KW Cin");
    FDMI_CHECK(((uintptr_t)a.A & 15) == 0, "gemm32: conv activation must be 16-byte aligned");
  }
  const int tiles = cdiv(a.M, 64) * cdiv(a.N, 64) * a.nb1 * a.nb2;
  int ksplit = 1;
  if (a.accum_atomic) {   // deep reductions over few tiles (the LoRA weight gradients): split along K
    const int ktiles = cdiv(a.K, 16);
    while (tiles * ksplit < 512 && ktiles / (ksplit * 2) >= 8) ksplit *= 2;
  }
  const int a_mfast = a.mode == GEMM_ROW && a.lda == 1 && a.a_sk != 1;
  const int w_nfast = a.ldw == 1 && a.w_sk != 1;
  dim3 grid(cdiv(a.M, 64), cdiv(a.N, 64), a.nb1 * a.nb2 * ksplit);
  FDMI_CHECK(grid.y <= 65535 && grid.z <= 65535, "gemm32: grid too large");
  if (a.mode == GEMM_ROW) hipLaunchKernelGGL(gemm32_kernel<GEMM_ROW>, grid, dim3(256), 0, st, a, ksplit, a_mfast, w_nfast);
  else hipLaunchKernelGGL(gemm32_kernel<GEMM_CONV>, grid, dim3(256), 0, st, a, ksplit, a_mfast, w_nfast);
  FDMI_HIP(hipGetLastError());
  return 0;
}

// C[n1][n2] += sum_m X[m][n1] Y[m][n2]
int launch_wgrad_tn32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t M, int N1, int N2, float* C, int64_t ldc,
                      hipStream_t st) {
  GemmArgs g;
  g.f32 = 1; g.M = N1; g.N = N2; g.K = (int)M;
  g.A = (const bf16_t*)X; g.lda = 1; g.a_sk = ldx;
  g.W = (const bf16_t*)Y; g.ldw = 1; g.w_sk = ldy;
  g.C = C; g.ldc = ldc; g.out_f32 = 1; g.accum_atomic = 1;
  return launch_gemm32(g, st);
}

static int64_t attn32_skvp(int Skv) { return (Skv + 3) & ~3; }
int64_t attn32_scratch_elems(int B, int H, int Sq, int Skv, int bwd) {
  // scores (and, backward, their gradient) of as many samples at a time as fit ~1 GiB, at least one
  const int64_t per = (int64_t)H * Sq * attn32_skvp(Skv) * (bwd ? 2 : 1);
  int64_t nb = ((int64_t)1 << 28) / per;
  if (nb < 1) nb = 1;
  if (nb > B) nb = B;
  return nb * per;
}
static GemmArgs attn32_gemm(int M, int N, int K, const float* A, int64_t lda, int64_t a_sk, const float* W, int64_t ldw,
                            int64_t w_sk, float* C, int64_t ldc, float alpha, int nb, int H) {
  GemmArgs g;
  g.f32 = 1; g.M = M; g.N = N; g.K = K;
  g.A = (const bf16_t*)A; g.lda = lda; g.a_sk = a_sk;
  g.W = (const bf16_t*)W; g.ldw = ldw; g.w_sk = w_sk;
  g.C = C; g.ldc = ldc; g.out_f32 = 1; g.alpha = alpha;
  g.nb1 = nb; g.nb2 = H;
  return g;
}
// O = softmax(scale Q K^T) V on [B, S, ld] operands with head h at column offset h * d
int launch_attn32_fwd(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                      int B, int H, int Sq, int Skv, int d, float scale, float* scratch, int64_t scratch_elems, hipStream_t st,
                      int causal, const float* bias, const float* kbias) {
  FDMI_CHECK(!causal || Sq == Skv, "attn32: the causal mask is defined for self-attention (Sq == Skv)");
  const int64_t Sp = attn32_skvp(Skv), per = (int64_t)H * Sq * Sp;
  const int cb = (int)(scratch_elems / per < B ? scratch_elems / per : B);
  FDMI_CHECK(cb >= 1, "attn32: scratch too small");
  for (int b0 = 0; b0 < B; b0 += cb) {
    const int nb = B - b0 < cb ? B - b0 : cb;
    GemmArgs s = attn32_gemm(Sq, Skv, d, Q + (int64_t)b0 * Sq * ldq, ldq, 1, K + (int64_t)b0 * Skv * ldk, ldk, 1, scratch, Sp,
                             scale, nb, H);
    s.a_b1 = (int64_t)Sq * ldq; s.a_b2 = d; s.w_b1 = (int64_t)Skv * ldk; s.w_b2 = d; s.c_b1 = per; s.c_b2 = (int64_t)Sq * Sp;
    if (int rc = launch_gemm32(s, st)) return rc;
    const int64_t rows = (int64_t)nb * H * Sq;
    hipLaunchKernelGGL(softmax32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, scratch, rows, Skv, Sp, causal ? Sq : 0, bias,
                       kbias ? kbias + (int64_t)b0 * Skv : nullptr, H, Sq);
    GemmArgs o = attn32_gemm(Sq, d, Skv, scratch, Sp, 1, V + (int64_t)b0 * Skv * ldv, 1, ldv, O + (int64_t)b0 * Sq * ldo, ldo, 1.f,
                             nb, H);
    o.a_b1 = per; o.a_b2 = (int64_t)Sq * Sp; o.w_b1 = (int64_t)Skv * ldv; o.w_b2 = d; o.c_b1 = (int64_t)Sq * ldo; o.c_b2 = d;
    if (int rc = launch_gemm32(o, st)) return rc;
  }
  FDMI_HIP(hipGetLastError());
  return 0;
}
// dQ, dK, dV (written, not accumulated) from dO; scores recomputed
int launch_attn32_bwd(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, const float* dO,
                      int64_t lddo, float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV, int64_t lddv, int B, int H, int Sq,
                      int Skv, int d, float scale, float* scratch, int64_t scratch_elems, hipStream_t st) {
  const int64_t Sp = attn32_skvp(Skv), per = (int64_t)H * Sq * Sp;
  const int cb = (int)(scratch_elems / (2 * per) < B ? scratch_elems / (2 * per) : B);
  FDMI_CHECK(cb >= 1, "attn32 bwd: scratch too small");
  for (int b0 = 0; b0 < B; b0 += cb) {
    const int nb = B - b0 < cb ? B - b0 : cb;
    float* P = scratch;
    float* dP = scratch + (int64_t)nb * per;
    const float* q = Q + (int64_t)b0 * Sq * ldq;
    const float* k = K + (int64_t)b0 * Skv * ldk;
    const float* v = V + (int64_t)b0 * Skv * ldv;
    const float* go = dO + (int64_t)b0 * Sq * lddo;
    const int64_t pb2 = (int64_t)Sq * Sp;
    GemmArgs s = attn32_gemm(Sq, Skv, d, q, ldq, 1, k, ldk, 1, P, Sp, scale, nb, H);
    s.a_b1 = (int64_t)Sq * ldq; s.a_b2 = d; s.w_b1 = (int64_t)Skv * ldk; s.w_b2 = d; s.c_b1 = per; s.c_b2 = pb2;
    if (int rc = launch_gemm32(s, st)) return rc;
    const int64_t rows = (int64_t)nb * H * Sq;
    hipLaunchKernelGGL(softmax32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, rows, Skv, Sp, 0);
    // dV[j][c] = sum_i P[i][j] dO[i][c]
    GemmArgs gv = attn32_gemm(Skv, d, Sq, P, 1, Sp, go, 1, lddo, dV + (int64_t)b0 * Skv * lddv, lddv, 1.f, nb, H);
    gv.a_b1 = per; gv.a_b2 = pb2; gv.w_b1 = (int64_t)Sq * lddo; gv.w_b2 = d; gv.c_b1 = (int64_t)Skv * lddv; gv.c_b2 = d;
    if (int rc = launch_gemm32(gv, st)) return rc;
    // dP = dO V^T
    GemmArgs gp = attn32_gemm(Sq, Skv, d, go, lddo, 1, v, ldv, 1, dP, Sp, 1.f, nb, H);
    gp.a_b1 = (int64_t)Sq * lddo; gp.a_b2 = d; gp.w_b1 = (int64_t)Skv * ldv; gp.w_b2 = d; gp.c_b1 = per; gp.c_b2 = pb2;
    if (int rc = launch_gemm32(gp, st)) return rc;
    hipLaunchKernelGGL(softmax32_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, Skv, Sp);
    // dQ = scale dS K ; dK = scale dS^T Q
    GemmArgs gq = attn32_gemm(Sq, d, Skv, dP, Sp, 1, k, 1, ldk, dQ + (int64_t)b0 * Sq * lddq, lddq, scale, nb, H);
    gq.a_b1 = per; gq.a_b2 = pb2; gq.w_b1 = (int64_t)Skv * ldk; gq.w_b2 = d; gq.c_b1 = (int64_t)Sq * lddq; gq.c_b2 = d;
    if (int rc = launch_gemm32(gq, st)) return rc;
    GemmArgs gk = attn32_gemm(Skv, d, Sq, dP, 1, Sp, q, 1, ldq, dK + (int64_t)b0 * Skv * lddk, lddk, scale, nb, H);
    gk.a_b1 = per; gk.a_b2 = pb2; gk.w_b1 = (int64_t)Sq * ldq; gk.w_b2 = d; gk.c_b1 = (int64_t)Skv * lddk; gk.c_b2 = d;
    if (int rc = launch_gemm32(gk, st)) return rc;
  }
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_groupnorm32_fwd(const float* x, const float* gamma, const float* beta, float* stats, float* y, int B, int HW, int C, int G,
                           float eps, int silu, hipStream_t st) {
  FDMI_CHECK(C % G == 0, "groupnorm32: C % G != 0");
  hipLaunchKernelGGL(gn32_fwd_kernel, dim3(G, B), dim3(256), 0, st, x, gamma, beta, stats, y, HW, C, G, eps, silu);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_groupnorm32_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* stats, float* dx,
                           int B, int HW, int C, int G, int silu, int accumulate, hipStream_t st) {
  FDMI_CHECK(C % G == 0, "groupnorm32: C % G != 0");
  hipLaunchKernelGGL(gn32_bwd_kernel, dim3(G, B), dim3(256), 0, st, x, dy, gamma, beta, stats, dx, HW, C, G, silu, accumulate);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_layernorm32_fwd(const float* x, const float* gamma, const float* beta, const float* shift, const float* scale,
                           int64_t mod_ld, int rows_per_batch, float* y, int64_t rows, int C, float eps, hipStream_t st, float* stats) {
  hipLaunchKernelGGL(ln32_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, gamma, beta, shift, scale, mod_ld,
                     rows_per_batch > 0 ? rows_per_batch : 1, y, rows, C, eps, stats);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_layernorm32_bwd(const float* x, const float* dy, const float* gamma, const float* scale, int64_t mod_ld, int rows_per_batch,
                           float* dx, int64_t rows, int C, float eps, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(ln32_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, dy, gamma, scale, mod_ld,
                     rows_per_batch > 0 ? rows_per_batch : 1, dx, rows, C, eps, accumulate);
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_nchw_to_nhwc32(const float* x, float* y, int B, int C, int HW, int Cpad, hipStream_t st) {
  LAUNCH32(nchw_to_nhwc32_kernel, (int64_t)B * HW * Cpad, x, y, B, C, HW, Cpad)
}
int launch_nhwc_to_nchw32(const float* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate, hipStream_t st) {
  LAUNCH32(nhwc_to_nchw32_kernel, (int64_t)B * C * HW, x, ldx, y, B, C, HW, accumulate)
}
int launch_add_nchw_to_nhwc32(const float* r, float scale, float* y, int B, int C, int HW, hipStream_t st) {
  LAUNCH32(add_nchw_to_nhwc32_kernel, (int64_t)B * HW * C, r, scale, y, B, C, HW)
}
int launch_timestep_embed32(const float* t, float* out, int B, int dim, int flip, float shift, hipStream_t st) {
  LAUNCH32(timestep_embed32_kernel, (int64_t)B * (dim / 2), t, out, B, dim, flip, shift)
}
int launch_silu32(const float* x, float* y, int64_t n, hipStream_t st) { LAUNCH32(silu32_kernel, n, x, y, n) }
int launch_silu32_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st) { LAUNCH32(silu32_bwd_kernel, n, x, dy, dx, n) }
int launch_gelu_tanh32(const float* x, float* y, int64_t n, hipStream_t st) { LAUNCH32(gelu_tanh32_kernel, n, x, y, n) }
int launch_gelu_tanh32_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st) {
  LAUNCH32(gelu_tanh32_bwd_kernel, n, x, dy, dx, n)
}
int launch_copy2d32(const float* src, int64_t lds, int sc0, float* dst, int64_t ldd, int dc0, int64_t rows, int cols, int accumulate,
                    hipStream_t st) {
  LAUNCH32(copy2d32_kernel, rows * cols, src, lds, sc0, dst, ldd, dc0, rows, cols, accumulate)
}
int launch_pool2x2_sum32(const float* dy, float* dx, int B, int H, int W, int C, int accumulate, hipStream_t st) {
  LAUNCH32(pool2x2_sum32_kernel, (int64_t)B * H * W * C, dy, dx, B, H, W, C, accumulate)
}
int launch_geglu32_bwd(const float* pre, const float* dout, float* dpre, int64_t M, int F, hipStream_t st) {
  FDMI_CHECK(F % 16 == 0, "geglu32_bwd: F % 16 != 0");
  LAUNCH32(geglu32_bwd_kernel, M * F, pre, dout, dpre, M, F)
}
int launch_pad_cols32(const float* src, int cols, float* dst, int cols_pad, int64_t rows, hipStream_t st) {
  LAUNCH32(pad_cols32_kernel, rows * cols_pad, src, cols, dst, cols_pad, rows)
}
int launch_im2col32(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                    hipStream_t st) {
  LAUNCH32(im2col32_kernel, (int64_t)B * Ho * Wo * KH * KW * C, x, out, B, H, W, C, Ho, Wo, KH, KW, stride, pad)
}
int launch_colsum32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int64_t rows, int C, int HW, int G,
                    hipStream_t st) {
  int by = (int)((rows + 63) / 64);
  if (by > 256) by = 256;
  if (by < 1) by = 1;
  hipLaunchKernelGGL(colsum32_kernel, dim3(cdiv(C, 64), by), dim3(256), 0, st, dy, x, stats, out0, out1, rows, C, HW > 0 ? HW : 1,
                     G > 0 ? G : 1);
  FDMI_HIP(hipGetLastError());
  return 0;
}
int launch_gate_residual32(const float* x, const float* gate, int64_t gate_ld, const float* res, float* y, int64_t rows, int C,
                           int rows_per_batch, hipStream_t st) {
  LAUNCH32(gate_residual32_kernel, rows * C, x, gate, gate_ld, res, y, rows, C, rows_per_batch)
}
int launch_batch_colsum32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int B, int rows_per_batch,
                          int C, hipStream_t st) {
  hipLaunchKernelGGL(batch_colsum32_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, st, dy, x, stats, out0, out1, rows_per_batch, C);
  FDMI_HIP(hipGetLastError());
  return 0;
}
