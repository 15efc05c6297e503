// GroupNorm(+SiLU) and LayerNorm, forward and input-gradient, on NHWC / token-major bf16 tensors.
// HBM-bound kernels: 16-byte (8 x bf16) vector accesses, fp32 statistics, wavefront (64-lane)
// reductions.  The affine parameters are frozen on the distillation path, so no dgamma/dbeta.
#include "ops.h"

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics: x[B][HW][C] -> stats[B][G][2] (sum, sum of squares), accumulated with
// atomics (stats must be zeroed).  Thread t owns channel chunk(s) t % CPR (+256) so every
// thread's channels are fixed across rows and partial sums stay in registers.
// For the backward pass the same kernel shape reduces (sum dxhat, sum dxhat*xhat).
// ---------------------------------------------------------------------------------------------
struct GnArgs {
  const bf16_t* x; const bf16_t* dy; const float* gamma; const float* beta;
  float* stats;        // [B][G][2] forward sums
  float* bstats;       // [B][G][2] backward sums
  bf16_t* y;           // forward output / dx output
  int B, HW, C, G; float eps; int silu; int accumulate;
};

template <bool BWD>
__global__ __launch_bounds__(256) void gn_reduce_kernel(GnArgs a, int rows_per_block) {
  __shared__ float sred[64][2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int CPR = a.C >> 3, cpg = a.C / a.G;
  const int R = CPR <= 256 ? 256 / CPR : 1;
  const int nslot = (CPR + 255) >> 8;
  if (tid < 64) sred[tid][0] = sred[tid][1] = 0.f;
  __syncthreads();
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(a.HW, row0 + rows_per_block);
  const float inv_n = 1.f / ((float)a.HW * cpg);
  for (int s = 0; s < nslot; ++s) {
    int chunk, roff;
    if (CPR <= 256) {
      chunk = tid % CPR;
      roff = tid / CPR;
      if (roff >= R) continue;
    } else {
      chunk = tid + 256 * s;
      roff = 0;
      if (chunk >= CPR) continue;
    }
    const int c0 = chunk * 8;
    float s0[8], s1[8], mean[8], rstd[8], gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s0[e] = s1[e] = 0.f;
      if (BWD) {
        const int gi = (c0 + e) / cpg;
        const float sm = a.stats[((int64_t)b * a.G + gi) * 2 + 0] * inv_n;
        const float sq = a.stats[((int64_t)b * a.G + gi) * 2 + 1] * inv_n;
        mean[e] = sm;
        rstd[e] = rsqrtf(fmaxf(sq - sm * sm, 0.f) + a.eps);
        gm[e] = a.gamma[c0 + e];
        bt[e] = a.beta[c0 + e];
      }
    }
    for (int r = row0 + roff; r < row1; r += R) {
      const int64_t off = ((int64_t)b * a.HW + r) * a.C + c0;
      const u16x8 xv = *(const u16x8*)(a.x + off);
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = bf2f(xv[e]);
          s0[e] += v;
          s1[e] += v * v;
        }
      } else {
        const u16x8 dv = *(const u16x8*)(a.dy + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (bf2f(xv[e]) - mean[e]) * rstd[e];
          float dz = bf2f(dv[e]);
          if (a.silu) dz *= dsilu_f(gm[e] * xh + bt[e]);
          const float dxh = dz * gm[e];
          s0[e] += dxh;
          s1[e] += dxh * xh;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int gi = (c0 + e) / cpg;
      atomicAdd(&sred[gi][0], s0[e]);
      atomicAdd(&sred[gi][1], s1[e]);
    }
  }
  __syncthreads();
  float* dst = BWD ? a.bstats : a.stats;
  if (tid < a.G * 2) atomicAdd(dst + (int64_t)b * a.G * 2 + tid, sred[tid >> 1][tid & 1]);
}

template <bool BWD>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a) {
  const int CPR = a.C >> 3, cpg = a.C / a.G;
  const int64_t total = (int64_t)a.B * a.HW * CPR;
  const float inv_n = 1.f / ((float)a.HW * cpg);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / CPR;
    const int c0 = (int)(i - row * CPR) * 8;
    const int b = (int)(row / a.HW);
    const u16x8 xv = *(const u16x8*)(a.x + row * a.C + c0);
    u16x8 dv, ov, prev;
    if (BWD) {
      dv = *(const u16x8*)(a.dy + row * a.C + c0);
      if (a.accumulate) prev = *(const u16x8*)(a.y + row * a.C + c0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int gi = (c0 + e) / cpg;
      const float sm = a.stats[((int64_t)b * a.G + gi) * 2 + 0] * inv_n;
      const float sq = a.stats[((int64_t)b * a.G + gi) * 2 + 1] * inv_n;
      const float rstd = rsqrtf(fmaxf(sq - sm * sm, 0.f) + a.eps);
      const float xh = (bf2f(xv[e]) - sm) * rstd;
      const float gm = a.gamma[c0 + e], bt = a.beta[c0 + e];
      if (!BWD) {
        float z = gm * xh + bt;
        if (a.silu) z = silu_f(z);
        ov[e] = f2bf(z);
      } else {
        float dz = bf2f(dv[e]);
        if (a.silu) dz *= dsilu_f(gm * xh + bt);
        const float dxh = dz * gm;
        const float m1 = a.bstats[((int64_t)b * a.G + gi) * 2 + 0] * inv_n;
        const float m2 = a.bstats[((int64_t)b * a.G + gi) * 2 + 1] * inv_n;
        float dx = rstd * (dxh - m1 - xh * m2);
        if (a.accumulate) dx += bf2f(prev[e]);
        ov[e] = f2bf(dx);
      }
    }
    *(u16x8*)(a.y + row * a.C + c0) = ov;
  }
}

static int gn_rows_per_block(int B, int HW) {
  // aim for >= ~1024 blocks while keeping >= 8 rows per block
  int rpb = (int)(((int64_t)B * HW + 1023) / 1024);
  if (rpb < 8) rpb = 8;
  if (rpb > HW) rpb = HW;
  return rpb;
}

int launch_groupnorm_fwd(const bf16_t* x, const float* gamma, const float* beta, float* stats,
                         bf16_t* y, int B, int HW, int C, int G, float eps, int silu,
                         hipStream_t st) {
  FDMI_CHECK(C % 8 == 0 && C % G == 0 && G <= 32 && C <= 4096, "groupnorm: unsupported C/G");
  GnArgs a{x, nullptr, gamma, beta, stats, nullptr, y, B, HW, C, G, eps, silu, 0};
  FDMI_HIP(hipMemsetAsync(stats, 0, (size_t)B * G * 2 * sizeof(float), st));
  const int rpb = gn_rows_per_block(B, HW);
  hipLaunchKernelGGL(gn_reduce_kernel<false>, dim3(cdiv(HW, rpb), B), dim3(256), 0, st, a, rpb);
  const int64_t total = (int64_t)B * HW * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(blocks), dim3(256), 0, st, a);
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_groupnorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* beta,
                         const float* stats, float* bstats, bf16_t* dx, int B, int HW, int C, int G,
                         float eps, int silu, int accumulate, hipStream_t st) {
  FDMI_CHECK(C % 8 == 0 && C % G == 0 && G <= 32 && C <= 4096, "groupnorm: unsupported C/G");
  GnArgs a{x, dy, gamma, beta, (float*)stats, bstats, dx, B, HW, C, G, eps, silu, accumulate};
  FDMI_HIP(hipMemsetAsync(bstats, 0, (size_t)B * G * 2 * sizeof(float), st));
  const int rpb = gn_rows_per_block(B, HW);
  hipLaunchKernelGGL(gn_reduce_kernel<true>, dim3(cdiv(HW, rpb), B), dim3(256), 0, st, a, rpb);
  const int64_t total = (int64_t)B * HW * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(blocks), dim3(256), 0, st, a);
  FDMI_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, row held in registers (C <= 2048), two-pass statistics.
//   fwd: y = (x-mean)*rstd*gamma + beta          (gamma/beta may be null: non-affine)
//        optional adaLN modulate: y = y*(1+scale[b]) + shift[b]   (DiT)
//   bwd: dx = rstd*(dxh - mean(dxh) - xh*mean(dxh*xh)),  dxh = dy*gamma*(1+scale)
// ---------------------------------------------------------------------------------------------
struct LnArgs {
  const bf16_t* x; const bf16_t* dy; const float* gamma; const float* beta;
  const bf16_t* shift; const bf16_t* scale; int64_t mod_ld; int rows_per_batch;
  bf16_t* y; int64_t rows; int C; float eps; int accumulate;
};

template <bool BWD>
__global__ __launch_bounds__(256) void ln_kernel(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int CPR = a.C >> 3;
  constexpr int MAXC = 4;
  float xv[MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    const int ch = lane + 64 * s;
    if (ch < CPR) {
      const u16x8 v = *(const u16x8*)(a.x + row * a.C + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[s][e] = bf2f(v[e]);
        sum += xv[s][e];
      }
    }
  }
  const float mean = wave_sum(sum) / a.C;
  float sq = 0.f;
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    const int ch = lane + 64 * s;
    if (ch < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = xv[s][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / a.C + a.eps);
  const int64_t mrow = a.scale ? (row / a.rows_per_batch) * a.mod_ld : 0;
  if (!BWD) {
#pragma unroll
    for (int s = 0; s < MAXC; ++s) {
      const int ch = lane + 64 * s;
      if (ch < CPR) {
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = ch * 8 + e;
          float v = (xv[s][e] - mean) * rstd;
          if (a.gamma) v = v * a.gamma[c] + a.beta[c];
          if (a.scale) v = v * (1.f + bf2f(a.scale[mrow + c])) + bf2f(a.shift[mrow + c]);
          o[e] = f2bf(v);
        }
        *(u16x8*)(a.y + row * a.C + ch * 8) = o;
      }
    }
  } else {
    float dxh[MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < MAXC; ++s) {
      const int ch = lane + 64 * s;
      if (ch < CPR) {
        const u16x8 d = *(const u16x8*)(a.dy + row * a.C + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = ch * 8 + e;
          float gmul = a.gamma ? a.gamma[c] : 1.f;
          if (a.scale) gmul *= (1.f + bf2f(a.scale[mrow + c]));
          dxh[s][e] = bf2f(d[e]) * gmul;
          const float xh = (xv[s][e] - mean) * rstd;
          s1 += dxh[s][e];
          s2 += dxh[s][e] * xh;
        }
      }
    }
    const float m1 = wave_sum(s1) / a.C, m2 = wave_sum(s2) / a.C;
#pragma unroll
    for (int s = 0; s < MAXC; ++s) {
      const int ch = lane + 64 * s;
      if (ch < CPR) {
        u16x8 o, prev;
        if (a.accumulate) prev = *(const u16x8*)(a.y + row * a.C + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[s][e] - mean) * rstd;
          float dx = rstd * (dxh[s][e] - m1 - xh * m2);
          if (a.accumulate) dx += bf2f(prev[e]);
          o[e] = f2bf(dx);
        }
        *(u16x8*)(a.y + row * a.C + ch * 8) = o;
      }
    }
  }
}

int launch_layernorm_fwd(const bf16_t* x, const float* gamma, const float* beta, const bf16_t* shift,
                         const bf16_t* scale, int64_t mod_ld, int rows_per_batch, bf16_t* y,
                         int64_t rows, int C, float eps, hipStream_t st) {
  FDMI_CHECK(C % 8 == 0 && C <= 2048, "layernorm: C must be a multiple of 8 and <= 2048");
  LnArgs a{x, nullptr, gamma, beta, shift, scale, mod_ld, rows_per_batch, y, rows, C, eps, 0};
  hipLaunchKernelGGL(ln_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_layernorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const bf16_t* scale,
                         int64_t mod_ld, int rows_per_batch, bf16_t* dx, int64_t rows, int C,
                         float eps, int accumulate, hipStream_t st) {
  FDMI_CHECK(C % 8 == 0 && C <= 2048, "layernorm: C must be a multiple of 8 and <= 2048");
  LnArgs a{x, dy, gamma, nullptr, nullptr, scale, mod_ld, rows_per_batch, dx, rows, C, eps, accumulate};
  hipLaunchKernelGGL(ln_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
  FDMI_HIP(hipGetLastError());
  return 0;
}
