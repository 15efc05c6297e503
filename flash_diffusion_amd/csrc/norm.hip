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
  // x as a channel concatenation that was never materialised (the UNet's up-path [h | skip]): channels c < C1 are read from
  // x[row][C1], channels c >= C1 from x2[row][C - C1] (x2 == nullptr: one tensor of C channels).  C1 % 8 == 0; a thread's
  // channel chunk is fixed, so the choice is made once per thread.  dy / y always span all C channels.
  const bf16_t* x2; int C1;
};
// this thread's x source for channel chunk c0: pointer to (row 0, channel c0) and the row stride
__device__ __forceinline__ const bf16_t* gn_xsrc(const GnArgs& a, int c0, int64_t& xld) {
  if (a.x2 && c0 >= a.C1) { xld = a.C - a.C1; return a.x2 + (c0 - a.C1); }
  xld = a.x2 ? a.C1 : a.C;
  return a.x + c0;
}

// UNR: rows fetched per loop trip before any of them is consumed (UNR 16-byte loads in flight per thread instead of one; a
// plain loop waits for each load before issuing the next: round 1 measured that at 1.6 TB/s, 20 % of the HBM roofline).
template <bool BWD, int UNR = 4>
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
    int64_t xld;
    const bf16_t* xp = gn_xsrc(a, c0, xld);
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
    auto consume = [&](const u16x8& xv, const u16x8& dv) {
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = bf2f(xv[e]);
          s0[e] += v;
          s1[e] += v * v;
        }
      } else {
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
    };
    int r = row0 + roff;
    if constexpr (UNR > 1) {
      for (; r + (UNR - 1) * R < row1; r += UNR * R) {
        u16x8 xv[UNR], dv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int64_t row = (int64_t)b * a.HW + r + u * R;
          xv[u] = *(const u16x8*)(xp + row * xld);
          if (BWD) dv[u] = *(const u16x8*)(a.dy + row * a.C + c0);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) consume(xv[u], BWD ? dv[u] : xv[u]);
      }
    }
    for (; r < row1; r += R) {
      const int64_t row = (int64_t)b * a.HW + r;
      const u16x8 xv = *(const u16x8*)(xp + row * xld);
      const u16x8 dv = BWD ? *(const u16x8*)(a.dy + row * a.C + c0) : xv;
      consume(xv, dv);
    }
    if (cpg >= 8) {
      // a chunk of 8 channels touches at most two groups when a group is >= 8 channels wide: the thread folds its 8 partial
      // sums into those two before the LDS atomics (4 instead of 16 same-address `ds_add_f32` per thread: -2 ms of the C2 step)
      const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;   // elements e < split belong to g0, the rest to g0 + 1
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e < split) { a0 += s0[e]; a1 += s1[e]; }
        else { b0 += s0[e]; b1 += s1[e]; }
      }
      atomicAdd(&sred[g0][0], a0);
      atomicAdd(&sred[g0][1], a1);
      if (split < 8) {
        atomicAdd(&sred[g0 + 1][0], b0);
        atomicAdd(&sred[g0 + 1][1], b1);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int gi = (c0 + e) / cpg;
        atomicAdd(&sred[gi][0], s0[e]);
        atomicAdd(&sred[gi][1], s1[e]);
      }
    }
  }
  __syncthreads();
  float* dst = BWD ? a.bstats : a.stats;
  if (tid < a.G * 2) atomicAdd(dst + (int64_t)b * a.G * 2 + tid, sred[tid >> 1][tid & 1]);
}

// Deterministic twin of gn_reduce_kernel (fdmi_det(), common.h): ONE block per sample walks all HW rows with the same thread ->
// (channel chunk, row lane) assignment, every thread's sums run in row order, and the block's reduction is ORDERED: the per-thread
// partial sums of a slot go to LDS and the 2 G reducer threads add the contributions to their (group, component) in thread order,
// element order -- no LDS atomics, no global atomics (the block is the sample's only contributor).  A test mode: ~B blocks per launch.
template <bool BWD>
__global__ __launch_bounds__(256) void gn_reduce_det_kernel(GnArgs a) {
  __shared__ float part[256][16];   // s0[8], s1[8] of each thread for the current slot
  __shared__ int pc0[256];          // first channel of the thread's chunk (-1: idle in this slot)
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int CPR = a.C >> 3, cpg = a.C / a.G;
  const int R = CPR <= 256 ? 256 / CPR : 1;
  const int nslot = (CPR + 255) >> 8;
  const float inv_n = 1.f / ((float)a.HW * cpg);
  float tot = 0.f;                  // reducer thread t < 2 G: the running sum of (group t >> 1, component t & 1)
  for (int s = 0; s < nslot; ++s) {
    int chunk, roff;
    bool active = true;
    if (CPR <= 256) {
      chunk = tid % CPR;
      roff = tid / CPR;
      active = roff < R;
    } else {
      chunk = tid + 256 * s;
      roff = 0;
      active = chunk < CPR;
    }
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    if (active) {
      const int c0 = chunk * 8;
      int64_t xld;
      const bf16_t* xp = gn_xsrc(a, c0, xld);
      float mean[8], rstd[8], gm[8], bt[8];
      if (BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int gi = (c0 + e) / cpg;
          const float sm = a.stats[((int64_t)b * a.G + gi) * 2 + 0] * inv_n;
          const float sq = a.stats[((int64_t)b * a.G + gi) * 2 + 1] * inv_n;
          mean[e] = sm;
          rstd[e] = rsqrtf(fmaxf(sq - sm * sm, 0.f) + a.eps);
          gm[e] = a.gamma[c0 + e];
          bt[e] = a.beta[c0 + e];
        }
      }
      for (int r = roff; r < a.HW; r += R) {
        const int64_t row = (int64_t)b * a.HW + r;
        const u16x8 xv = *(const u16x8*)(xp + row * xld);
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = bf2f(xv[e]);
            s0[e] += v;
            s1[e] += v * v;
          }
        } else {
          const u16x8 dv = *(const u16x8*)(a.dy + row * a.C + c0);
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
    }
    __syncthreads();   // (the previous slot's reducers are done with part / pc0)
    pc0[tid] = active ? chunk * 8 : -1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      part[tid][e] = s0[e];
      part[tid][8 + e] = s1[e];
    }
    __syncthreads();
    if (tid < a.G * 2) {
      const int gi = tid >> 1, comp = tid & 1;
      for (int t = 0; t < 256; ++t) {
        const int c0 = pc0[t];
        if (c0 < 0 || (c0 + 7) / cpg < gi || c0 / cpg > gi) continue;
        for (int e = 0; e < 8; ++e)
          if ((c0 + e) / cpg == gi) tot += part[t][comp * 8 + e];
      }
    }
  }
  float* dst = BWD ? a.bstats : a.stats;
  if (tid < a.G * 2) dst[(int64_t)b * a.G * 2 + tid] += tot;   // (the only contributor of this sample's sums)
}

// apply pass: thread t owns channel chunk t % CPR for rows t / CPR, +R, +2R, ... of its row block, so
// the per-channel affine coefficients (mean/rstd/gamma/beta -> a, b) are derived ONCE per thread and
// the row loop is a 16-byte load, 8 fmas (+SiLU) and a 16-byte store.
// UNR (forward only): rows fetched per trip before the first store -- x and y may alias as far as the compiler knows, so a
// plain loop cannot start row r+1's load before row r's store
template <bool BWD, int UNR = 4>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a, int rows_per_block) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int CPR = a.C >> 3, cpg = a.C / a.G;
  const int R = CPR <= 256 ? 256 / CPR : 1;
  const int nslot = (CPR + 255) >> 8;
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
    int64_t xld;
    const bf16_t* xp = gn_xsrc(a, c0, xld);
    float ca[8], cb[8], gm[8], bt[8], m1[8], m2[8], mean[8], rstd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int gi = (c0 + e) / cpg;
      const float sm = a.stats[((int64_t)b * a.G + gi) * 2 + 0] * inv_n;
      const float sq = a.stats[((int64_t)b * a.G + gi) * 2 + 1] * inv_n;
      mean[e] = sm;
      rstd[e] = rsqrtf(fmaxf(sq - sm * sm, 0.f) + a.eps);
      gm[e] = a.gamma[c0 + e];
      bt[e] = a.beta[c0 + e];
      ca[e] = rstd[e] * gm[e];            // z = ca * x + cb
      cb[e] = bt[e] - sm * ca[e];
      if (BWD) {
        m1[e] = a.bstats[((int64_t)b * a.G + gi) * 2 + 0] * inv_n;
        m2[e] = a.bstats[((int64_t)b * a.G + gi) * 2 + 1] * inv_n;
      }
    }
    int r = row0 + roff;
    if constexpr (!BWD && UNR > 1) {
      for (; r + (UNR - 1) * R < row1; r += UNR * R) {
        u16x8 xu[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) xu[u] = *(const u16x8*)(xp + ((int64_t)b * a.HW + r + u * R) * xld);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          u16x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float z = fmaf(ca[e], bf2f(xu[u][e]), cb[e]);
            if (a.silu) z = silu_f(z);
            ov[e] = f2bf(z);
          }
          *(u16x8*)(a.y + ((int64_t)b * a.HW + r + u * R) * a.C + c0) = ov;
        }
      }
    }
    for (; r < row1; r += R) {
      const int64_t off = ((int64_t)b * a.HW + r) * a.C + c0;
      const u16x8 xv = *(const u16x8*)(xp + ((int64_t)b * a.HW + r) * xld);
      u16x8 ov;
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = fmaf(ca[e], bf2f(xv[e]), cb[e]);
          if (a.silu) z = silu_f(z);
          ov[e] = f2bf(z);
        }
      } else {
        const u16x8 dv = *(const u16x8*)(a.dy + off);
        u16x8 prev;
        if (a.accumulate) prev = *(const u16x8*)(a.y + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (bf2f(xv[e]) - mean[e]) * rstd[e];
          float dz = bf2f(dv[e]);
          if (a.silu) dz *= dsilu_f(fmaf(gm[e], xh, bt[e]));
          const float dxh = dz * gm[e];
          float dx = rstd[e] * (dxh - m1[e] - xh * m2[e]);
          if (a.accumulate) dx += bf2f(prev[e]);
          ov[e] = f2bf(dx);
        }
      }
      *(u16x8*)(a.y + off) = ov;
    }
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
                         hipStream_t st, bool stats_zeroed, bool stats_ready, const bf16_t* x2, int C1) {
  FDMI_CHECK(C % 8 == 0 && C % G == 0 && G <= 32 && C <= 4096, "groupnorm: unsupported C/G");
  FDMI_CHECK(!x2 || (C1 > 0 && C1 < C && C1 % 8 == 0), "groupnorm: a two-part input needs 0 < C1 < C, C1 % 8 == 0");
  GnArgs a{x, nullptr, gamma, beta, stats, nullptr, y, B, HW, C, G, eps, silu, 0, x2, x2 ? C1 : 0};
  const int rpb = gn_rows_per_block(B, HW);
  if (!stats_ready) {
    if (!stats_zeroed) FDMI_HIP(hipMemsetAsync(stats, 0, (size_t)B * G * 2 * sizeof(float), st));
    if (fdmi_det()) hipLaunchKernelGGL(gn_reduce_det_kernel<false>, dim3(B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gn_reduce_kernel<false>, dim3(cdiv(HW, rpb), B), dim3(256), 0, st, a, rpb);
  }
  hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(cdiv(HW, rpb), B), dim3(256), 0, st, a, rpb);
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_groupnorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* beta,
                         const float* stats, float* bstats, bf16_t* dx, int B, int HW, int C, int G,
                         float eps, int silu, int accumulate, hipStream_t st, bool stats_zeroed, const bf16_t* x2, int C1) {
  FDMI_CHECK(C % 8 == 0 && C % G == 0 && G <= 32 && C <= 4096, "groupnorm: unsupported C/G");
  FDMI_CHECK(!x2 || (C1 > 0 && C1 < C && C1 % 8 == 0), "groupnorm: a two-part input needs 0 < C1 < C, C1 % 8 == 0");
  GnArgs a{x, dy, gamma, beta, (float*)stats, bstats, dx, B, HW, C, G, eps, silu, accumulate, x2, x2 ? C1 : 0};
  if (!stats_zeroed) FDMI_HIP(hipMemsetAsync(bstats, 0, (size_t)B * G * 2 * sizeof(float), st));
  const int rpb = gn_rows_per_block(B, HW);
  if (fdmi_det()) hipLaunchKernelGGL(gn_reduce_det_kernel<true>, dim3(B), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gn_reduce_kernel<true>, dim3(cdiv(HW, rpb), B), dim3(256), 0, st, a, rpb);
  hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(cdiv(HW, rpb), B), dim3(256), 0, st, a, rpb);
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
  float* stats;  // fwd, optional: [rows][2] = (mean, rstd) for the adaLN parameter gradients (dit.hip batch_colsum)
  const float* x32;  // optional: the fp32 master of x (the transformer denoisers' residual stream, round 5): read instead of x
};

// LPR lanes cooperate on one row (LPR = 8..64, a power of two chosen so each lane holds <= 5 chunks of
// 8 channels); a wavefront therefore streams 64/LPR rows at once with ~5 KB of loads in flight.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool BWD, int LPR>
__global__ __launch_bounds__(256) void ln_kernel(LnArgs a) {
  constexpr int RPW = 64 / LPR;  // rows per wavefront
  constexpr int MAXC = 5;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < a.rows;
  const int64_t rrow = live ? row : 0;
  const int CPR = a.C >> 3;
  const float invC = 1.f / a.C;
  float xv[MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    const int ch = sub + LPR * s;
    if (ch < CPR) {
      if (a.x32) {   // (uniform)
        const float4 v0 = *(const float4*)(a.x32 + rrow * a.C + ch * 8), v1 = *(const float4*)(a.x32 + rrow * a.C + ch * 8 + 4);
        xv[s][0] = v0.x; xv[s][1] = v0.y; xv[s][2] = v0.z; xv[s][3] = v0.w; xv[s][4] = v1.x; xv[s][5] = v1.y; xv[s][6] = v1.z; xv[s][7] = v1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[s][e];
      } else {
        const u16x8 v = *(const u16x8*)(a.x + rrow * a.C + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xv[s][e] = bf2f(v[e]);
          sum += xv[s][e];
        }
      }
    }
  }
  const float mean = group_sum<LPR>(sum) * invC;
  float sq = 0.f;
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    const int ch = sub + LPR * s;
    if (ch < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = xv[s][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(group_sum<LPR>(sq) * invC + a.eps);
  const int64_t mrow = a.scale ? (rrow / a.rows_per_batch) * a.mod_ld : 0;
  if (!BWD) {
    if (a.stats && live && sub == 0) {
      a.stats[row * 2] = mean;
      a.stats[row * 2 + 1] = rstd;
    }
#pragma unroll
    for (int s = 0; s < MAXC; ++s) {
      const int ch = sub + LPR * s;
      if (ch < CPR && live) {
        u16x8 o;
        const int c0 = ch * 8;
        float gmv[8], btv[8];
        if (a.gamma) {
          const float4 g0 = *(const float4*)(a.gamma + c0), g1 = *(const float4*)(a.gamma + c0 + 4);
          const float4 b0 = *(const float4*)(a.beta + c0), b1 = *(const float4*)(a.beta + c0 + 4);
          gmv[0] = g0.x; gmv[1] = g0.y; gmv[2] = g0.z; gmv[3] = g0.w; gmv[4] = g1.x; gmv[5] = g1.y; gmv[6] = g1.z; gmv[7] = g1.w;
          btv[0] = b0.x; btv[1] = b0.y; btv[2] = b0.z; btv[3] = b0.w; btv[4] = b1.x; btv[5] = b1.y; btv[6] = b1.z; btv[7] = b1.w;
        }
        u16x8 scv = {0, 0, 0, 0, 0, 0, 0, 0}, shv = scv;   // (one 16-byte load each: launch_ln checks the alignment)
        if (a.scale) {
          scv = *(const u16x8*)(a.scale + mrow + c0);
          shv = *(const u16x8*)(a.shift + mrow + c0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = (xv[s][e] - mean) * rstd;
          if (a.gamma) v = fmaf(v, gmv[e], btv[e]);
          if (a.scale) v = fmaf(v, 1.f + bf2f(scv[e]), bf2f(shv[e]));
          o[e] = f2bf(v);
        }
        *(u16x8*)(a.y + row * a.C + c0) = o;
      }
    }
  } else {
    float dxh[MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < MAXC; ++s) {
      const int ch = sub + LPR * s;
      if (ch < CPR) {
        const u16x8 d = *(const u16x8*)(a.dy + rrow * a.C + ch * 8);
        float gmv[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (a.gamma) {
          const float4 g0 = *(const float4*)(a.gamma + ch * 8), g1 = *(const float4*)(a.gamma + ch * 8 + 4);
          gmv[0] = g0.x; gmv[1] = g0.y; gmv[2] = g0.z; gmv[3] = g0.w; gmv[4] = g1.x; gmv[5] = g1.y; gmv[6] = g1.z; gmv[7] = g1.w;
        }
        u16x8 scv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (a.scale) scv = *(const u16x8*)(a.scale + mrow + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gmul = gmv[e];
          if (a.scale) gmul *= (1.f + bf2f(scv[e]));
          dxh[s][e] = bf2f(d[e]) * gmul;
          const float xh = (xv[s][e] - mean) * rstd;
          s1 += dxh[s][e];
          s2 += dxh[s][e] * xh;
        }
      }
    }
    const float m1 = group_sum<LPR>(s1) * invC, m2 = group_sum<LPR>(s2) * invC;
#pragma unroll
    for (int s = 0; s < MAXC; ++s) {
      const int ch = sub + LPR * s;
      if (ch < CPR && live) {
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

template <bool BWD>
static int launch_ln(const LnArgs& a, hipStream_t st) {
  FDMI_CHECK(((uintptr_t)a.scale & 15) == 0 && ((uintptr_t)a.shift & 15) == 0 && (!a.scale || (a.mod_ld & 7) == 0),
             "layernorm: the modulation vectors must be 16-byte aligned (column blocks of a [B][k C] tensor, C % 8 == 0)");
  const int CPR = a.C >> 3;
  int lpr = 8;
  while (lpr < 64 && lpr * 5 < CPR) lpr <<= 1;
  const int rpw = 64 / lpr;
  const unsigned blocks = (unsigned)((a.rows + 4 * rpw - 1) / (4 * rpw));
  switch (lpr) {
    case 8: hipLaunchKernelGGL((ln_kernel<BWD, 8>), dim3(blocks), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((ln_kernel<BWD, 16>), dim3(blocks), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((ln_kernel<BWD, 32>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((ln_kernel<BWD, 64>), dim3(blocks), dim3(256), 0, st, a); break;
  }
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_layernorm_fwd(const bf16_t* x, const float* gamma, const float* beta, const bf16_t* shift,
                         const bf16_t* scale, int64_t mod_ld, int rows_per_batch, bf16_t* y,
                         int64_t rows, int C, float eps, hipStream_t st, float* stats, const float* x32) {
  FDMI_CHECK(C % 8 == 0 && C <= 2560, "layernorm: C must be a multiple of 8 and <= 2560");
  FDMI_CHECK(!scale || (shift && rows_per_batch > 0 && mod_ld % 8 == 0), "layernorm: adaLN modulate needs shift, scale, rows_per_batch");
  FDMI_CHECK(((uintptr_t)x32 & 15) == 0, "layernorm: the fp32 input must be 16-byte aligned");
  LnArgs a{x, nullptr, gamma, beta, shift, scale, mod_ld, rows_per_batch, y, rows, C, eps, 0, stats, x32};
  return launch_ln<false>(a, st);
}

int launch_layernorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const bf16_t* scale,
                         int64_t mod_ld, int rows_per_batch, bf16_t* dx, int64_t rows, int C,
                         float eps, int accumulate, hipStream_t st, const float* x32) {
  FDMI_CHECK(C % 8 == 0 && C <= 2560, "layernorm: C must be a multiple of 8 and <= 2560");
  FDMI_CHECK(((uintptr_t)x32 & 15) == 0, "layernorm: the fp32 input must be 16-byte aligned");
  LnArgs a{x, dy, gamma, nullptr, nullptr, scale, mod_ld, rows_per_batch, dx, rows, C, eps, accumulate, nullptr, x32};
  return launch_ln<true>(a, st);
}
