// Hand-written bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[M,N] = A[M,K] * W[N,K]^T  (+ bias, + per-batch row vector, + residual, SiLU / GEGLU)
//
// * A operand: plain row-major activations, or an implicit im2col gather from an NHWC tensor
//   (3x3 / 4x4 / 1x1, stride 1|2, zero padding, optional fused nearest-2x upsample, and the
//   gather form of the transposed convolution for dgrad).  One 16-byte chunk = 8 channels.
// * Tiles: BM x BN x 64, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 16x16x32 bf16
//   MFMA fragments, fp32 accumulation.  Operands are swapped (D = Wfrag * Afrag^T) so that every
//   lane owns 4 consecutive output channels of one row -> 8-byte bf16 stores.
// * Staging: global -> LDS by LDS-DMA (global_load_lds, 16 B/lane), double buffered.  The DMA
//   destination is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
//   address (logical chunk = slot ^ ((row>>1)&7)) and again on the ds_read_b128 address.
//   Out-of-range rows / K tail / conv padding read a 16-byte zero page instead.
// * blockIdx -> tile mapping is XCD-aware (8 XCDs, private L2s): consecutive tiles (same A rows,
//   successive N tiles) are placed on one XCD.
#include "gemm.h"
#include <math.h>

static __device__ uint4 g_zero16[4] = {};

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ---------------------------------------------------------------------------------------------
// epilogue helpers (shared by the GEMM kernel and the split-K finalize kernel)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void epi_terms(const GemmArgs& a, int64_t m, int n, float v[4]) {
  // v = alpha*v + bias + rowvec + residual   (n..n+3, caller guarantees n+3 < N or handles tail)
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] *= a.alpha;
  if (a.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.N) v[r] += a.bias[n + r];
  }
  if (a.rowvec) {
    const bf16_t* rv = a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.N) v[r] = a.rowvec_mul ? v[r] * bf2f(rv[r]) : v[r] + bf2f(rv[r]);
  }
}

__device__ __forceinline__ void epi_store(const GemmArgs& a, int act, int64_t m, int nout, int Nout, float v[4]) {
  // + residual (indexed in OUTPUT columns), activation, store 4 consecutive output columns
  if (a.residual) {
    const bf16_t* rs = a.residual + m * a.ldr + nout;
    if (nout + 3 < Nout && ((a.ldr | nout) & 3) == 0) {
      u16x4 t = *(const u16x4*)rs;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bf2f(t[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nout + r < Nout) v[r] += bf2f(rs[r]);
    }
  }
  if (a.residual32) {   // the fp32 residual stream (gemm.h): N, ldr32 % 8 == 0 (gemm_r32_ok), whole 4-column groups
    const float4 t = *(const float4*)(a.residual32 + m * a.ldr32 + nout);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  if (a.C32) *(float4*)(a.C32 + m * a.ldc32 + nout) = make_float4(v[0], v[1], v[2], v[3]);
  if (!a.C) return;   // (an fp32-stream output without its bf16 shadow: gemm_r32_ok problems only -- no activation, bf16 C)
  if (act == ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
  } else if (act == ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  } else if (act == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
  } else if (act == ACT_GELU_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]);
  }
  if (a.out_f32) {
    float* c = (float*)a.C + m * a.ldc + nout;
    if (nout + 3 < Nout && ((a.ldc | nout) & 3) == 0) {
      *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nout + r < Nout) c[r] = v[r];
    }
  } else {
    bf16_t* c = (bf16_t*)a.C + m * a.ldc + nout;
    if (nout + 3 < Nout && ((a.ldc | nout) & 3) == 0) {
      uint2 pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      *(uint2*)c = pk;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nout + r < Nout) c[r] = f2bf(v[r]);
    }
  }
}

__device__ __forceinline__ void save_preact(const GemmArgs& a, int64_t m, int n, const float v[4]) {
  bf16_t* p = a.preact + m * a.ldp + n;
  if (n + 3 < a.N && ((a.ldp | n) & 3) == 0) {
    uint2 pk;
    pk.x = pack2bf(v[0], v[1]);
    pk.y = pack2bf(v[2], v[3]);
    *(uint2*)p = pk;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.N) p[r] = f2bf(v[r]);
  }
}

// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int MODE, bool GLDS>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int AROWS = BM / 32, WROWS = BN / 32;  // 16-B chunks per thread per K tile
  constexpr int STAGE = (BM + BN) * 128;           // bytes per pipeline stage
  constexpr int MF = BM / 32, NF = BN / 32;        // 16x16 fragments per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, j = lane & 15;

  // ---- XCD-aware tile mapping (bijective for any tile count) ----
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int T = tilesM * tilesN;
  int idx;
  {
    const int b = blockIdx.x, xcd = b & 7, q = T >> 3, r = T & 7;
    idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tn = idx % tilesN, tm = idx / tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- K range (split-K over blockIdx.y) ----
  int kbeg = 0, kend = a.K;
  if (a.splitk > 1) {
    const int ktiles = (a.K + 63) >> 6, per = (ktiles + a.splitk - 1) / a.splitk;
    kbeg = blockIdx.y * per * 64;
    kend = min(a.K, kbeg + per * 64);
    if (kbeg >= kend) return;
  }
  const int nk = (kend - kbeg + 63) >> 6;

  // ---- loader state: thread owns slot p of rows lr + 32*i; logical chunk c is row-group invariant
  const int p = tid & 7, lr = tid >> 3;
  const int c = p ^ ((lr >> 1) & 7);
  int kcur = kbeg + c * 8;
  int ky = 0, kx = 0, cc = 0;
  if (MODE == GEMM_CONV) {
    const int tap = kcur / a.Cin;
    cc = kcur - tap * a.Cin;
    ky = tap / a.KW;
    kx = tap - ky * a.KW;
  }
  const bf16_t* arow[AROWS];
  int abase_y[AROWS], abase_x[AROWS], apix[AROWS];
  bool avalid[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + lr + 32 * i;
    avalid[i] = m < a.M;
    if (MODE == GEMM_ROW) {
      arow[i] = a.A + (int64_t)(avalid[i] ? m : 0) * a.lda;
      abase_y[i] = abase_x[i] = apix[i] = 0;
    } else {
      const int hw = a.Hout * a.Wout;
      const int mm = avalid[i] ? m : 0;
      const int b = mm / hw, rem = mm - b * hw;
      const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
      arow[i] = a.A;
      apix[i] = b * a.Hin * a.Win;
      abase_y[i] = a.dgrad ? (oy + a.pad) : (oy * a.stride - a.pad);
      abase_x[i] = a.dgrad ? (ox + a.pad) : (ox * a.stride - a.pad);
    }
  }
  const bf16_t* wrow[WROWS];
  bool wvalid[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int n = n0 + lr + 32 * i;
    wvalid[i] = n < a.N;
    wrow[i] = a.W + (int64_t)(wvalid[i] ? n : 0) * a.ldw;
  }
  const bf16_t* zero = (const bf16_t*)g_zero16;
  uint4 areg[AROWS], wreg[WROWS];  // register staging (GLDS == false)

  auto src_a = [&](int i) -> const bf16_t* {
    if (!avalid[i] || kcur >= kend) return zero;
    if (MODE == GEMM_ROW) return arow[i] + kcur;
    if (ky >= a.KH) return zero;
    int sy, sx;
    if (a.dgrad) {
      const int ty = abase_y[i] - ky, tx = abase_x[i] - kx;
      if (ty < 0 || tx < 0) return zero;
      if (a.stride == 2) {
        if ((ty | tx) & 1) return zero;
        sy = ty >> 1;
        sx = tx >> 1;
      } else {
        sy = ty;
        sx = tx;
      }
      if (sy >= a.Hin || sx >= a.Win) return zero;
    } else {
      const int iy = abase_y[i] + ky, ix = abase_x[i] + kx;
      const int Hv = a.Hin << a.ups, Wv = a.Win << a.ups;
      if (iy < 0 || ix < 0 || iy >= Hv || ix >= Wv) return zero;
      sy = iy >> a.ups;
      sx = ix >> a.ups;
    }
    return arow[i] + ((int64_t)(apix[i] + sy * a.Win + sx) * a.Cin + cc);
  };
  auto src_w = [&](int i) -> const bf16_t* {
    if (!wvalid[i] || kcur >= kend) return zero;
    return wrow[i] + kcur;
  };
  auto advance = [&]() {
    kcur += 64;
    if (MODE == GEMM_CONV) {
      cc += 64;
      while (cc >= a.Cin) {
        cc -= a.Cin;
        if (++kx == a.KW) {
          kx = 0;
          ++ky;
        }
      }
    }
  };
  auto issue = [&](int stage) {  // global -> LDS (DMA) or global -> regs
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const bf16_t* s = src_a(i);
      if (GLDS) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)s,
            (LDS_AS void*)(smem + stage * STAGE + (wave * 64 + 256 * i) * 16), 16, 0, 0);
      } else {
        areg[i] = *(const uint4*)s;
      }
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const bf16_t* s = src_w(i);
      if (GLDS) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)s,
            (LDS_AS void*)(smem + stage * STAGE + BM * 128 + (wave * 64 + 256 * i) * 16), 16, 0, 0);
      } else {
        wreg[i] = *(const uint4*)s;
      }
    }
    advance();
  };
  auto commit = [&](int stage) {  // regs -> LDS (register staging only)
#pragma unroll
    for (int i = 0; i < AROWS; ++i) *(uint4*)(smem + stage * STAGE + (tid + 256 * i) * 16) = areg[i];
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      *(uint4*)(smem + stage * STAGE + BM * 128 + (tid + 256 * i) * 16) = wreg[i];
  };

  f32x4 acc[NF][MF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* sa = smem + stage * STAGE;
    const char* sw = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pc = (ks * 4 + g) ^ (j >> 1);  // swizzled slot of logical chunk ks*4+g
      bf16x8 af[MF], wf[NF];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
        af[mf] = *(const bf16x8*)(sa + ((wm * (BM / 2) + mf * 16 + j) * 8 + pc) * 16);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        wf[nf] = *(const bf16x8*)(sw + ((wn * (BN / 2) + nf * 16 + j) * 8 + pc) * 16);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], af[mf], acc[nf][mf], 0, 0, 0);
    }
  };

  // ---- main loop: 2-stage pipeline, one barrier per K tile ----
  int stage = 0;
  if (GLDS) {
    issue(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) issue(stage ^ 1);
      compute(stage);
      __syncthreads();
      stage ^= 1;
    }
  } else {
    issue(0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) issue(stage ^ 1);
      compute(stage);
      if (kt + 1 < nk) commit(stage ^ 1);
      __syncthreads();
      stage ^= 1;
    }
  }

  // ---- epilogue: lane (g, j) owns rows m = ..+j, columns n = ..+4g..4g+3 of each fragment ----
  if (a.accum_atomic) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = m0 + wm * (BM / 2) + mf * 16 + j;
        const int n = n0 + wn * (BN / 2) + nf * 16 + g * 4;
        if (m < a.M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.N) atomicAdd((float*)a.C + m * a.ldc + n + r, acc[nf][mf][r] * a.alpha);
        }
      }
    return;
  }
  if (a.splitk > 1) {  // raw partial sums -> this split's fp32 slab (plain stores, deterministic)
    float* slab = a.ws + (int64_t)blockIdx.y * a.M * a.N;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = m0 + wm * (BM / 2) + mf * 16 + j;
        const int n = n0 + wn * (BN / 2) + nf * 16 + g * 4;
        if (m < a.M && n < a.N) {
          float* d = slab + m * a.N + n;
          if (n + 3 < a.N && (a.N & 3) == 0) {
            *(float4*)d = make_float4(acc[nf][mf][0], acc[nf][mf][1], acc[nf][mf][2], acc[nf][mf][3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < a.N) d[r] = acc[nf][mf][r];
          }
        }
      }
    return;
  }
  if (a.act == ACT_GEGLU) {
    const int Nout = a.N >> 1;
#pragma unroll
    for (int q = 0; q < NF / 2; ++q)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = m0 + wm * (BM / 2) + mf * 16 + j;
        const int n = n0 + wn * (BN / 2) + q * 32 + g * 4;  // value cols n.., gate cols n+16..
        if (m < a.M && n < a.N) {
          float val[4], gate[4], o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            val[r] = acc[2 * q][mf][r];
            gate[r] = acc[2 * q + 1][mf][r];
          }
          epi_terms(a, m, n, val);
          epi_terms(a, m, n + 16, gate);
          if (a.preact) {
            save_preact(a, m, n, val);
            save_preact(a, m, n + 16, gate);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = val[r] * gelu_f(gate[r]);
          epi_store(a, ACT_NONE, m, ((n0 + wn * (BN / 2)) >> 1) + q * 16 + g * 4, Nout, o);
        }
      }
    return;
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t m = m0 + wm * (BM / 2) + mf * 16 + j;
      const int n = n0 + wn * (BN / 2) + nf * 16 + g * 4;
      if (m < a.M && n < a.N) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nf][mf][r];
        epi_terms(a, m, n, v);
        epi_store(a, a.act, m, n, a.N, v);
      }
    }
}

// split-K finalize: ws[M][N] f32 raw sums -> epilogue -> C
__global__ __launch_bounds__(256) void gemm_finalize_kernel(const GemmArgs a) {
  const int ngrp = (a.N + 3) >> 2;
  const int64_t total = (int64_t)a.M * ngrp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / ngrp;
    const int n = (int)(i - m * ngrp) * 4;
    if (a.act == ACT_GEGLU) {
      // process (value, gate) 16-blocks: n indexes within a 32-wide pair block
      const int blk = n >> 5, off = n & 31;
      if (off >= 16) continue;
      float val[4], gate[4], o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        val[r] = gate[r] = 0.f;
        for (int s = 0; s < a.splitk; ++s) {
          const float* slab = a.ws + (int64_t)s * a.M * a.N + m * a.N;
          if (n + r < a.N) val[r] += slab[n + r];
          if (n + 16 + r < a.N) gate[r] += slab[n + 16 + r];
        }
      }
      epi_terms(a, m, n, val);
      epi_terms(a, m, n + 16, gate);
      if (a.preact) {
        save_preact(a, m, n, val);
        save_preact(a, m, n + 16, gate);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = val[r] * gelu_f(gate[r]);
      epi_store(a, ACT_NONE, m, blk * 16 + off, a.N >> 1, o);
    } else {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int64_t sstride = (int64_t)a.M * a.N;
      const float* slab = a.ws + m * a.N + n;
      if (n + 3 < a.N && (a.N & 3) == 0) {
        // the slabs four at a time: the loads of a batch are independent and in flight together, the sums stay in slab order
        // (round 5: with one load per loop trip the kernel waited out a memory latency per slab -- 15.7 us per launch for data
        // that mostly sits in the Infinity Cache, 2.3 TB/s over 357 launches of the C2 step)
        int s = 0;
        for (; s + 4 <= a.splitk; s += 4) {
          const float4 t0 = *(const float4*)(slab + (s + 0) * sstride), t1 = *(const float4*)(slab + (s + 1) * sstride);
          const float4 t2 = *(const float4*)(slab + (s + 2) * sstride), t3 = *(const float4*)(slab + (s + 3) * sstride);
          v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
          v[0] += t1.x; v[1] += t1.y; v[2] += t1.z; v[3] += t1.w;
          v[0] += t2.x; v[1] += t2.y; v[2] += t2.z; v[3] += t2.w;
          v[0] += t3.x; v[1] += t3.y; v[2] += t3.z; v[3] += t3.w;
        }
        if (s + 2 <= a.splitk) {
          const float4 t0 = *(const float4*)(slab + (s + 0) * sstride), t1 = *(const float4*)(slab + (s + 1) * sstride);
          v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
          v[0] += t1.x; v[1] += t1.y; v[2] += t1.z; v[3] += t1.w;
          s += 2;
        }
        if (s < a.splitk) {
          const float4 t0 = *(const float4*)(slab + s * sstride);
          v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
        }
      } else {
        for (int s = 0; s < a.splitk; ++s) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.N) v[r] += slab[s * sstride + r];
        }
      }
      epi_terms(a, m, n, v);
      epi_store(a, a.act, m, n, a.N, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int MODE, bool GLDS>
static int launch_t(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int smem = 2 * (BM + BN) * 128;
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, MODE, GLDS>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  dim3 grid(tiles, a.splitk > 1 ? a.splitk : 1, 1);
  const bool prof = fdmi_prof_on();
  if (prof) gemm_prof_shape(a);
  if (prof) fdmi_prof_begin(stream, PROF_GEMM0 + MODE * 4 + (BM == 128 ? 0 : 2) + (BN == 128 ? 0 : 1), gemm_flops(a), gemm_bytes(a));
  FDMI_KLAUNCH(prof, (gemm_kernel<BM, BN, MODE, GLDS>), grid, dim3(256), smem, stream, a);
  if (prof) fdmi_prof_end(stream);
  FDMI_HIP(hipGetLastError());
  return 0;
}

template <int MODE, bool GLDS>
static int launch_tile(const GemmArgs& a, int BM, int BN, hipStream_t stream) {
  if (BM == 128 && BN == 128) return launch_t<128, 128, MODE, GLDS>(a, stream);
  if (BM == 128 && BN == 64) return launch_t<128, 64, MODE, GLDS>(a, stream);
  if (BM == 64 && BN == 128) return launch_t<64, 128, MODE, GLDS>(a, stream);
  if (BM == 64 && BN == 64) return launch_t<64, 64, MODE, GLDS>(a, stream);
  FDMI_CHECK(false, "gemm: unsupported tile");
}

// ---- kernel / tile / split-K selection by a small cost model -------------------------------------
// block rates calibrated on MI355X (scripts/kbench.py): the 256-row ring kernel sustains ~3.4 TFLOP/s
// per CU (1 block/CU), the 128x128 kernel ~1.05 per block (2/CU), 128x64 ~0.6 (3/CU), 64-row ~0.45.
static double est_us(double M, double N, double K, int BM, int BN, int slots, double rate_tf, int sk, bool atomic) {
  const double tiles = (double)cdiv((int64_t)M, BM) * cdiv((int64_t)N, BN);
  const double rounds = ceil(tiles * sk / slots);
  const double kt = ceil(K / 64.0 / sk);
  double t = rounds * (2.0 * BM * BN * 64.0 * kt / (rate_tf * 1e6) + 2.0 + 0.3 * 3);  // us; +fill/drain
  if (sk > 1 && !atomic) t += sk * M * N * 8.0 / 3.0e6 + 3.0;  // slab write + read at ~3 TB/s, + finalize launch
  return t;
}

GemmPlan plan_gemm(const GemmArgs& a, bool ws_available) {
  GemmPlan p;
  const int ktiles = cdiv(a.K, 64);
  if (a.force_tile) {
    p.BM = a.force_tile >> 16;
    p.BN = a.force_tile & 0xffff;
    p.big = p.BM == 256 ? ((p.BN == 320 || p.BN == 192 || p.BN == 384) ? 2 : 1) : ((p.BM == 128 && p.BN == 320) ? 3 : 0);
    p.splitk = a.splitk > 0 ? a.splitk : 1;
    return p;
  }
  const bool can_split = ws_available && a.splitk <= 0;
  const int forced_sk = a.splitk > 0 ? a.splitk : 0;
  double best = 1e30;
  auto consider = [&](int big, int BM, int BN, int slots, double rate) {
    for (int sk = 1; sk <= 16; ++sk) {
      if (forced_sk && sk != forced_sk) continue;
      if (!forced_sk && sk > 1 && (!can_split || ktiles / sk < 6)) break;
      const double t = est_us(a.M, a.N, a.K, BM, BN, slots, rate, sk, a.accum_atomic != 0);
      if (t < best * 0.97) {
        best = t;
        p.big = big; p.BM = BM; p.BN = BN; p.splitk = sk;
      }
    }
  };
  if (gemm3_eligible(a)) {
    consider(1, 256, gemm3_pick_bn(a), 256, 3.4);
    // the other column width may quantise better (e.g. M=4096, N=1280: 128 tiles of 256x160 vs 160 of 256x128)
    if (a.act != ACT_GEGLU && gemm3_pick_bn(a) == 160 && (a.N % 128) == 0) consider(1, 256, 128, 256, 3.4);
  }
  if (gemm4_eligible(a)) consider(2, 256, 320, 256, 4.8);
  // 256 x 192 (the widths of the transformer denoisers, which 320 does not divide): measured -6.6 % on the C4 step against the
  // 256 x 128 ring kernel it displaces (profiles/r2_knob12_c4.txt); A/B switch 12 = 1 takes it out of the planner
  if (!fdmi_tune_get(12) && gemm4_eligible(a, 192)) consider(2, 256, 192, 256, 3.9);
  // 256 x 384 (round 6): the wide tile for the same widths -- +9 ... +16 % per K loop over 256 x 192 in the micro-benchmark
  // (profiles/r6_kloop_dit_widths.txt); A/B switch 51 = 1 takes it out of the planner
  if (!fdmi_tune_get(12) && !fdmi_tune_get(51) && gemm4_eligible(a, 384)) consider(2, 256, 384, 256, 4.4);
  const bool geglu = a.act == ACT_GEGLU;
  if (a.A2) return p;   // a two-segment A operand: only the LDS-DMA kernels above read it (gemm_a2_ok)
  consider(0, 128, 128, 512, 1.05);
  consider(0, 128, 64, 768, 0.60);
  consider(0, 64, 128, 768, 0.50);
  if (a.N <= 64 || a.M <= 64) consider(0, 64, 64, 1280, 0.22);
  (void)geglu;
  return p;
}

bool gemm_a2_ok(const GemmArgs& a) {
  if (a.f32 || a.mode != GEMM_ROW || !a.A2) return false;
  if (a.K1 <= 0 || a.K1 >= a.K || (a.K1 & 63) || (a.K & 63) || (a.lda2 & 7) || ((uintptr_t)a.A2 & 15)) return false;
  return gemm3_eligible(a);   // (whenever the 256 x 320 / 256 x 192 kernel is eligible, so is the 256-row ring kernel)
}

bool gemm_r32_ok(const GemmArgs& a) {
  if (a.f32 || a.mode != GEMM_ROW || a.act != ACT_NONE || a.out_f32 || a.accum_atomic || a.preact || a.gn_stats) return false;
  if ((a.N & 7) || (a.ldc & 7) || (a.residual && (a.ldr & 7)) || (a.rowvec && (a.rowvec_ld & 7))) return false;
  if (a.residual32 && ((a.ldr32 & 7) || ((uintptr_t)a.residual32 & 15))) return false;
  if (a.C32 && ((a.ldc32 & 7) || ((uintptr_t)a.C32 & 15))) return false;
  return true;
}

bool gemm_gn_ok(const GemmArgs& a, bool ws_available) {
  if (a.f32) return false;
  if (!a.gn_stats || a.gn_cpg < 8 || a.gn_G < 1 || a.gn_G > 32 || a.N != a.gn_cpg * a.gn_G) return false;
  if (a.gn_rows <= 0 || (a.gn_rows & 255) != 0 || (a.M & 255) != 0 || (a.M % a.gn_rows) != 0) return false;
  if (a.out_f32 || a.accum_atomic || a.act == ACT_GEGLU || a.preact || a.force_tile || a.splitk > 1) return false;
  if ((a.N & 7) || (a.ldc & 7) || (a.residual && (a.ldr & 7)) || (a.rowvec && (a.rowvec_ld & 7))) return false;
  const GemmPlan p = plan_gemm(a, ws_available);
  return p.big != 0 && p.splitk == 1 && (p.big != 2 || p.BN == 320) && (a.N % p.BN) == 0;
}

size_t gemm_ws_bytes(const GemmArgs& a) {
  if (a.f32) return 0;
  GemmArgs b = a;
  b.splitk = 0;
  const GemmPlan p = plan_gemm(b, true);
  return p.splitk > 1 ? (size_t)p.splitk * a.M * a.N * sizeof(float) : 0;
}

// developer aid: FDMI_GEMM_LOG=1 prints a histogram of the launched problems at exit
#include <map>
#include <mutex>
#include <tuple>
#include <cstdlib>
namespace {
struct GemmLog {
  std::map<std::tuple<int, int, int, int, int, int, int, int>, long> n;
  bool on = getenv("FDMI_GEMM_LOG") != nullptr;
  ~GemmLog() {
    if (!on) return;
    for (auto& kv : n) {
      const auto& k = kv.first;
      fprintf(stderr, "GEMMLOG mode=%d M=%d N=%d K=%d act=%d flags=%d kern=%d sk=%d count=%ld\n", std::get<0>(k), std::get<1>(k), std::get<2>(k),
              std::get<3>(k), std::get<4>(k), std::get<5>(k), std::get<6>(k), std::get<7>(k), kv.second);
    }
  }
} g_gemm_log;
}  // namespace

// ---- in-launch split-K reduction: per-stream ticket regions ---------------------------------------------------------------------
// One counter per output tile of a launch, zero between launches (the tile's reducer resets it).  Two launches that run at the same
// time must not share counters: launches of ONE stream are serialised by the stream, so every stream gets its own region of a
// static device array (first come, first served; a 17th stream falls back to the finalize kernel).  The library allocates nothing.
namespace {
constexpr int SK_STREAMS = 16, SK_TICKETS = 4096;
__device__ int g_sk_tickets[SK_STREAMS * SK_TICKETS];
std::mutex g_sk_mu;
hipStream_t g_sk_stream[SK_STREAMS];
int g_sk_nstreams = 0;
//  a __device__ symbol has one address PER DEVICE: the cache is keyed by the current device, and the persistent kernels'
// grid (one block per CU, a multiple of 8) is read from the device the launch goes to -- the bound on the tiles one block may
// collect (SkList: 12) is computed from THAT grid, not from an assumed 248 resident blocks
constexpr int SK_MAX_DEVICES = 16;
int* g_sk_base[SK_MAX_DEVICES] = {};
int g_sk_ncu[SK_MAX_DEVICES] = {};
int sk_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return (dev >= 0 && dev < SK_MAX_DEVICES) ? dev : -1;
}
int sk_persistent_grid(int64_t items) {   // grid of gemm3 / gemm4 for `items` work items on the current device
  const int dev = sk_current_device();
  if (dev < 0) return 0;
  if (!g_sk_ncu[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int n = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount & ~7) : 256;
    g_sk_ncu[dev] = n < 8 ? 8 : n;
  }
  return items < g_sk_ncu[dev] ? (int)items : g_sk_ncu[dev];
}
int* sk_ticket_region(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  const int dev = sk_current_device();
  if (dev < 0) return nullptr;
  if (!g_sk_base[dev] && hipGetSymbolAddress((void**)&g_sk_base[dev], HIP_SYMBOL(g_sk_tickets)) != hipSuccess) {
    (void)hipGetLastError();
    g_sk_base[dev] = nullptr;
    return nullptr;
  }
  for (int i = 0; i < g_sk_nstreams; ++i)
    if (g_sk_stream[i] == st) return g_sk_base[dev] + i * SK_TICKETS;   // (a stream belongs to one device)
  if (g_sk_nstreams == SK_STREAMS) return nullptr;
  g_sk_stream[g_sk_nstreams] = st;
  return g_sk_base[dev] + (g_sk_nstreams++) * SK_TICKETS;
}
}  // namespace

int launch_gemm(const GemmArgs& a_in, hipStream_t stream) {
  if (a_in.f32) return launch_gemm32(a_in, stream);   // fp32 validation mode: its own kernel family (ref32.hip)
  GemmArgs a = a_in;
  if (!a.dev) a.dev = fdmi_tune_get(40);   // developer bits (gemm.h)
  FDMI_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
  FDMI_CHECK((a.K % 8) == 0 && (a.ldw % 8) == 0, "gemm: K and ldw must be multiples of 8");
  FDMI_CHECK(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "gemm: operands must be 16-B aligned");
  if (a.mode == GEMM_ROW) {
    FDMI_CHECK((a.lda % 8) == 0, "gemm: lda must be a multiple of 8");
  } else {
    FDMI_CHECK((a.Cin % 8) == 0, "conv: Cin must be a multiple of 8");
    FDMI_CHECK(a.K == a.KH * a.KW * a.Cin, "conv: K != KH*KW*Cin");
    FDMI_CHECK(a.stride == 1 || a.stride == 2, "conv: stride must be 1 or 2");
    FDMI_CHECK(!(a.ups && a.dgrad), "conv: ups+dgrad unsupported (dgrad at the upsampled size, then pool)");
  }
  if (a.A2) FDMI_CHECK(gemm_a2_ok(a), "gemm: a second A segment needs a row GEMM with M >= 256, N >= 128, K1 % 64 == 0, K % 64 == 0");
  if (a.act == ACT_GEGLU) FDMI_CHECK((a.N % 32) == 0, "geglu: N must be a multiple of 32");
  if (a.rowvec_mul) FDMI_CHECK(a.rowvec != nullptr && a.act != ACT_GEGLU, "gemm: rowvec_mul needs a row vector and is not available with GEGLU");
  if (a.accum_atomic) FDMI_CHECK(a.out_f32, "accum_atomic needs f32 C");
  FDMI_CHECK(a.C || a.C32, "gemm: no output");
  if (a.residual32 || a.C32) FDMI_CHECK(gemm_r32_ok(a), "gemm: the fp32 residual stream (residual32 / C32) needs a plain row GEMM, ACT_NONE, bf16 C, 8-aligned N and leading dims");
  const GemmPlan p = plan_gemm(a, a.ws != nullptr || a.accum_atomic);
  if (a.A2) FDMI_CHECK(p.big != 0, "gemm: a second A segment is read by the LDS-DMA kernels only (forced tile?)");
  if (p.big == 1) FDMI_CHECK(gemm3_eligible(a) && (p.BN == 128 || p.BN == 160), "gemm: 256-row tile not applicable to this problem");
  if (p.big == 2) FDMI_CHECK(gemm4_eligible(a, p.BN), "gemm: 256x320 / 256x192 tile not applicable to this problem");
  if (p.big == 3) FDMI_CHECK(p.splitk <= 1 && gemm5_eligible(a), "gemm: 128x320 tile not applicable to this problem");
  a.splitk = p.splitk;
  if (a.accum_atomic && fdmi_det()) a.splitk = 1;   // deterministic mode: one contributor per element of the atomic accumulation
  {  // every split must own at least one K tile (slabs of empty splits would stay uninitialised)
    const int kt = cdiv(a.K, 64);
    while (a.splitk > 1 && (a.splitk - 1) * cdiv(kt, a.splitk) >= kt) --a.splitk;
  }
  if (a.splitk > 1 && !a.accum_atomic) {
    FDMI_CHECK(a.ws != nullptr, "gemm: split-K needs a workspace of splitk*M*N floats");
  }
  if (a.gn_stats) FDMI_CHECK(gemm_gn_ok(a_in, a_in.ws != nullptr || a_in.accum_atomic), "gemm: gn_stats requested for a problem whose kernel cannot accumulate them (ask gemm_gn_ok first)");
  // In-launch reduction of the split-K slabs by the 256-row kernels (gemm_tile.h::splitk_arrive): correct and deterministic, but
  // MEASURED SLOWER on the C2 step than the finalize kernel it replaces -- the tile's last block re-reads the slabs alone while
  // the finalize kernel spreads the same bytes over every CU: in-process A/B 178.4 ms (finalize kernel) vs 182.7 ms (in-launch up to
  // 4 slabs) vs 184.0 ms (up to 8), profiles/r5_knob_ab_inlaunch_splitk.txt.  OFF by default; knob 48 = n reduces up to n slabs in the
  // launch (the A/B switch the parity tests flip).
  a.sk_tickets = nullptr;
  {
    const int sk_max = fdmi_tune_get(48) > 0 ? fdmi_tune_get(48) : 0;
    const int tiles = (p.big == 1 || p.big == 2) ? cdiv(a.M, 256) * cdiv(a.N, p.BN) : 0;
    const int64_t items = (int64_t)tiles * a.splitk;     // (a block of the persistent kernels takes ceil(items / 256) of them)
    if ((p.big == 1 || p.big == 2) && p.BN != 384 && a.splitk > 1 && a.splitk <= sk_max && !a.accum_atomic && a.act != ACT_GEGLU && !(a.dev & (16 | 32 | 0x800)) &&
        !a.residual32 && !a.C32 &&
        tiles <= SK_TICKETS) {
      const int grid = sk_persistent_grid(items);
      if (grid > 0 && (items + grid - 1) / grid <= 12) a.sk_tickets = sk_ticket_region(stream);
    }
  }
  if (g_gemm_log.on)
    ++g_gemm_log.n[std::make_tuple(a.mode, a.M, a.N, a.K, a.act, (a.residual ? 1 : 0) | (a.preact ? 2 : 0) | (a.accum_atomic ? 4 : 0) | (a.out_f32 ? 8 : 0) | (a.dgrad ? 16 : 0),
                                   p.big ? p.big * 1000 + p.BN : p.BM * 1000 + p.BN, a.splitk)];
  int rc;
  if (p.big == 3)
    rc = launch_gemm5(a, stream);
  else if (p.big == 2)
    rc = launch_gemm4(a, stream, p.BN);
  else if (p.big)
    rc = launch_gemm3(a, p.BN, stream);
  else if (a.mode == GEMM_ROW)
    rc = a.use_glds ? launch_tile<GEMM_ROW, true>(a, p.BM, p.BN, stream) : launch_tile<GEMM_ROW, false>(a, p.BM, p.BN, stream);
  else
    rc = a.use_glds ? launch_tile<GEMM_CONV, true>(a, p.BM, p.BN, stream) : launch_tile<GEMM_CONV, false>(a, p.BM, p.BN, stream);
  if (rc) return rc;
  if (a.splitk > 1 && !a.accum_atomic && !a.sk_tickets) {
    const int64_t total = (int64_t)a.M * ((a.N + 3) >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_finalize_kernel, dim3(blocks), dim3(256), 0, stream, a);
    FDMI_HIP(hipGetLastError());
  }
  return 0;
}
