// Shared pieces of the large-tile LDS-DMA GEMM kernels (gemm3.hip: 256 x {128,160}; gemm4.hip: 256 x 320):
// LDS-DMA issue / counted waits, and the epilogue of one wave's MF x NF block of 16x16 accumulators
// (swapped operands: lane (g, j) holds output row j, columns g*4..g*4+3 of each fragment).
#pragma once
#include "gemm.h"
#include <type_traits>

static __device__ uint4 g_zero16b[4] = {};

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

__device__ __forceinline__ void epi_terms3(const GemmArgs& a, int64_t m, int n, float v[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] *= a.alpha;
  if (a.bias) {
    if (n + 3 < a.N) {
      const float4 b = *(const float4*)(a.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.N) v[r] += a.bias[n + r];
    }
  }
  if (a.rowvec) {
    const bf16_t* rv = a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.N) v[r] = a.rowvec_mul ? v[r] * bf2f(rv[r]) : v[r] + bf2f(rv[r]);
  }
}
__device__ __forceinline__ void epi_store3(const GemmArgs& a, int act, int64_t m, int nout, int Nout, float v[4]) {
  if (a.residual) {
    const bf16_t* rs = a.residual + m * a.ldr + nout;
    if (nout + 3 < Nout && ((a.ldr | nout) & 3) == 0) {
      const u16x4 t = *(const u16x4*)rs;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bf2f(t[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nout + r < Nout) v[r] += bf2f(rs[r]);
    }
  }
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
__device__ __forceinline__ void save_preact3(const GemmArgs& a, int64_t m, int n, const float v[4]) {
  bf16_t* p = a.preact + m * a.ldp + n;
  uint2 pk;
  pk.x = pack2bf(v[0], v[1]);
  pk.y = pack2bf(v[2], v[3]);
  *(uint2*)p = pk;
}

// ---- 8-wide epilogue: after v_permlane16_swap of a fragment pair every lane owns 8 consecutive output
// columns of one row, so residual loads and output stores are 16 B per lane (half the store
// instructions of the native 4-per-lane MFMA layout; the store tail of short-K GEMMs is issue-bound).
__device__ __forceinline__ void swap16(float& x, float& y) {  // rows 1,3 of x <-> rows 0,2 of y
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
#define SWAP16(X, Y) do { float x_ = (X), y_ = (Y); swap16(x_, y_); (X) = x_; (Y) = y_; } while (0)
__device__ __forceinline__ void epi_terms8(const GemmArgs& a, int64_t m, int n, float v[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] *= a.alpha;
  if (a.bias) {
    const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (a.rowvec) {
    const u16x8 rv = *(const u16x8*)(a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n);
    if (a.rowvec_mul) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] *= bf2f(rv[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(rv[r]);
    }
  }
}
// activation + store of 8 consecutive columns (the residual, if any, already added)
__device__ __forceinline__ void epi_finish8(const GemmArgs& a, int act, int64_t m, int nout, float v[8]);
__device__ __forceinline__ void epi_store8(const GemmArgs& a, int act, int64_t m, int nout, float v[8]) {
  if (a.residual) {
    const u16x8 t = *(const u16x8*)(a.residual + m * a.ldr + nout);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += bf2f(t[r]);
  }
  epi_finish8(a, act, m, nout, v);
}
__device__ __forceinline__ void epi_finish8(const GemmArgs& a, int act, int64_t m, int nout, float v[8]) {
  if (act == ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = silu_f(v[r]);
  } else if (act == ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
  } else if (act == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
  } else if (act == ACT_GELU_TANH) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast(v[r]);
  }
  if (a.out_f32) {
    float* c = (float*)a.C + m * a.ldc + nout;
    *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 pk;
    pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
    *(uint4*)((bf16_t*)a.C + m * a.ldc + nout) = pk;
  }
}

// ---- GroupNorm statistics of the CONSUMER in the producing GEMM's epilogue (GemmArgs::gn_stats; opt-in, see gemm_gn_ok) ----
// gn_stats[b][G][2] receives (sum, sum of squares) of the bf16-rounded values this wave stores -- what norm.hip's
// gn_reduce_kernel would read back from HBM.  A lane owns NC (8 or 4) consecutive columns n.. of one row per row fragment;
// cpg >= 8 >= NC, so its columns touch at most two groups: A = n / cpg (the first `split` columns) and B = A + 1.
template <int NC>
__device__ __forceinline__ void gn_lane_add(const float* v, int split, float (&s)[4]) {
#pragma unroll
  for (int e = 0; e < NC; ++e) {
    const float r = bf2f(f2bf(v[e]));
    if (e < split) { s[0] += r; s[1] = fmaf(r, r, s[1]); }
    else           { s[2] += r; s[3] = fmaf(r, r, s[3]); }
  }
}
// sum over the 16 lanes j of a lane group (the 16 rows of a fragment), then one lane adds the wave's partial sums
__device__ __forceinline__ void gn_flush(const GemmArgs& a, int64_t mw, int n, int nc, int j, float (&s)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s[q] += __shfl_xor(s[q], o, 64);
  }
  if (j == 0 && n < a.N) {
    const int gA = n / a.gn_cpg;
    float* d = a.gn_stats + ((mw / a.gn_rows) * a.gn_G + gA) * 2;
    atomicAdd(d, s[0]);
    atomicAdd(d + 1, s[1]);
    if ((n + nc - 1) / a.gn_cpg != gA) {
      atomicAdd(d + 2, s[2]);
      atomicAdd(d + 3, s[3]);
    }
  }
}

// LDS-DMA issued from inline asm: hipcc then keeps no scoreboard entry for it, so the ONLY waits on
// these loads are the counted ones placed by hand below (with the builtin, the waitcnt pass drained
// the ring with vmcnt(0) at every loop back-edge of the persistent loop).  M0 (the LDS destination
// base) is written and restored inside the statement; lds_addr is wave-uniform.
__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(lds_addr));
}

// counted wait on the LDS-DMA pieces (plus: every LDS read of this wave has returned, so the ring slot
// it read from may be refilled once the workgroup has passed the barrier that follows)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// ---- lean epilogue of the common case (round 4).  Measured on the 256 x 320 row kernel (scripts/rowbench.py dev, knob 40 = 32:
// the K loop alone): M = 131072, N = K = 320 with a residual runs 78 us of which 53 us are the general epilogue below -- 40
// (column pair, row fragment) steps per lane of ~250 executed instructions each (64-bit address products per element, scalar
// branches on run-time flags, v_readlane reloads of spilled scalars), i.e. the epilogue is instruction-bound at 3 TB/s, not
// memory-bound.  This variant fixes the operand set at compile time (RES: a residual tile; ADD: the bias; VEC: the wave's row
// of a per-sample vector; the terms are added in the general epilogue's order, so the results are bit-identical), hoists every pointer product out of the loops (one 64-bit pointer per lane,
// row-fragment steps by addition, column pairs as immediate offsets) and keeps the residual one column pair ahead: ~40
// instructions per 16-byte store.  Whole tiles, 8-column-aligned operands, bf16 output, no activation (epi_fast_ok).
__device__ __forceinline__ bool epi_fast_ok(const GemmArgs& a) {   // (a reduced split-K tile takes the general epilogue)
  return !a.accum_atomic && a.splitk <= 1 && !a.residual32 && !a.C32 && a.act == ACT_NONE && !a.out_f32 && !a.preact && a.alpha == 1.f &&
         (a.N & 7) == 0 && (a.ldc & 7) == 0 && (!a.residual || (a.ldr & 7) == 0) &&
         (!a.rowvec || ((a.rowvec_ld & 7) == 0 && !a.rowvec_mul && (a.rows_per_batch & 63) == 0));
}
// Line-wide accesses (round 4, scripts/ubench/stream_patterns.hip -> profiles/r4_stream_patterns.txt): a wave instruction of the
// MFMA layout touches 16 rows x 64 bytes; a copy in that pattern runs at 3.66 TB/s, in whole 128-byte lines at 5.0 TB/s.  Two
// column pairs P, P + 1 of a row fragment are therefore exchanged between lanes j and j ^ 8 of each 16-lane row (two DPP row
// rotations per register): afterwards lane (g, j) holds rows j & 7 and (j & 7) + 8 of column pair P + (j >> 3), so that a wave
// instruction covers 8 rows x 128 contiguous bytes.  `phase`: the wave tile starts in the second half of a line, its first pair
// stays single.  Residual, bias and row vector are fetched in the exchanged layout, so nothing has to be exchanged back.
__device__ __forceinline__ float dpp_ror8_hi(float keep, float src) {   // lanes j >= 8 of every 16-lane row := src of lane j - 8
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), 0x128, 0xF, 0xC, false));
}
__device__ __forceinline__ float dpp_ror8_lo(float keep, float src) {   // lanes j < 8 := src of lane j + 8
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), 0x128, 0xF, 0x3, false));
}
// s_setprio around the MFMA groups of the ring kernels' K loops (compile-time experiment switch: -DFDMI_NO_SETPRIO builds without it;
// round 6: scripts/ubench/gemm_kloop.hip measures 0 ... +2 % without it on the lockstep structure, the step A/B decides)
#ifdef FDMI_NO_SETPRIO
#define FDMI_SETPRIO(x) do { } while (0)
#else
#define FDMI_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifndef FDMI_EPI_RD
#define FDMI_EPI_RD 4   // residual chunks (16 bytes each) in flight per lane in the lean epilogue (compile-time experiment switch)
#endif
// the epilogue's walk as compile-time tables: unit k = (first pair, double?), micro-step i = one 16-byte residual chunk in
// processing order (unit, row fragment, row of a double) -- the residual ring below is addressed by micro-step
template <int NP, int MF, int PHASE>
struct EpiWalk {
  static constexpr int NU = PHASE == 2 ? NP : PHASE + (NP - PHASE) / 2 + ((NP - PHASE) & 1);
  static constexpr __host__ __device__ int unit_p(int k) { return PHASE == 2 ? k : ((PHASE && k == 0) ? 0 : PHASE + 2 * (k - PHASE)); }
  static constexpr __host__ __device__ bool unit_dbl(int k) { return PHASE != 2 && !(PHASE && k == 0) && unit_p(k) + 1 < NP; }
  static constexpr __host__ __device__ int micro_count() {
    int n = 0;
    for (int k = 0; k < NU; ++k) n += MF * (unit_dbl(k) ? 2 : 1);
    return n;
  }
  // micro-step i -> k * 64 + mf * 2 + s
  static constexpr __host__ __device__ int micro(int i) {
    int n = 0;
    for (int k = 0; k < NU; ++k) {
      const int per = unit_dbl(k) ? 2 : 1;
      if (i < n + MF * per) return k * 64 + ((i - n) / per) * 2 + ((i - n) % per);
      n += MF * per;
    }
    return -1;
  }
};
// GNS: also accumulate the consumer's GroupNorm sums (GemmArgs::gn_stats, see gn_lane_add / gn_flush above): per unit the lane's
// partial sums run over the unit's row fragments (all of one sample), then ONE cross-lane reduction over the lanes that hold the
// same columns -- 16 for a single pair, 8 for a double (lanes j < 8 and j >= 8 hold different pairs there)
template <int NF, int MF, bool RES, bool ADD, bool VEC, int PHASE, bool GNS = false>
__device__ __forceinline__ void tile_epilogue_fast(const GemmArgs& a, int mw, int nw, f32x4 (&acc)[NF][MF], int g, int j) {
  constexpr int NP = NF / 2;
  // units: a double (pairs P, P + 1: line-wide) or a single pair.  PHASE 0: doubles from pair 0; 1: pair 0 single, doubles from
  // pair 1; 2: every pair single (A/B switch)
  using Wk = EpiWalk<NP, MF, PHASE>;
  constexpr int NU = Wk::NU, NS = Wk::micro_count(), RD = FDMI_EPI_RD;   // RD: residual chunks in flight per lane
  const int h = j >> 3, jr = j & 7;
  const int col0 = nw + (g & 1) * 16 + (g >> 1) * 8;           // single pair pr: this lane's 8 columns start at col0 + 32 pr, row j
  const int colT = col0 + 32 * h;                              // double (P, P + 1): columns colT + 32 P, rows jr and jr + 8
  // ONE pointer per operand (the exchanged layout's); a single pair's row j = jr + 8 h, column col0 = colT - 32 h is an offset away
  bf16_t* const cpT = (bf16_t*)a.C + (int64_t)(mw + jr) * a.ldc + colT;
  const bf16_t* const rpT = RES ? a.residual + (int64_t)(mw + jr) * a.ldr + colT : nullptr;
  const int64_t cstep = 16 * a.ldc, rstep = RES ? 16 * a.ldr : 0;   // one row fragment down
  const int64_t c8 = 8 * a.ldc, r8s = RES ? 8 * a.ldr : 0;          // the second row of a double
  const int csg = h ? (int)c8 - 32 : 0, rsg = (RES && h) ? (int)r8s - 32 : 0;   // (elements; 8 rows of any operand fit 31 bits)
  // a 64-row wave tile lies inside one sample (rows_per_batch % 64 == 0): the per-sample vector is one row for the whole wave
  const bf16_t* const vrow = VEC ? a.rowvec + (int64_t)(mw / a.rows_per_batch) * a.rowvec_ld : nullptr;
  u16x8 rb[RD];
  auto load_micro = [&](int i) {   // residual chunk of micro-step i into its ring register
    if (!RES || i >= NS) return;
    const int c = Wk::micro(i), k = c >> 6, mf = (c >> 1) & 31, s2 = c & 1;
    if (Wk::unit_dbl(k)) rb[i % RD] = *(const u16x8*)(rpT + mf * rstep + 32 * Wk::unit_p(k) + (s2 ? r8s : 0));
    else rb[i % RD] = *(const u16x8*)(rpT + rsg + mf * rstep + 32 * Wk::unit_p(k));
  };
#pragma unroll
  for (int i = 0; i < RD; ++i) {
    rb[i] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
    load_micro(i);
  }
  float gs[4] = {0.f, 0.f, 0.f, 0.f};   // (GNS) this unit's partial sums: group of the lane's first columns / the next group
  int gsplit = 8;
  auto finish = [&](float (&v)[8], const float (&b8)[8], const float (&r8)[8], const u16x8& res, bf16_t* dst) {
    if (ADD) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += b8[r];
    }
    if (VEC) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += r8[r];
    }
    if (RES) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(res[r]);
    }
    uint4 pk;
    pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
    *(uint4*)dst = pk;
    if (GNS) gn_lane_add<8>(v, gsplit, gs);
  };
  int mi = 0;   // micro-step (a compile-time constant at every use once the loops are unrolled)
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int P = Wk::unit_p(k);
    const bool dbl = Wk::unit_dbl(k);
    const int cb = (dbl ? colT : col0) + 32 * P;                // this lane's first column in this unit
    if (GNS) {
      gs[0] = gs[1] = gs[2] = gs[3] = 0.f;
      gsplit = (cb / a.gn_cpg + 1) * a.gn_cpg - cb;
    }
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ADD) {
      const float4 b0 = *(const float4*)(a.bias + cb), b1 = *(const float4*)(a.bias + cb + 4);
      b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
    }
    if (VEC) {
      const u16x8 t = *(const u16x8*)(vrow + cb);
#pragma unroll
      for (int r = 0; r < 8; ++r) r8[r] = bf2f(t[r]);
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if (dbl) {
        float va[8], vb[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          SWAP16(acc[2 * P][mf][r], acc[2 * P + 1][mf][r]);
          SWAP16(acc[2 * P + 2][mf][r], acc[2 * P + 3][mf][r]);
          va[r] = acc[2 * P][mf][r];     va[4 + r] = acc[2 * P + 1][mf][r];      // row j, pair P
          vb[r] = acc[2 * P + 2][mf][r]; vb[4 + r] = acc[2 * P + 3][mf][r];      // row j, pair P + 1
        }
        float v0[8], v1[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          v0[r] = dpp_ror8_hi(va[r], vb[r]);   // j < 8: (row j, P);      j >= 8: (row j - 8, P + 1)
          v1[r] = dpp_ror8_lo(vb[r], va[r]);   // j < 8: (row j + 8, P);  j >= 8: (row j, P + 1)
        }
        const u16x8 res0 = rb[mi % RD];
        load_micro(mi + RD);
        finish(v0, b8, r8, res0, cpT + mf * cstep + 32 * P);
        ++mi;
        const u16x8 res1 = rb[mi % RD];
        load_micro(mi + RD);
        finish(v1, b8, r8, res1, cpT + mf * cstep + 32 * P + c8);
        ++mi;
      } else {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          SWAP16(acc[2 * P][mf][r], acc[2 * P + 1][mf][r]);
          v[r] = acc[2 * P][mf][r];
          v[4 + r] = acc[2 * P + 1][mf][r];
        }
        const u16x8 res0 = rb[mi % RD];
        load_micro(mi + RD);
        finish(v, b8, r8, res0, cpT + csg + mf * cstep + 32 * P);
        ++mi;
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the next steps' loads / address math from piling up registers (the tile is at the VGPR cap)
    }
    if (GNS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 1; o < (dbl ? 8 : 16); o <<= 1) gs[q] += __shfl_xor(gs[q], o, 64);
      }
      if ((dbl ? jr : j) == 0) {
        const int gA = cb / a.gn_cpg;
        float* d = a.gn_stats + ((mw / a.gn_rows) * a.gn_G + gA) * 2;
        atomicAdd(d, gs[0]);
        atomicAdd(d + 1, gs[1]);
        if ((cb + 7) / a.gn_cpg != gA) {
          atomicAdd(d + 2, gs[2]);
          atomicAdd(d + 3, gs[3]);
        }
      }
    }
  }
  if constexpr ((NF & 1) != 0) {   // BN = 160: the odd fragment, 4 columns per lane
    const int cl = nw + (NF - 1) * 16 + g * 4;
    bf16_t* const cq = (bf16_t*)a.C + (int64_t)(mw + j) * a.ldc + cl;
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
    int gsplit4 = 4;
    if (GNS) {
      gs[0] = gs[1] = gs[2] = gs[3] = 0.f;
      gsplit4 = (cl / a.gn_cpg + 1) * a.gn_cpg - cl;
    }
    if (ADD) {
      const float4 b = *(const float4*)(a.bias + cl);
      b4[0] = b.x; b4[1] = b.y; b4[2] = b.z; b4[3] = b.w;
    }
    if (VEC) {
      const u16x4 t = *(const u16x4*)(vrow + cl);
#pragma unroll
      for (int r = 0; r < 4; ++r) r4[r] = bf2f(t[r]);
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[NF - 1][mf][r];
        if (ADD) v[r] += b4[r];
        if (VEC) v[r] += r4[r];
      }
      if (RES) {
        const u16x4 t = *(const u16x4*)(a.residual + (int64_t)(mw + mf * 16 + j) * a.ldr + cl);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f(t[r]);
      }
      uint2 pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      *(uint2*)(cq + mf * cstep) = pk;
      if (GNS) gn_lane_add<4>(v, gsplit4, gs);
    }
    if (GNS) gn_flush(a, mw, cl, 4, j, gs);
  }
}

// GEGLU twin of the lean epilogue: out[m][n/2] = value * gelu(gate) of the 16-wide (value | gate) interleaved columns, bias
// present, no row vector / residual, bf16 output (+ the pre-activation save of a taped forward: PRE).  NF % 2 == 0.
__device__ __forceinline__ bool epi_fast_geglu_ok(const GemmArgs& a) {
  return a.act == ACT_GEGLU && !a.accum_atomic && a.splitk <= 1 && !a.out_f32 && a.alpha == 1.f && a.bias && !a.rowvec && !a.residual &&
         (a.N & 7) == 0 && (a.ldc & 7) == 0 && (!a.preact || (a.ldp & 7) == 0);
}
template <int NF, int MF, bool PRE>
__device__ __forceinline__ void tile_epilogue_fast_geglu(const GemmArgs& a, int mw, int nw, f32x4 (&acc)[NF][MF], int g, int j) {
  constexpr int NQUAD = NF / 4;
  // quad t = fragments 4t .. 4t+3 = (value, gate, value, gate): after the swaps even lane groups hold pair 2t, odd ones 2t + 1
  const int q = g & 1, h8 = (g >> 1) * 8;
  const int ncol = nw + q * 32 + h8;                          // value columns ncol .. +7 of quad 0, gate columns + 16 (quad t: + 64 t)
  const int ocol = (nw >> 1) + q * 16 + h8;                   // output columns of quad 0 (quad t: + 32 t)
  bf16_t* const cp = (bf16_t*)a.C + (int64_t)(mw + j) * a.ldc + ocol;
  bf16_t* const pp = PRE ? a.preact + (int64_t)(mw + j) * a.ldp + ncol : nullptr;
  const int64_t cstep = 16 * a.ldc, pstep = PRE ? 16 * a.ldp : 0;
  const float* const bp = a.bias + ncol;
#pragma unroll
  for (int t = 0; t < NQUAD; ++t) {
    const float4 v0 = *(const float4*)(bp + 64 * t), v1 = *(const float4*)(bp + 64 * t + 4);
    const float4 g0 = *(const float4*)(bp + 64 * t + 16), g1 = *(const float4*)(bp + 64 * t + 20);
    const float bv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w}, bg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        SWAP16(acc[4 * t][mf][r], acc[4 * t + 2][mf][r]);
        SWAP16(acc[4 * t + 1][mf][r], acc[4 * t + 3][mf][r]);
      }
      float val[8], gate[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        val[r] = acc[4 * t][mf][r] + bv[r];       val[4 + r] = acc[4 * t + 2][mf][r] + bv[4 + r];
        gate[r] = acc[4 * t + 1][mf][r] + bg[r];  gate[4 + r] = acc[4 * t + 3][mf][r] + bg[4 + r];
      }
      if (PRE) {
        uint4 pk;
        pk.x = pack2bf(val[0], val[1]); pk.y = pack2bf(val[2], val[3]); pk.z = pack2bf(val[4], val[5]); pk.w = pack2bf(val[6], val[7]);
        *(uint4*)(pp + mf * pstep + 64 * t) = pk;
        pk.x = pack2bf(gate[0], gate[1]); pk.y = pack2bf(gate[2], gate[3]); pk.z = pack2bf(gate[4], gate[5]); pk.w = pack2bf(gate[6], gate[7]);
        *(uint4*)(pp + mf * pstep + 64 * t + 16) = pk;
      }
      uint4 o;   // (two gates per gelu_f2: packed fp32 arithmetic, bit-identical to gelu_f per component)
      const fdmi_f2 o0 = (fdmi_f2){val[0], val[1]} * gelu_f2((fdmi_f2){gate[0], gate[1]});
      const fdmi_f2 o1 = (fdmi_f2){val[2], val[3]} * gelu_f2((fdmi_f2){gate[2], gate[3]});
      const fdmi_f2 o2 = (fdmi_f2){val[4], val[5]} * gelu_f2((fdmi_f2){gate[4], gate[5]});
      const fdmi_f2 o3 = (fdmi_f2){val[6], val[7]} * gelu_f2((fdmi_f2){gate[6], gate[7]});
      o.x = pack2bf(o0.x, o0.y);
      o.y = pack2bf(o1.x, o1.y);
      o.z = pack2bf(o2.x, o2.y);
      o.w = pack2bf(o3.x, o3.y);
      *(uint4*)(cp + mf * cstep + 32 * t) = o;
    }
  }
  if constexpr ((NF & 3) != 0) {   // the odd (value, gate) pair of the wave tile (NF = 10): 4 columns per lane
    constexpr int qq = NF / 2 - 1;
    const int n = nw + qq * 32 + g * 4, no = (nw >> 1) + qq * 16 + g * 4;
    const float4 bv = *(const float4*)(a.bias + n), bg = *(const float4*)(a.bias + n + 16);
    bf16_t* const cq = (bf16_t*)a.C + (int64_t)(mw + j) * a.ldc + no;
    bf16_t* const pq = PRE ? a.preact + (int64_t)(mw + j) * a.ldp + n : nullptr;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const float val[4] = {acc[2 * qq][mf][0] + bv.x, acc[2 * qq][mf][1] + bv.y, acc[2 * qq][mf][2] + bv.z, acc[2 * qq][mf][3] + bv.w};
      const float gate[4] = {acc[2 * qq + 1][mf][0] + bg.x, acc[2 * qq + 1][mf][1] + bg.y, acc[2 * qq + 1][mf][2] + bg.z,
                             acc[2 * qq + 1][mf][3] + bg.w};
      if (PRE) {
        uint2 pk;
        pk.x = pack2bf(val[0], val[1]); pk.y = pack2bf(val[2], val[3]);
        *(uint2*)(pq + mf * pstep) = pk;
        pk.x = pack2bf(gate[0], gate[1]); pk.y = pack2bf(gate[2], gate[3]);
        *(uint2*)(pq + mf * pstep + 16) = pk;
      }
      uint2 o;
      const fdmi_f2 o0 = (fdmi_f2){val[0], val[1]} * gelu_f2((fdmi_f2){gate[0], gate[1]});
      const fdmi_f2 o1 = (fdmi_f2){val[2], val[3]} * gelu_f2((fdmi_f2){gate[2], gate[3]});
      o.x = pack2bf(o0.x, o0.y);
      o.y = pack2bf(o1.x, o1.y);
      *(uint2*)(cq + mf * cstep) = o;
    }
  }
}


// ---- in-launch split-K reduction (round 5; cdna_hip_programming.md, "In-launch split-K reduction") --------------------------------
// Every block of a tile's `splitk` items writes its fp32 slab (tile_epilogue's split path), then the workgroup hands off ONCE:
// every wave waits for its slab stores, the workgroup meets, thread 0 releases at agent scope (L2 write-back: the slabs of a tile
// may come from different XCDs) and draws a ticket; the block that draws splitk - 1 is the tile's reducer: thread 0 acquires at
// agent scope, resets the ticket for the next launch and tells the workgroup through 4 bytes of LDS behind the ring.  The reducer
// then sums the slabs in slab order -- its own included, read back from memory, so the result does not depend on which block
// came last (bit-identical to gemm_finalize_kernel's sum) -- into its accumulators and runs the ordinary epilogue (FINAL).
// Nobody ever WAITS for another block (no spinning: two such kernels on two streams cannot starve each other of CUs).
// Order matters on ROCm 7.2: fence FIRST, then the ticket; the asm waits restate what the compiler may drop around the fence.
// The acquire side runs once, before the block's reductions (after its last item): every slab it will read was released before
// the ticket it drew last.
// The reduction itself runs AFTER the kernel's persistent loop (SkList below): a reducer inlined into the loop lengthened live
// ranges across the K loop -- a spill reload there drains the whole LDS-DMA ring once per K tile (scripts/kloop_spill_audit.py) --
// and an out-of-line reducer made hipcc keep the accumulators in scratch.  In the loop only this hand-off happens: thread 0 appends
// the tile to a short list in LDS (behind the ring) when its block drew the last ticket.
struct SkList { int n; int pad[3]; int m0[12]; int n0[12]; };   // 112 bytes; launch_gemm bounds the items per block by 12
constexpr int SK_LDS_BYTES = 128;
__device__ __forceinline__ void splitk_arrive(const GemmArgs& a, int tile, int m0, int n0, char* lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int old = __hip_atomic_fetch_add(a.sk_tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == a.splitk - 1) {
      __hip_atomic_store(a.sk_tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch on this stream
      volatile SkList* l = (volatile SkList*)lds;
      const int k = l->n;
      if (k >= 12) __builtin_trap();   // (launch_gemm bounds the items per block by the list's size from the actual grid: unreachable)
      l->m0[k] = m0;
      l->n0[k] = n0;
      l->n = k + 1;
    }
  }
}
// the reducer's sums: acc[nf][mf] := slab_0 + slab_1 + ... (this wave's fragment positions; rows mw + mf*16 + j, columns nw + nf*16 + g*4)
template <int NF, int MF, bool FULL>
__device__ __forceinline__ void splitk_sum_slabs(const GemmArgs& a, int mw, int nw, f32x4 (&acc)[NF][MF], int g, int j) {
  const int64_t sstride = (int64_t)a.M * a.N;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t m = mw + mf * 16 + j;
      const int n = nw + nf * 16 + g * 4;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (FULL || (m < a.M && n < a.N)) {
        const float* p = a.ws + m * a.N + n;
        if (FULL || (n + 3 < a.N && (a.N & 3) == 0)) {
          int s = 0;
          for (; s + 2 <= a.splitk; s += 2) {   // two slabs in flight, summed in slab order
            const float4 t0 = *(const float4*)(p + s * sstride), t1 = *(const float4*)(p + (s + 1) * sstride);
            v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
            v[0] += t1.x; v[1] += t1.y; v[2] += t1.z; v[3] += t1.w;
          }
          if (s < a.splitk) {
            const float4 t0 = *(const float4*)(p + s * sstride);
            v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
          }
        } else {
          for (int s = 0; s < a.splitk; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < a.N) v[r] += p[s * sstride + r];
        }
      }
      acc[nf][mf] = v;
    }
}

// ---- epilogue of the fp32-residual-stream GEMMs (GemmArgs::residual32 / C32; the R32 instantiations of gemm3 / gemm4) --------------
// v = (alpha acc + bias) (* | +) rowvec + residual (bf16) + residual32 (fp32); C32 <- v (fp32), C <- bf16(v).  8 columns per lane
// after the fragment-pair swap (16-byte bf16 stores, 2 x 16-byte fp32 loads / stores), bounds-checked; split-K slabs as usual.
// ACTRT (the 256 x 384 tile, whose ONLY epilogue this is -- the general one's live ranges do not fit beside its 192 accumulators): a run-time
// tanh-GELU (GemmArgs::act == ACT_GELU_TANH, uniform) after every additive term, where the general epilogue applies it
template <int NF, int MF, bool ACTRT = false>
__device__ __forceinline__ void tile_epilogue_r32(const GemmArgs& a, int mw, int nw, int z, f32x4 (&acc)[NF][MF], int g, int j) {
  if (a.splitk > 1) {
    float* slab = a.ws + (int64_t)z * a.M * a.N;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
        const int n = nw + nf * 16 + g * 4;
        if (m < a.M && n < a.N) *(float4*)(slab + m * a.N + n) = make_float4(acc[nf][mf][0], acc[nf][mf][1], acc[nf][mf][2], acc[nf][mf][3]);
      }
    return;
  }
  // column pair outer, row fragment inner (bias / per-sample row vector of a column group fetched once); every operand access is a
  // whole 16-byte (bf16) or 2 x 16-byte (fp32) vector -- gemm_r32_ok guarantees the alignments
  const bool rvec = a.rowvec != nullptr, rmul = a.rowvec_mul != 0;
  const bf16_t* const vrow = rvec ? a.rowvec + (int64_t)(mw / a.rows_per_batch) * a.rowvec_ld : nullptr;   // (a 64-row wave tile lies in one sample when rows_per_batch % 64 == 0; else per row below)
  const bool vuni = rvec && (a.rows_per_batch & 63) == 0;
  auto finish8 = [&](int64_t m, int n, float (&v)[8], const float (&b8)[8], const float (&r8)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = fmaf(v[r], a.alpha, b8[r]);
    if (rvec) {
      float q8[8];
      if (vuni) {
#pragma unroll
        for (int r = 0; r < 8; ++r) q8[r] = r8[r];
      } else {
        const u16x8 t = *(const u16x8*)(a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n);
#pragma unroll
        for (int r = 0; r < 8; ++r) q8[r] = bf2f(t[r]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = rmul ? v[r] * q8[r] : v[r] + q8[r];
    }
    if (a.residual) {
      const u16x8 t = *(const u16x8*)(a.residual + m * a.ldr + n);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(t[r]);
    }
    if (a.residual32) {
      const float4 t0 = *(const float4*)(a.residual32 + m * a.ldr32 + n), t1 = *(const float4*)(a.residual32 + m * a.ldr32 + n + 4);
      v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
    }
    if constexpr (ACTRT) {
      if (a.act == ACT_GELU_TANH) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast(v[r]);
      }
    }
    if (a.C32) {
      float* c = a.C32 + m * a.ldc32 + n;
      *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (a.C) {   // (nullptr: nobody reads the bf16 shadow of this fp32-stream output -- its consumer is a LayerNorm, which reads C32)
      uint4 pk;
      pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
      *(uint4*)((bf16_t*)a.C + m * a.ldc + n) = pk;
    }
  };
  auto finish = [&](int64_t m, int n, float* v, int nc) {   // the odd fragment: 4 consecutive columns n .. of row m
    for (int r = 0; r < nc; ++r) v[r] *= a.alpha;
    if (a.bias)
      for (int r = 0; r < nc; ++r) v[r] += a.bias[n + r];
    if (a.rowvec) {
      const bf16_t* rv = a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n;
      for (int r = 0; r < nc; ++r) v[r] = a.rowvec_mul ? v[r] * bf2f(rv[r]) : v[r] + bf2f(rv[r]);
    }
    if (a.residual) {
      const bf16_t* rs = a.residual + m * a.ldr + n;
      for (int r = 0; r < nc; ++r) v[r] += bf2f(rs[r]);
    }
    if (a.residual32) {
      const float* rs = a.residual32 + m * a.ldr32 + n;
      for (int r = 0; r < nc; r += 4) {
        const float4 t = *(const float4*)(rs + r);
        v[r] += t.x; v[r + 1] += t.y; v[r + 2] += t.z; v[r + 3] += t.w;
      }
    }
    if (a.C32) {
      float* c = a.C32 + m * a.ldc32 + n;
      for (int r = 0; r < nc; r += 4) *(float4*)(c + r) = make_float4(v[r], v[r + 1], v[r + 2], v[r + 3]);
    }
    if (!a.C) return;
    bf16_t* c = (bf16_t*)a.C + m * a.ldc + n;
    for (int r = 0; r < nc; r += 4) {
      uint2 pk;
      pk.x = pack2bf(v[r], v[r + 1]);
      pk.y = pack2bf(v[r + 2], v[r + 3]);
      *(uint2*)(c + r) = pk;
    }
  };
#pragma unroll
  for (int pr = 0; pr < NF / 2; ++pr) {
    const int nf = 2 * pr;
    const int n = nw + (nf + (g & 1)) * 16 + (g >> 1) * 8;
    const bool nok = n < a.N;
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.bias && nok) {
      const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
      b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
    }
    if (vuni && nok) {
      const u16x8 t = *(const u16x8*)(vrow + n);
#pragma unroll
      for (int r = 0; r < 8; ++r) r8[r] = bf2f(t[r]);
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t m = mw + mf * 16 + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) SWAP16(acc[nf][mf][r], acc[nf + 1][mf][r]);
      if (m < a.M && nok) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[nf][mf][r];
          v[4 + r] = acc[nf + 1][mf][r];
        }
        finish8(m, n, v, b8, r8);
      }
    }
  }
  if constexpr ((NF & 1) != 0) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t m = mw + mf * 16 + j;
      const int n = nw + (NF - 1) * 16 + g * 4;
      if (m < a.M && n < a.N) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[NF - 1][mf][r];
        finish(m, n, v, 4);
      }
    }
  }
}

// epilogue of one wave: rows mw + mf*16 + j, columns nw + nf*16 + g*4 .. +3; z = split-K slab index
// EPI selects the compiled paths: 0 = everything but GEGLU, 1 = GEGLU only, 2 = all
// GN: also accumulate the consumer's GroupNorm statistics (full tiles, `wide` layout, no split-K: gemm_gn_ok)
// FULL: the caller guarantees whole tiles (M % 256 == 0, N % BN == 0: gemm4's eligibility) -- no per-lane bounds checks
// PF: fetch the residual one column pair ahead of its use (16 more live registers: the row-GEMM kernels have them in their
// epilogue, the conv kernels -- whose gather state stays live across items -- do not, and their long K loops need it least)
// FINAL: `acc` holds the sums over ALL split-K slabs (in-launch reduction): run the real epilogue although a.splitk > 1
template <int NF, int MF, int EPI = 2, bool GN = false, bool FULL = false, bool PF = false, bool FINAL = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, int mw, int nw, int z, f32x4 (&acc)[NF][MF], int g, int j) {
  if (EPI != 1 && a.accum_atomic) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
        const int n = nw + nf * 16 + g * 4;
        if (m < a.M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.N) atomicAdd((float*)a.C + m * a.ldc + n + r, acc[nf][mf][r] * a.alpha);
        }
      }
    return;
  }
  // wide (8 columns per lane) path needs 8-element alignment of every row pointer involved
  const bool wide = (a.N & 7) == 0 && (a.ldc & 7) == 0 && (!a.residual || (a.ldr & 7) == 0) &&
                    (!a.rowvec || (a.rowvec_ld & 7) == 0) && (!a.preact || (a.ldp & 7) == 0);
  if (EPI != 1 && a.splitk > 1 && !FINAL) {  // raw partial sums -> this split's fp32 slab (plain stores, deterministic)
    float* slab = a.ws + (int64_t)z * a.M * a.N;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
        const int n = nw + nf * 16 + g * 4;
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
  if constexpr (EPI != 1) {
    // (knob 40 = 64: the general epilogue below for every problem -- the A/B switch of the lean one)
    if (epi_fast_ok(a) && GN == (a.gn_stats != nullptr) && !(a.dev & 64) && (FULL || (mw + MF * 16 <= a.M && nw + NF * 16 <= a.N))) {
      // knob 40 = 256: every pair single (the 16-row x 64-byte pattern of the MFMA layout), the A/B switch of the line-wide form
      // (the GroupNorm-sum instantiations keep every pair single: with the doubles' extra live registers hipcc spilled an LDS
      // offset of the conv kernel's K LOOP -- and waits vmcnt(0), i.e. for the whole LDS-DMA ring, at its reload)
      const int ph = ((a.dev & 256) || GN) ? 2 : ((nw >> 5) & 1);
      const int sel = (a.residual ? 4 : 0) | (a.bias ? 2 : 0) | (a.rowvec ? 1 : 0);
#define FDMI_EPI_FAST(R_, A_, V_)                                                                 \
  do {                                                                                            \
    if (!GN && ph == 0) tile_epilogue_fast<NF, MF, R_, A_, V_, 0, false>(a, mw, nw, acc, g, j);   \
    else if (!GN && ph == 1) tile_epilogue_fast<NF, MF, R_, A_, V_, 1, false>(a, mw, nw, acc, g, j); \
    else tile_epilogue_fast<NF, MF, R_, A_, V_, 2, GN>(a, mw, nw, acc, g, j);                     \
  } while (0)
      switch (sel) {
        case 0: FDMI_EPI_FAST(false, false, false); break;
        case 1: FDMI_EPI_FAST(false, false, true); break;
        case 2: FDMI_EPI_FAST(false, true, false); break;
        case 3: FDMI_EPI_FAST(false, true, true); break;
        case 4: FDMI_EPI_FAST(true, false, false); break;
        case 5: FDMI_EPI_FAST(true, false, true); break;
        case 6: FDMI_EPI_FAST(true, true, false); break;
        default: FDMI_EPI_FAST(true, true, true); break;
      }
#undef FDMI_EPI_FAST
      return;
    }
  }
  if constexpr (EPI != 0 && (NF & 1) == 0 && !GN) {
    if (epi_fast_geglu_ok(a) && !(a.dev & 64) && (FULL || (mw + MF * 16 <= a.M && nw + NF * 16 <= a.N))) {
      if (a.preact) tile_epilogue_fast_geglu<NF, MF, true>(a, mw, nw, acc, g, j);
      else tile_epilogue_fast_geglu<NF, MF, false>(a, mw, nw, acc, g, j);
      return;
    }
  }
  if (EPI != 0 && (EPI == 1 || a.act == ACT_GEGLU)) {
    // fragment 2q holds 16 value columns, fragment 2q+1 the matching 16 gate columns
    const int Nout = a.N >> 1;
    constexpr int NQUAD = NF / 4;  // groups of two (value, gate) pairs that can take the 8-wide path
    constexpr int QW = NQUAD * 2;  // pairs covered by the 8-wide path when `wide`
    if (wide) {
#pragma unroll
      for (int t = 0; t < NQUAD; ++t) {
        // pair value fragments (4t, 4t+2) and gate fragments (4t+1, 4t+3): even lane groups get pair 2t, odd 2t+1
        const int q = g & 1;
        const int n = nw + t * 64 + q * 32 + (g >> 1) * 8;  // value cols n..n+7, gate cols n+16..n+23
        const bool nok = FULL || n < a.N;
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a.bias && nok) {   // the bias of this column group: once, not once per row fragment
          const float4 v0 = *(const float4*)(a.bias + n), v1 = *(const float4*)(a.bias + n + 4);
          const float4 g0 = *(const float4*)(a.bias + n + 16), g1 = *(const float4*)(a.bias + n + 20);
          bv[0] = v0.x; bv[1] = v0.y; bv[2] = v0.z; bv[3] = v0.w; bv[4] = v1.x; bv[5] = v1.y; bv[6] = v1.z; bv[7] = v1.w;
          bg[0] = g0.x; bg[1] = g0.y; bg[2] = g0.z; bg[3] = g0.w; bg[4] = g1.x; bg[5] = g1.y; bg[6] = g1.z; bg[7] = g1.w;
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            SWAP16(acc[4 * t][mf][r], acc[4 * t + 2][mf][r]);
            SWAP16(acc[4 * t + 1][mf][r], acc[4 * t + 3][mf][r]);
          }
          const int64_t m = mw + mf * 16 + j;
          if ((FULL || m < a.M) && nok) {
            float val[8], gate[8], o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              val[r] = acc[4 * t][mf][r]; val[4 + r] = acc[4 * t + 2][mf][r];
              gate[r] = acc[4 * t + 1][mf][r]; gate[4 + r] = acc[4 * t + 3][mf][r];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              val[r] = fmaf(val[r], a.alpha, bv[r]);
              gate[r] = fmaf(gate[r], a.alpha, bg[r]);
            }
            if (a.rowvec) {
              const bf16_t* rvp = a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n;
              const u16x8 rv0 = *(const u16x8*)rvp, rv1 = *(const u16x8*)(rvp + 16);
#pragma unroll
              for (int r = 0; r < 8; ++r) { val[r] += bf2f(rv0[r]); gate[r] += bf2f(rv1[r]); }
            }
            if (a.preact) {
              uint4 pk;
              pk.x = pack2bf(val[0], val[1]); pk.y = pack2bf(val[2], val[3]); pk.z = pack2bf(val[4], val[5]); pk.w = pack2bf(val[6], val[7]);
              *(uint4*)(a.preact + m * a.ldp + n) = pk;
              pk.x = pack2bf(gate[0], gate[1]); pk.y = pack2bf(gate[2], gate[3]); pk.z = pack2bf(gate[4], gate[5]); pk.w = pack2bf(gate[6], gate[7]);
              *(uint4*)(a.preact + m * a.ldp + n + 16) = pk;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = val[r] * gelu_f(gate[r]);
            epi_store8(a, ACT_NONE, m, (nw >> 1) + t * 32 + q * 16 + (g >> 1) * 8, o);
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the next block's address math / loads from piling up registers
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NF / 2; ++q) {
      if (wide && q < QW) continue;
      if (FULL && wide) {   // whole tiles, aligned rows: the odd (value, gate) pair of the wave tile, 4 columns per lane
        const int n = nw + q * 32 + g * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
        if (a.bias) { bv = *(const float4*)(a.bias + n); bg = *(const float4*)(a.bias + n + 16); }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t m = mw + mf * 16 + j;
          float val[4], gate[4];
          val[0] = fmaf(acc[2 * q][mf][0], a.alpha, bv.x); val[1] = fmaf(acc[2 * q][mf][1], a.alpha, bv.y);
          val[2] = fmaf(acc[2 * q][mf][2], a.alpha, bv.z); val[3] = fmaf(acc[2 * q][mf][3], a.alpha, bv.w);
          gate[0] = fmaf(acc[2 * q + 1][mf][0], a.alpha, bg.x); gate[1] = fmaf(acc[2 * q + 1][mf][1], a.alpha, bg.y);
          gate[2] = fmaf(acc[2 * q + 1][mf][2], a.alpha, bg.z); gate[3] = fmaf(acc[2 * q + 1][mf][3], a.alpha, bg.w);
          if (a.rowvec) {
            const bf16_t* rvp = a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n;
            const u16x4 r0 = *(const u16x4*)rvp, r1 = *(const u16x4*)(rvp + 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) { val[r] += bf2f(r0[r]); gate[r] += bf2f(r1[r]); }
          }
          if (a.preact) {
            save_preact3(a, m, n, val);
            save_preact3(a, m, n + 16, gate);
          }
          const int no = (nw >> 1) + q * 16 + g * 4;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = val[r] * gelu_f(gate[r]);
          if (a.residual) {   // (as epi_store8 does for the 8-wide pairs of the same tile; `wide`: ldr % 8 == 0)
            const u16x4 t = *(const u16x4*)(a.residual + m * a.ldr + no);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += bf2f(t[r]);
          }
          if (a.out_f32) {
            *(float4*)((float*)a.C + m * a.ldc + no) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
            uint2 pk;
            pk.x = pack2bf(o[0], o[1]);
            pk.y = pack2bf(o[2], o[3]);
            *(uint2*)((bf16_t*)a.C + m * a.ldc + no) = pk;
          }
        }
        continue;
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
        const int n = nw + q * 32 + g * 4;
        if (m < a.M && n < a.N) {
          float val[4], gate[4], o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            val[r] = acc[2 * q][mf][r];
            gate[r] = acc[2 * q + 1][mf][r];
          }
          epi_terms3(a, m, n, val);
          epi_terms3(a, m, n + 16, gate);
          if (a.preact) {
            save_preact3(a, m, n, val);
            save_preact3(a, m, n + 16, gate);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = val[r] * gelu_f(gate[r]);
          epi_store3(a, ACT_NONE, m, (nw >> 1) + q * 16 + g * 4, Nout, o);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
  if constexpr (EPI == 1) return;
  if constexpr (GN) {
    if (wide) {   // (launch_gemm only picks a GN instantiation when gemm_gn_ok: gn_stats set, wide operands)
      // column-pair outer, row fragment inner: the per-lane partial sums of a column chunk run over the wave's 16 * MF rows
      // (all of one sample: gn_rows % 256 == 0) before ONE cross-lane reduction per chunk
      // row kernels: the residual one column pair ahead (see the plain path below) -- where the registers allow it: with
      // the statistics' partial sums live, the 40-fragment accumulator of the 256 x 320 tile leaves no room
      const bool pf = PF && NF * MF <= 24 && a.residual != nullptr;
      u16x8 rnext[MF];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        rnext[mf] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (pf) rnext[mf] = *(const u16x8*)(a.residual + (int64_t)(mw + mf * 16 + j) * a.ldr + nw + (g & 1) * 16 + (g >> 1) * 8);
      }
#pragma unroll
      for (int pr = 0; pr < NF / 2; ++pr) {
        const int nf = 2 * pr;
        const int n = nw + (nf + (g & 1)) * 16 + (g >> 1) * 8;
        const int split = (n / a.gn_cpg + 1) * a.gn_cpg - n;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t m = mw + mf * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) SWAP16(acc[nf][mf][r], acc[nf + 1][mf][r]);
          const u16x8 rcur = rnext[mf];
          if (pf && pr + 1 < NF / 2) rnext[mf] = *(const u16x8*)(a.residual + m * a.ldr + n + 32);   // (full tiles: gemm_gn_ok)
          if (m < a.M && n < a.N) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[nf][mf][r];
              v[4 + r] = acc[nf + 1][mf][r];
            }
            epi_terms8(a, m, n, v);
            if (pf) {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += bf2f(rcur[r]);
              epi_finish8(a, a.act, m, n, v);
            } else {
              epi_store8(a, a.act, m, n, v);   // leaves the stored (pre-rounding) values in v
            }
            gn_lane_add<8>(v, split, s);
          }
        }
        gn_flush(a, mw, n, 8, j, s);
      }
      if constexpr ((NF & 1) != 0) {
        const int n = nw + (NF - 1) * 16 + g * 4;
        const int split = (n / a.gn_cpg + 1) * a.gn_cpg - n;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t m = mw + mf * 16 + j;
          if (m < a.M && n < a.N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[NF - 1][mf][r];
            epi_terms3(a, m, n, v);
            epi_store3(a, a.act, m, n, a.N, v);
            gn_lane_add<4>(v, split, s);
          }
        }
        gn_flush(a, mw, n, 4, j, s);
      }
      return;
    }
  }
  if (!GN && !PF && wide) {   // the register-lean order (conv kernels): row fragment outer, everything fetched at its use
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t m = mw + mf * 16 + j;
#pragma unroll
      for (int pr = 0; pr < NF / 2; ++pr) {
        const int nf = 2 * pr;
#pragma unroll
        for (int r = 0; r < 4; ++r) SWAP16(acc[nf][mf][r], acc[nf + 1][mf][r]);
        const int n = nw + (nf + (g & 1)) * 16 + (g >> 1) * 8;
        if (FULL || (m < a.M && n < a.N)) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[nf][mf][r];
            v[4 + r] = acc[nf + 1][mf][r];
          }
          epi_terms8(a, m, n, v);
          epi_store8(a, a.act, m, n, v);
        }
      }
      if constexpr ((NF & 1) != 0) {
        const int n = nw + (NF - 1) * 16 + g * 4;
        if (m < a.M && n < a.N) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[NF - 1][mf][r];
          epi_terms3(a, m, n, v);
          epi_store3(a, a.act, m, n, a.N, v);
        }
      }
    }
    return;
  }
  if (!GN && wide) {
    // column pair outer, row fragment inner: the bias of a column group is loaded once, and the residual tile -- the one
    // operand of this epilogue that comes from HBM -- is fetched one column pair (MF 16-byte loads per lane) AHEAD of its use,
    // so its latency hides behind the previous pair's arithmetic and stores instead of stalling every fragment
    constexpr int NP = NF / 2;
    const bool has_res = a.residual != nullptr;
    auto ncol = [&](int pr) { return nw + (2 * pr + (g & 1)) * 16 + (g >> 1) * 8; };
    u16x8 rnext[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) rnext[mf] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (has_res && NP > 0) {
      const int n0 = ncol(0);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
        if ((FULL || m < a.M) && (FULL || n0 < a.N)) rnext[mf] = *(const u16x8*)(a.residual + m * a.ldr + n0);
      }
    }
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int nf = 2 * pr;
      const int n = ncol(pr);
      const bool nok = FULL || n < a.N;
      float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (a.bias && nok) {
        const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
        b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) SWAP16(acc[nf][mf][r], acc[nf + 1][mf][r]);
        const bool ok = (FULL || m < a.M) && nok;
        const u16x8 rcur = rnext[mf];
        if (has_res && pr + 1 < NP) {   // this row's residual chunk of the NEXT column pair
          const int n1 = ncol(pr + 1);
          if ((FULL || m < a.M) && (FULL || n1 < a.N)) rnext[mf] = *(const u16x8*)(a.residual + m * a.ldr + n1);
        }
        if (ok) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = fmaf(acc[nf][mf][r], a.alpha, b8[r]);
            v[4 + r] = fmaf(acc[nf + 1][mf][r], a.alpha, b8[4 + r]);
          }
          if (a.rowvec) {
            const u16x8 rv = *(const u16x8*)(a.rowvec + (m / a.rows_per_batch) * a.rowvec_ld + n);
            if (a.rowvec_mul) {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] *= bf2f(rv[r]);
            } else {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += bf2f(rv[r]);
            }
          }
          if (has_res) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += bf2f(rcur[r]);
          }
          if (a.act == ACT_SILU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = silu_f(v[r]);
          } else if (a.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
          } else if (a.act == ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
          } else if (a.act == ACT_GELU_TANH) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast(v[r]);
          }
          if (a.out_f32) {
            float* c = (float*)a.C + m * a.ldc + n;
            *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            uint4 pk;
            pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
            *(uint4*)((bf16_t*)a.C + m * a.ldc + n) = pk;
          }
        }
      }
    }
    if constexpr ((NF & 1) != 0) {
      const int n = nw + (NF - 1) * 16 + g * 4;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = mw + mf * 16 + j;
        if (m < a.M && n < a.N) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[NF - 1][mf][r];
          epi_terms3(a, m, n, v);
          epi_store3(a, a.act, m, n, a.N, v);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t m = mw + mf * 16 + j;
      const int n = nw + nf * 16 + g * 4;
      if (m < a.M && n < a.N) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nf][mf][r];
        epi_terms3(a, m, n, v);
        epi_store3(a, a.act, m, n, a.N, v);
      }
    }
}

}  // namespace
