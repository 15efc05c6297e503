// Fused (flash-style) attention for gfx950: forward, dQ and dK/dV kernels, MFMA 16x16x32 bf16.
//
// Layout: Q/K/V/O are token-major [B, S, H*d] (the GEMM outputs, no permutes); the PV-type products
// contract over the sequence index, so their "row" operand comes from head-transposed copies
// XT[B,H,dvpad,spad] (zero padded) written by transpose_heads_kernel.
//
// The score product is computed SWAPPED (S^T = K Q^T) so that a lane owns one query column:
// lane (g = lane>>4, j = lane&15) of fragment (kf, qf) holds S^T[kv = 16kf+4g+r][q = 16qf+j].
// Row max / row sum then need only two cross-lane steps (xor 16, 32) and the probabilities feed
// the second MFMA straight from registers as its B operand (k index 8g+i <-> kv = 32s+4g+i for
// i<4, 32s+16+4g+(i-4) otherwise; the A operand applies the same permutation when it reads X^T).
// K/V tiles (64 keys) are staged through LDS with 16-B padded rows (conflict-free ds_read_b128),
// prefetched into registers one tile ahead so the HBM latency hides under the MFMAs.
#include <type_traits>

#include "ops.h"

namespace {

constexpr int KVB = 64;           // keys per tile
constexpr int TROWB = KVB * 2 + 16;  // bytes per row of a transposed [dd][64] tile

template <int DK> struct RowTile {
  static constexpr int CPRW = DK / 8;
  static constexpr int ROWB = DK * 2 + 16;
  static constexpr int BYTES = KVB * ROWB;
  static constexpr int NREG = (KVB * CPRW + 255) / 256;
};
template <int DV> struct TrTile {
  static constexpr int BYTES = DV * TROWB;
  static constexpr int NREG = (DV * 8 + 255) / 256;
};

// ---- global -> registers -> LDS tile movers --------------------------------------------------
template <int DK>
__device__ __forceinline__ void rows_g2r(uint4 (&reg)[RowTile<DK>::NREG], const bf16_t* X, int64_t ld,
                                         int b, int S, int s0, int hoff, int d, int tid) {
  constexpr int CPRW = RowTile<DK>::CPRW;
#pragma unroll
  for (int k = 0; k < RowTile<DK>::NREG; ++k) {
    const int i = tid + 256 * k;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i < KVB * CPRW) {
      const int r = i / CPRW, c = i - r * CPRW, s = s0 + r;
      if (s < S && c * 8 < d) v = *(const uint4*)(X + ((int64_t)b * S + s) * ld + hoff + c * 8);
    }
    reg[k] = v;
  }
}
template <int DK>
__device__ __forceinline__ void rows_r2s(char* lds, const uint4 (&reg)[RowTile<DK>::NREG], int tid) {
  constexpr int CPRW = RowTile<DK>::CPRW;
#pragma unroll
  for (int k = 0; k < RowTile<DK>::NREG; ++k) {
    const int i = tid + 256 * k;
    if (i < KVB * CPRW) {
      const int r = i / CPRW, c = i - r * CPRW;
      *(uint4*)(lds + r * RowTile<DK>::ROWB + c * 16) = reg[k];
    }
  }
}
template <int DV>
__device__ __forceinline__ void tr_g2r(uint4 (&reg)[TrTile<DV>::NREG], const bf16_t* XT, int b, int H,
                                       int h, int SP, int s0, int tid) {
#pragma unroll
  for (int k = 0; k < TrTile<DV>::NREG; ++k) {
    const int i = tid + 256 * k;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i < DV * 8) {
      const int r = i >> 3, c = i & 7;
      v = *(const uint4*)(XT + (((int64_t)b * H + h) * DV + r) * SP + s0 + c * 8);
    }
    reg[k] = v;
  }
}
template <int DV>
__device__ __forceinline__ void tr_r2s(char* lds, const uint4 (&reg)[TrTile<DV>::NREG], int tid) {
#pragma unroll
  for (int k = 0; k < TrTile<DV>::NREG; ++k) {
    const int i = tid + 256 * k;
    if (i < DV * 8) *(uint4*)(lds + (i >> 3) * TROWB + (i & 7) * 16) = reg[k];
  }
}

// ---- fragment readers --------------------------------------------------------------------------
template <int DK>
__device__ __forceinline__ bf16x8 row_frag(const char* lds, int f, int ks, int g, int j) {
  return *(const bf16x8*)(lds + (16 * f + j) * RowTile<DK>::ROWB + (4 * ks + g) * 16);
}
// One 8-byte LDS read that stays ONE `ds_read_b64` (round 6).  hipcc's load/store optimizer pairs plain 8-byte reads off one base
// into `ds_read2_b64` / `ds_read2st64_b64`; on gfx950 the paired form moves 128 B/clk instead of 256 and banks modulo 32 instead of
// 64 (MI355X_MICROARCH.md, LDS table), which turns the V^T fragment layouts below -- conflict-free for `ds_read_b64` -- into 2-way
// conflicts: the d = 64 forward spent 40 % of its LDS-active cycles in bank conflicts (profiles/r6_attn.txt).  A relaxed
// wavefront-scope atomic load is the same instruction with the same waitcnt bookkeeping, but not a pairing candidate.
__device__ __forceinline__ uint2 lds_read8(const char* p) {
#ifdef FDMI_PAIRED_LDS_READS   // (A/B build switch: the plain load hipcc pairs; scripts/build_variant.sh)
  return *(const uint2*)p;
#endif
  const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int df, int s2, int g, int j) {
  const char* p = lds + (16 * df + j) * TROWB + (32 * s2 + 4 * g) * 2;
  const uint2 lo = lds_read8(p);
  const uint2 hi = lds_read8(p + 32);
  union { uint4 u; bf16x8 v; } t;
  t.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return t.v;
}
// this block's own rows as an MFMA B operand: X[row0 + j][32ks + 8g .. +8]
__device__ __forceinline__ bf16x8 own_frag(const bf16_t* X, int64_t ld, int b, int S, int row, int hoff,
                                           int d, int ks, int g) {
  union { uint4 u; bf16x8 v; } t;
  t.u = make_uint4(0, 0, 0, 0);
  const int c = 32 * ks + 8 * g;
  if (row < S && c < d) t.u = *(const uint4*)(X + ((int64_t)b * S + row) * ld + hoff + c);
  return t.v;
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  union { uint4 u; bf16x8 v; } t;
  t.u = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3]));
  return t.v;
}
// reductions over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48) with the gfx950 lane-swap
// VALU ops instead of two dependent ds_bpermute round trips through the LDS
__device__ __forceinline__ float other16(float v) {  // value of lane l ^ 16
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  // after the swap r[0] = {row0, row0, row2, row2}, r[1] = {row1, row1, row3, row3}
  return (threadIdx.x & 16) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
// a bf16 fragment times a scalar (scale * log2 e folded into Q or K once per block: the MFMA output is then the
// exp2 argument, and -lse / -max rides in as the C operand)
__device__ __forceinline__ bf16x8 scaled_frag(bf16x8 v, float c) {
  union { bf16x8 v; uint32_t u[4]; } t;
  t.v = v;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    t.u[e] = pack2bf(__uint_as_float(t.u[e] << 16) * c, __uint_as_float(t.u[e] & 0xffff0000u) * c);
  return t.v;
}
constexpr float NEG_BIG = -1.0e30f;
// in-place accumulate (vdst = srcC) from inline asm: hipcc otherwise lets the P V MFMAs write a fresh register set and
// copies the whole O accumulator back at every loop latch (12 v_mov_b64 per KV tile).  The operands come from VALU
// (cvt_pk) results: the wait states the compiler would insert for a builtin are provided by the caller (s_nop).
__device__ __forceinline__ void mfma_inplace(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// Leaving the single-body tile loop (the tile before it, its last iteration, the ragged tail): hipcc may read or copy the O
// accumulators right behind the asynchronous inline-asm MFMAs -- register shuffles on the edge to differently allocated code,
// or the epilogue's first reads hoisted above a separate `s_nop` statement -- and the hardware does not interlock a VALU read of
// an in-flight MFMA result (found in round 2 at head dims 56 / 64: denominators and whole query fragments read before they
// landed).  pv_landed() is the wait (the matrix pipe is in order: once the last issued MFMA has had its wait states, all have
// written) and pv_pin() an empty statement per accumulator behind it: asm volatile statements keep their order, and a later use
// of an accumulator has to follow the statement that (as far as the compiler knows) last defined it.
__device__ __forceinline__ void pv_landed() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ void pv_landed_if(int leave) {   // the loop body's variant: tested inside the statement (no C++ branch:
  asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 15\n\ts_nop 7\n1:"   // that would split the block)
               :
               : "s"(__builtin_amdgcn_readfirstlane(leave))
               : "scc", "memory");
}
__device__ __forceinline__ void pv_pin(f32x4& acc) { asm volatile("" : "+v"(acc)); }

// =============================================================================================
// forward:  O = softmax(scale Q K^T) V ;  lse (log2 domain) optional
// block = 4 waves, each wave QF query fragments (16 rows each)
// =============================================================================================
// raw v_exp_f32 (2^x); inputs here are <= 0 or the NEG_BIG sentinel, denormal results flush harmlessly
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- LDS-DMA staging (DMA variant of the forward kernel, head dims <= 64) --------------------------------------
// K tile [64 keys][64 dims] and V^T tile [64 dims][64 keys], both as dense 128-B rows of eight 16-B chunks; the chunk at
// physical position pc of row r holds logical chunk pc ^ ((r >> 1) & 7) (the swizzle is applied on the SOURCE address:
// global_load_lds writes lane-linear), which makes the ds_read_b128 (K fragments) and ds_read_b64 (V^T fragments)
// patterns below bank-conflict free.  Chunks / rows that do not exist (dims >= d, keys >= Skv) are fetched from a zero page.
static __device__ uint4 g_attn_zero[1] = {};
#define ATTN_LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ void attn_glds16(const void* gptr, unsigned lds_addr) {  // M0 saved / restored inside
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(lds_addr));
}
__device__ __forceinline__ bf16x8 row_frag_sw(const char* lds, int f, int ks, int g, int j) {
  return *(const bf16x8*)(lds + (16 * f + j) * 128 + (((4 * ks + g) ^ ((j >> 1) & 7)) * 16));
}
__device__ __forceinline__ bf16x8 tr_frag_sw(const char* lds, int df, int s2, int g, int j) {
  const int sw = (j >> 1) & 7, c0 = 4 * s2 + (g >> 1);
  const char* row = lds + (16 * df + j) * 128 + (g & 1) * 8;
  const uint2 lo = lds_read8(row + ((c0 ^ sw) * 16));
  const uint2 hi = lds_read8(row + (((c0 + 2) ^ sw) * 16));
  union { uint4 u; bf16x8 v; } t;
  t.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return t.v;
}

// ONES: the V^T copy carries a row of ones at dd = d (spare padded row), so the PV MFMA accumulates the
// softmax denominator for free and the VALU row-sum disappears.
template <int DK, int DV, int QF, bool ONES, bool DMA>
__global__ __launch_bounds__(256, (DK <= 96 ? 2 : 1)) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!DMA || (DK == 64 && DV <= 64), "DMA staging is laid out for 64-wide tiles");
  constexpr int KBYTES = DMA ? 64 * 128 : RowTile<DK>::BYTES;
  constexpr int STAGEB = DMA ? 2 * 64 * 128 : RowTile<DK>::BYTES + TrTile<DV>::BYTES;  // K tile + V^T tile; two stages
  char* sK = smem;
  char* sV = smem + KBYTES;
  constexpr int KS = DK / 32, DF = DV / 16;
  constexpr int QP = QF >= 2 ? 2 : 1;  // query fragments processed together (bounds live registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int hoff = h * a.d;
  const int q0 = blockIdx.x * (64 * QF) + wave * (16 * QF);
  const int SP = attn_spad(a.Skv);
  const float sc = a.scale * 1.4426950408889634f;

  // Q is multiplied by scale*log2(e) once, here, so the MFMA output is already the exp2 argument
  bf16x8 qf[QF][KS];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      union { bf16x8 v; uint32_t u[4]; } t;
      t.v = own_frag(a.Q, a.ldq, b, a.Sq, q0 + 16 * f + j, hoff, a.d, ks, g);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        t.u[e] = pack2bf(__uint_as_float(t.u[e] << 16) * sc, __uint_as_float(t.u[e] & 0xffff0000u) * sc);
      qf[f][ks] = t.v;
    }

  f32x4 acc_o[DF][QF];
  float m[QF], l[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    m[f] = NEG_BIG;
    l[f] = 0.f;
#pragma unroll
    for (int df = 0; df < DF; ++df) acc_o[df][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // The zeros are materialised HERE: left alone, hipcc sinks each accumulator's `v_mov` to just in front of the first inline-asm
  // MFMA that takes it as C -- with one wait state in the d = 40 instantiation and none in the 64 x 64 ones-row one (head dims
  // 49..63: the first query fragment's denominators came out ~5 % low), and a VALU write needs two before an MFMA reads it
  // (scripts/asm_hazard_audit.py finds these statically; guide section 5.7 item 2).
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int df = 0; df < DF; ++df) pv_pin(acc_o[df][f]);
  asm volatile("s_nop 1");

  // one KV tile for query fragments [f0, f0+QP); TAIL masks keys >= Skv (last tile only)
  // DMA layout: this lane's swizzled fragment offsets inside a tile (loop invariant; fragments 16 rows = 2048 B apart)
  int kofs[KS > 0 ? KS : 1], vofs[2][2];
  {
    const int sw = (j >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kofs[ks] = j * 128 + (((4 * ks + g) ^ sw) & 7) * 16;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int c0 = 4 * s2 + (g >> 1);
      vofs[s2][0] = j * 128 + (g & 1) * 8 + ((c0 ^ sw) * 16);
      vofs[s2][1] = j * 128 + (g & 1) * 8 + (((c0 + 2) ^ sw) * 16);
    }
  }
  // Lazy running maximum.  The accumulators of K Q^T start at -m (the row's reference maximum so far), so the MFMA
  // result is directly the exp2 argument.  Three bodies of one KV tile:
  //   FIRST (tile 0): m := the tile's row maximum;
  //   FAST: m does not move, P = exp2(S - m) may exceed 1 but stays <= 2^TAU (harmless in bf16 / fp32) -- no
  //         rescale of O, no per-element subtraction; if some row exceeds m by more than TAU the tile is NOT
  //         processed (returns true) and the caller continues with
  //   SLOW: classic online softmax (m rises by max(rowmax - m, 0), O and the tile are re-based every tile).
  // The loop body is FAST with the re-basing block of SLOW kept as a cold, conditional block in front of the exp2.
  constexpr float TAU = 16.f;
  constexpr int T_FIRST = 0, T_FAST = 1, T_SLOW = 2;
  // leave (MODE == T_FAST: runtime, the loop's last iteration; every other body is followed by different code)
  auto tile_pair = [&](auto tail_tag, auto mode_tag, int f0, int t, int leave) -> bool {
    constexpr bool TAIL = decltype(tail_tag)::value;
    constexpr int MODE = decltype(mode_tag)::value;
    f32x4 s[4][QP];
    f32x4 cinit[QP];  // one read-only C operand per query fragment, shared by the four key fragments' first MFMA
#pragma unroll
    for (int f = 0; f < QP; ++f) {
      const float c0 = MODE == T_FIRST ? 0.f : -m[f0 + f];
      cinit[f] = (f32x4){c0, c0, c0, c0};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 kfr[4];
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
        kfr[kf] = DMA ? *(const bf16x8*)(sK + kofs[ks] + kf * 2048) : row_frag<DK>(sK, kf, ks, g, j);
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int f = 0; f < QP; ++f) s[kf][f] = MFMA(kfr[kf], qf[f0 + f][ks], ks == 0 ? cinit[f] : s[kf][f]);
    }
    float mx[QP];
#pragma unroll
    for (int f = 0; f < QP; ++f) {
      if (TAIL) {
        const int kvbase = t * KVB + 4 * g;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kvbase + 16 * kf + r >= a.Skv) s[kf][f][r] = NEG_BIG;
      }
      float v = fmaxf(fmaxf(s[0][f][0], s[0][f][1]), fmaxf(s[0][f][2], s[0][f][3]));
#pragma unroll
      for (int kf = 1; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaxf(v, s[kf][f][r]);
      mx[f] = xor_max(v);
    }
    bool rebase = MODE != T_FAST;
    if (MODE == T_FAST) {
      float worst = mx[0];
#pragma unroll
      for (int f = 1; f < QP; ++f) worst = fmaxf(worst, mx[f]);
      rebase = __any(worst > TAU);  // wave-uniform, rare
    }
    bf16x8 pb[QP][2];
#pragma unroll
    for (int f = 0; f < QP; ++f) {
      float alpha = 1.f;
      if (__builtin_expect(rebase, MODE != T_FAST)) {
        if (MODE == T_FAST) asm volatile("" ::: "memory");  // keep the block conditional (no speculation into the common path)
        const float up = MODE == T_FIRST ? mx[f] : fmaxf(mx[f], 0.f);  // the reference only ever rises
        if (MODE != T_FIRST) {
          alpha = fast_exp2(-up);
#pragma unroll
          for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[df][f0 + f][r] *= alpha;
          m[f0 + f] += up;
        } else {
          m[f0 + f] = up;
        }
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kf][f][r] -= up;
      }
      float ps = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(s[kf][f][r]);
          s[kf][f][r] = p;
          if (!ONES) ps += p;
        }
      if (!ONES) l[f0 + f] = l[f0 + f] * alpha + ps;
      pb[f][0] = pack8(s[0][f], s[1][f]);
      pb[f][1] = pack8(s[2][f], s[3][f]);
    }
    bf16x8 vfr[2][DF];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        if (DMA) {
          const uint2 lo = lds_read8(sV + vofs[s2][0] + df * 2048), hi = lds_read8(sV + vofs[s2][1] + df * 2048);
          union { uint4 u; bf16x8 v; } tt;
          tt.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
          vfr[s2][df] = tt.v;
        } else {
          vfr[s2][df] = tr_frag(sV, df, s2, g, j);
        }
      }
    // VALU (cvt_pk) results feed the inline-asm MFMAs: the P fragments are tied through the wait-state statement so
    // the compiler cannot schedule their producers after it (register-only VALU ops float past a plain asm)
    if constexpr (QP == 2) {
      asm volatile("s_nop 4" : "+v"(pb[0][0]), "+v"(pb[0][1]), "+v"(pb[1][0]), "+v"(pb[1][1]));
    } else {
      asm volatile("s_nop 4" : "+v"(pb[0][0]), "+v"(pb[0][1]));
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int f = 0; f < QP; ++f) mfma_inplace(acc_o[df][f0 + f], vfr[s2][df], pb[f][s2]);
    asm volatile("s_nop 3");  // the last MFMAs have read their A/B operands before compiler code may reuse those registers
    if constexpr (MODE == T_FAST) pv_landed_if(leave);
    else pv_landed();
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
      for (int f = 0; f < QP; ++f) pv_pin(acc_o[df][f0 + f]);
    return false;
  };

  const int nt = (a.Skv + KVB - 1) / KVB;
  const bool ragged = (a.Skv % KVB) != 0;
  // ---- staging.  Register path: two LDS stages, ONE barrier per KV tile: iteration t stores tile t+1 (fetched into
  // registers during iteration t-1) into the other stage, fetches tile t+2, then computes tile t.
  // DMA path: iteration t issues the LDS-DMA of tile t+1 into the other stage right after the barrier, computes tile t
  // and then waits for its own pieces (vmcnt(0)); the next barrier publishes them.  No staging registers, no ds_write,
  // per-lane source pointers advance by a constant per tile (4 pieces of 1 KiB per wave per tile). ----
  uint4 rk[DMA ? 1 : RowTile<DK>::NREG], rv[DMA ? 1 : TrTile<DV>::NREG];
  const bf16_t* dsrc[4];   // DMA: this lane's source of pieces K0, K1, V0, V1 for the tile to fetch next
  int64_t dinc[4];
  const unsigned lds0 = (unsigned)(uintptr_t)((ATTN_LDS_AS char*)smem);
  auto dma_setup = [&](int t) {  // (re)derive the per-lane sources for tile t (also masks rows past Skv in a ragged tile)
    const bf16_t* zero = (const bf16_t*)g_attn_zero;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * (wave + 4 * i) + (lane >> 3), pc = lane & 7, c = pc ^ ((row >> 1) & 7);
      const bool kok = c * 8 < a.d && t * KVB + row < a.Skv;
      dsrc[i] = kok ? a.K + ((int64_t)b * a.Skv + t * KVB + row) * a.ldk + hoff + c * 8 : zero;
      dinc[i] = kok ? (int64_t)KVB * a.ldk : 0;
      const bool vok = row < attn_dvpad(a.d);
      dsrc[2 + i] = vok ? a.VT + (((int64_t)b * a.H + h) * attn_dvpad(a.d) + row) * SP + t * KVB + c * 8 : zero;
      dinc[2 + i] = vok ? KVB : 0;
    }
  };
  auto dma_issue = [&](int t) {  // tile t -> stage t&1; afterwards the sources point at tile t+1
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (t & 1) * STAGEB + wave * 1024);
    if (ragged && t == nt - 1) dma_setup(t);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      attn_glds16(dsrc[i], base + i * 4096);
      attn_glds16(dsrc[2 + i], base + KBYTES + i * 4096);
      dsrc[i] += dinc[i];
      dsrc[2 + i] += dinc[2 + i];
    }
  };
  if constexpr (DMA) {
    dma_setup(0);
    dma_issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    rows_g2r<DK>(rk, a.K, a.ldk, b, a.Skv, 0, hoff, a.d, tid);
    tr_g2r<DV>(rv, a.VT, b, a.H, h, SP, 0, tid);
    rows_r2s<DK>(smem, rk, tid);
    tr_r2s<DV>(smem + RowTile<DK>::BYTES, rv, tid);
    if (nt > 1) {
      rows_g2r<DK>(rk, a.K, a.ldk, b, a.Skv, KVB, hoff, a.d, tid);
      tr_g2r<DV>(rv, a.VT, b, a.H, h, SP, KVB, tid);
    }
  }
  auto stage = [&](int t) {
    __syncthreads();  // stage t&1 is complete; every wave is done with stage (t+1)&1
    sK = smem + (t & 1) * STAGEB;
    sV = sK + KBYTES;
    if constexpr (DMA) {
      if (t + 1 < nt) dma_issue(t + 1);
    } else {
      if (t + 1 < nt) {
        char* nK = smem + ((t + 1) & 1) * STAGEB;
        rows_r2s<DK>(nK, rk, tid);
        tr_r2s<DV>(nK + RowTile<DK>::BYTES, rv, tid);
        if (t + 2 < nt) {
          rows_g2r<DK>(rk, a.K, a.ldk, b, a.Skv, (t + 2) * KVB, hoff, a.d, tid);
          tr_g2r<DV>(rv, a.VT, b, a.H, h, SP, (t + 2) * KVB, tid);
        }
      }
    }
  };
  auto landed = [&]() {  // DMA: this wave's pieces of the next tile are in LDS (the next barrier publishes them)
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  using MFirst = std::integral_constant<int, T_FIRST>;
  using MFast = std::integral_constant<int, T_FAST>;
  using MSlow = std::integral_constant<int, T_SLOW>;
  static_assert(QF == QP, "the FAST -> SLOW hand-over assumes one tile_pair per KV tile");
  // complete tiles [0, nfull) take the unmasked bodies; a ragged last tile is handled after the loops with the masked
  // body (keeping it out of the loops leaves each loop a single straight-line body)
  const int nfull = ragged ? nt - 1 : nt;
  if (nfull > 0) {
    stage(0);
    tile_pair(std::false_type{}, MFirst{}, 0, 0, 1);
    landed();
    for (int t = 1; t < nfull; ++t) {
      stage(t);
      tile_pair(std::false_type{}, MFast{}, 0, t, t == nfull - 1 ? 1 : 0);
      landed();
    }
  }
  if (ragged) {
    stage(nt - 1);
    if (nt == 1) tile_pair(std::true_type{}, MFirst{}, 0, 0, 1);
    else tile_pair(std::true_type{}, MSlow{}, 0, nt - 1, 1);
  }
  // ---- epilogue ----
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last in-place MFMAs (inline asm) must have landed before VALU reads O
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int q = q0 + 16 * f + j;
    float lt;
    if (ONES) {  // denominator sits in the accumulator row dd = d: fragment d/16, lane group (d%16)/4, reg 0
      const int dfo = a.d >> 4, go = (a.d & 15) >> 2;
      float v = 0.f;
#pragma unroll
      for (int df = 0; df < DF; ++df)
        if (df == dfo) v = acc_o[df][f][0];
      lt = __shfl(v, go * 16 + j, 64);
    } else {
      lt = xor_sum(l[f]);
    }
    const float inv = 1.f / lt;
    if (q < a.Sq) {
      if (a.lse && g == 0) a.lse[((int64_t)b * a.H + h) * a.Sq + q] = m[f] + log2f(lt);
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        const int dd = 16 * df + 4 * g;
        if (dd < a.d) {
          uint2 pk;
          pk.x = pack2bf(acc_o[df][f][0] * inv, acc_o[df][f][1] * inv);
          pk.y = pack2bf(acc_o[df][f][2] * inv, acc_o[df][f][3] * inv);
          *(uint2*)(a.out + ((int64_t)b * a.Sq + q) * a.ldout + hoff + dd) = pk;
        }
      }
    }
  }
}

// =============================================================================================
// backward dQ:  per query block, loop over key tiles.
//   P^T = exp2(sc K Q^T - lse),  dP^T = V dO^T,  dS^T = P^T (dP^T - delta) scale,  dQ^T += K^T dS^T
// =============================================================================================
template <int DK, int DV, int QF>
__global__ __launch_bounds__(256, ((DK <= 64 || (DK <= 96 && QF == 1)) ? 2 : 1)) void attn_bwd_dq_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sVr = sK + RowTile<DK>::BYTES;
  char* sKT = sVr + RowTile<DK>::BYTES;
  constexpr int KS = DK / 32, DF = DV / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int hoff = h * a.d;
  const int q0 = blockIdx.x * (64 * QF) + wave * (16 * QF);
  const int SP = attn_spad(a.Skv);
  const float sc = a.scale * 1.4426950408889634f;

  bf16x8 qf[QF][KS], dof[QF][KS];
  float lse[QF], dl[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int q = q0 + 16 * f + j;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[f][ks] = scaled_frag(own_frag(a.Q, a.ldq, b, a.Sq, q, hoff, a.d, ks, g), sc);  // S comes out in exp2 units
      dof[f][ks] = own_frag(a.dO, a.lddo, b, a.Sq, q, hoff, a.d, ks, g);
    }
    const bool ok = q < a.Sq;
    lse[f] = ok ? a.lse[((int64_t)b * a.H + h) * a.Sq + q] : 1.0e30f;
    dl[f] = ok ? a.delta[((int64_t)b * a.H + h) * a.Sq + q] : 0.f;
  }
  f32x4 acc[DF][QF];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int df = 0; df < DF; ++df) acc[df][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 rk[RowTile<DK>::NREG], rv[RowTile<DK>::NREG], rt[TrTile<DV>::NREG];
  const int nt = (a.Skv + KVB - 1) / KVB;
  const bool ragged = (a.Skv % KVB) != 0;
  // one key tile; TAIL masks keys >= Skv (only the ragged last tile pays for it).  P^T = exp2(S - lse) with -lse as the
  // MFMA C operand; dS^T = P^T (dP^T - delta) -- the softmax scale is applied once, to dQ, in the epilogue
  auto tile = [&](auto tail_tag, int t) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    f32x4 s[4][QF], dp[4][QF];
    f32x4 cinit[QF];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < QF; ++f) cinit[f] = (f32x4){-lse[f], -lse[f], -lse[f], -lse[f]};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const bf16x8 kfr = row_frag<DK>(sK, kf, ks, g, j);
        const bf16x8 vfr = row_frag<DK>(sVr, kf, ks, g, j);
#pragma unroll
        for (int f = 0; f < QF; ++f) {
          s[kf][f] = MFMA(kfr, qf[f][ks], ks == 0 ? cinit[f] : s[kf][f]);
          dp[kf][f] = MFMA(vfr, dof[f][ks], ks == 0 ? zero4 : dp[kf][f]);
        }
      }
    }
    const int kvbase = t * KVB + 4 * g;
    bf16x8 ds[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = fast_exp2(s[kf][f][r]);
          if (TAIL && kvbase + 16 * kf + r >= a.Skv) p = 0.f;
          s[kf][f][r] = p * (dp[kf][f][r] - dl[f]);
        }
      ds[f][0] = pack8(s[0][f], s[1][f]);
      ds[f][1] = pack8(s[2][f], s[3][f]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        const bf16x8 ktf = tr_frag(sKT, df, s2, g, j);
#pragma unroll
        for (int f = 0; f < QF; ++f) acc[df][f] = MFMA(ktf, ds[f][s2], acc[df][f]);
      }
  };
  auto fetch = [&](int t) {
    rows_g2r<DK>(rk, a.K, a.ldk, b, a.Skv, t * KVB, hoff, a.d, tid);
    rows_g2r<DK>(rv, a.V, a.ldv, b, a.Skv, t * KVB, hoff, a.d, tid);
    tr_g2r<DV>(rt, a.KT, b, a.H, h, SP, t * KVB, tid);
  };
  fetch(0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();
    rows_r2s<DK>(sK, rk, tid);
    rows_r2s<DK>(sVr, rv, tid);
    tr_r2s<DV>(sKT, rt, tid);
    __syncthreads();
    if (t + 1 < nt) fetch(t + 1);  // the next tile's global loads fly during this tile's math
    if (ragged && t == nt - 1) tile(std::true_type{}, t);
    else tile(std::false_type{}, t);
  }
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[df][f][r] *= a.scale;
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int q = q0 + 16 * f + j;
    if (q < a.Sq) {
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        const int dd = 16 * df + 4 * g;
        if (dd < a.d) {
          uint2 pk;
          pk.x = pack2bf(acc[df][f][0], acc[df][f][1]);
          pk.y = pack2bf(acc[df][f][2], acc[df][f][3]);
          *(uint2*)(a.out + ((int64_t)b * a.Sq + q) * a.ldout + hoff + dd) = pk;
        }
      }
    }
  }
}

// =============================================================================================
// backward dK/dV: per key block (each wave KF key fragments), loop over query tiles.
//   S = Q K^T (un-swapped: lane (g, j=kv) holds q = 16qf+4g+r),  P = exp2(sc S - lse[q])
//   dP = dO V^T,  dS = P (dP - delta[q]) scale,  dV^T += dO^T P,  dK^T += Q^T dS
// =============================================================================================
template <int DK, int DV, int KF>
__global__ __launch_bounds__(256, ((DK <= 64 || (DK <= 96 && KF == 1)) ? 2 : 1)) void attn_bwd_dkv_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;
  char* sdO = sQ + RowTile<DK>::BYTES;
  char* sQT = sdO + RowTile<DK>::BYTES;
  char* sdOT = sQT + TrTile<DV>::BYTES;
  float* sL = (float*)(sdOT + TrTile<DV>::BYTES);  // [64] lse, [64] delta
  constexpr int KS = DK / 32, DF = DV / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int hoff = h * a.d;
  const int kv0 = blockIdx.x * (64 * KF) + wave * (16 * KF);
  const int SPq = attn_spad(a.Sq);
  const float sc = a.scale * 1.4426950408889634f;

  bf16x8 kf_[KF][KS], vf_[KF][KS];
#pragma unroll
  for (int f = 0; f < KF; ++f)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf_[f][ks] = scaled_frag(own_frag(a.K, a.ldk, b, a.Skv, kv0 + 16 * f + j, hoff, a.d, ks, g), sc);  // S in exp2 units
      vf_[f][ks] = own_frag(a.V, a.ldv, b, a.Skv, kv0 + 16 * f + j, hoff, a.d, ks, g);
    }
  f32x4 adk[DF][KF], adv[DF][KF];
#pragma unroll
  for (int f = 0; f < KF; ++f)
#pragma unroll
    for (int df = 0; df < DF; ++df) {
      adk[df][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      adv[df][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  uint4 rq[RowTile<DK>::NREG], rdo[RowTile<DK>::NREG], rqt[TrTile<DV>::NREG], rdot[TrTile<DV>::NREG];
  const int nt = (a.Sq + KVB - 1) / KVB;
  float nl_r = 0.f, dl_r = 0.f;  // (threads < 64) -lse / delta of the staged query tile
  auto fetch = [&](int t) {
    rows_g2r<DK>(rq, a.Q, a.ldq, b, a.Sq, t * KVB, hoff, a.d, tid);
    rows_g2r<DK>(rdo, a.dO, a.lddo, b, a.Sq, t * KVB, hoff, a.d, tid);
    tr_g2r<DV>(rqt, a.QT, b, a.H, h, SPq, t * KVB, tid);
    tr_g2r<DV>(rdot, a.dOT, b, a.H, h, SPq, t * KVB, tid);
    if (tid < 64) {
      const int q = t * KVB + tid;
      const bool ok = q < a.Sq;
      nl_r = ok ? -a.lse[((int64_t)b * a.H + h) * a.Sq + q] : -1.0e30f;  // negated: it is the MFMA C operand
      dl_r = ok ? a.delta[((int64_t)b * a.H + h) * a.Sq + q] : 0.f;
    }
  };
  fetch(0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();
    rows_r2s<DK>(sQ, rq, tid);
    rows_r2s<DK>(sdO, rdo, tid);
    tr_r2s<DV>(sQT, rqt, tid);
    tr_r2s<DV>(sdOT, rdot, tid);
    if (tid < 64) {
      sL[tid] = nl_r;
      sL[64 + tid] = dl_r;
    }
    __syncthreads();
    if (t + 1 < nt) fetch(t + 1);  // the next tile's global loads fly during this tile's math
    f32x4 s[4][KF], dp[4][KF];
    f32x4 nls[4];  // -lse of this lane's four query rows per query fragment: the C operand of the S MFMAs
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) nls[qf] = *(const f32x4*)(sL + 16 * qf + 4 * g);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int qf = 0; qf < 4; ++qf) {
        const bf16x8 qfr = row_frag<DK>(sQ, qf, ks, g, j);
        const bf16x8 dofr = row_frag<DK>(sdO, qf, ks, g, j);
#pragma unroll
        for (int f = 0; f < KF; ++f) {
          s[qf][f] = MFMA(qfr, kf_[f][ks], ks == 0 ? nls[qf] : s[qf][f]);
          dp[qf][f] = MFMA(dofr, vf_[f][ks], ks == 0 ? zero4 : dp[qf][f]);
        }
      }
    }
    bf16x8 pb[KF][2], dsb[KF][2];
#pragma unroll
    for (int f = 0; f < KF; ++f) {
#pragma unroll
      for (int qf = 0; qf < 4; ++qf) {
        const float4 dl = *(const float4*)(sL + 64 + 16 * qf + 4 * g);
        const float dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(s[qf][f][r]);
          s[qf][f][r] = p;
          dp[qf][f][r] = p * (dp[qf][f][r] - dlv[r]);  // the softmax scale is applied once, to dK, in the epilogue
        }
      }
      pb[f][0] = pack8(s[0][f], s[1][f]);
      pb[f][1] = pack8(s[2][f], s[3][f]);
      dsb[f][0] = pack8(dp[0][f], dp[1][f]);
      dsb[f][1] = pack8(dp[2][f], dp[3][f]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        const bf16x8 dotf = tr_frag(sdOT, df, s2, g, j);
        const bf16x8 qtf = tr_frag(sQT, df, s2, g, j);
#pragma unroll
        for (int f = 0; f < KF; ++f) {
          adv[df][f] = MFMA(dotf, pb[f][s2], adv[df][f]);
          adk[df][f] = MFMA(qtf, dsb[f][s2], adk[df][f]);
        }
      }
  }
#pragma unroll
  for (int f = 0; f < KF; ++f)
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
      for (int r = 0; r < 4; ++r) adk[df][f][r] *= a.scale;
#pragma unroll
  for (int f = 0; f < KF; ++f) {
    const int kv = kv0 + 16 * f + j;
    if (kv < a.Skv) {
#pragma unroll
      for (int df = 0; df < DF; ++df) {
        const int dd = 16 * df + 4 * g;
        if (dd < a.d) {
          uint2 pk;
          pk.x = pack2bf(adk[df][f][0], adk[df][f][1]);
          pk.y = pack2bf(adk[df][f][2], adk[df][f][3]);
          *(uint2*)(a.dK + ((int64_t)b * a.Skv + kv) * a.lddk + hoff + dd) = pk;
          pk.x = pack2bf(adv[df][f][0], adv[df][f][1]);
          pk.y = pack2bf(adv[df][f][2], adv[df][f][3]);
          *(uint2*)(a.dV + ((int64_t)b * a.Skv + kv) * a.lddv + hoff + dd) = pk;
        }
      }
    }
  }
}

// =============================================================================================
// X[B,S,ld] (head h at column h*d) -> XT[B,H,DVP,SP], zero padded.  64x(16-dd) LDS transpose.
// =============================================================================================
__global__ __launch_bounds__(256) void transpose_heads_kernel(const bf16_t* X, int64_t ld, bf16_t* XT,
                                                              int B, int H, int S, int d, int ones_row) {
  __shared__ bf16_t tile[64][66];
  const int DVP = attn_dvpad(d), SP = attn_spad(S);
  const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
  const int bh = blockIdx.z, b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x;
  // load 64 rows (s) x 64 cols (dd) -> tile[s][dd]
  for (int i = tid; i < 64 * 8; i += 256) {
    const int r = i >> 3, c = (i & 7) * 8;
    const int s = s0 + r, dd = d0 + c;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s < S && dd < d) v = *(const u16x8*)(X + ((int64_t)b * S + s) * ld + h * d + dd);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][c + e] = v[e];
  }
  __syncthreads();
  for (int i = tid; i < 64 * 8; i += 256) {
    const int r = i >> 3, c = (i & 7) * 8;  // r = dd, c = s offset
    const int dd = d0 + r;
    if (dd < DVP) {
      u16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[c + e][r];
      if (ones_row && dd == d) {  // bf16 1.0 for valid keys: the PV product then also yields sum_k P
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (s0 + c + e < S) ? (bf16_t)0x3F80 : (bf16_t)0;
      }
      *(u16x8*)(XT + (((int64_t)b * H + h) * DVP + dd) * SP + s0 + c) = v;
    }
  }
}

__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* O, int64_t ldo, const bf16_t* dO,
                                                         int64_t lddo, float* delta, int B, int H, int S,
                                                         int d) {
  // one 8-lane group per (b, s, h)
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const int64_t total = (int64_t)B * S * H;
  float acc = 0.f;
  if (gid < total) {
    const int h = (int)(gid % H);
    const int64_t bs = gid / H;
    for (int c = sub * 8; c < d; c += 64) {
      const u16x8 o = *(const u16x8*)(O + bs * ldo + h * d + c);
      const u16x8 g = *(const u16x8*)(dO + bs * lddo + h * d + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += bf2f(o[e]) * bf2f(g[e]);
    }
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (gid < total && sub == 0) {
    const int h = (int)(gid % H);
    const int64_t bs = gid / H;
    const int b = (int)(bs / S), s = (int)(bs - (int64_t)b * S);
    delta[((int64_t)b * H + h) * S + s] = acc;
  }
}

// =============================================================================================
// forward on 32x32x16 MFMAs (round 3), head dims <= 64, LDS-DMA staged K / V^T tiles as above.
//
// Why a second forward kernel: the 16x16x32 kernel is ISSUE-bound at d = 40 -- 28 MFMAs and ~136 other instructions per
// (64 keys x 32 queries) wave tile, and a SIMD issues one instruction per ~4 cycles whatever pipe it goes to, so the matrix pipe
// idles while exp / max / cvt / ds_read issue (VALU 61 % + MFMA 28 % busy).  32x32x16 MFMAs do the same
// FLOPs in HALF the instructions (14 per tile: 2 key blocks x KS k-steps of K Q^T with the head dim padded to 16 KS instead of 64,
// 2 d-blocks x 4 key steps of V^T P), a query's 64 scores of a tile sit in TWO lanes (l, l ^ 32) instead of four, so the row
// maximum is one cross-lane step, and the 8 consecutive C registers of a (key block, half) ARE the B operand of one V^T P step:
// no cross-lane traffic between the two products.  All MFMAs are compiler builtins: hazards and accumulator placement are the
// compiler's job (the inline-asm accumulation of the 16x16 kernel needed hand-placed wait states).
//
// Fragment maps (32x32x16 bf16): A[i = l & 31][k = 8 (l >> 5) + e], B[k = 8 (l >> 5) + e][j = l & 31], e = 0..7;
// C[row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)][col = l & 31].  S^T = K Q^T: rows = keys, cols = queries; lane l holds, for query
// l & 31, the keys 8 i + 4 h + r (h = l >> 5) of each 32-key block in reg 4 i + r.  O^T = V^T P: the contraction slot e of half h in
// key step s = 2 kb + hs is key 32 kb + 16 hs + 8 (e >> 2) + 4 h + (e & 3) = the key of C register 8 hs + e of block kb: P feeds
// straight from the registers, and the V^T fragment reads the same permutation from LDS (two 8-byte reads, 16 bytes apart).
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
__device__ __forceinline__ float other32(float v) {  // value of lane l ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

// KS: 16-wide k-steps of K Q^T (3: d <= 48, 4: d <= 64).  ONES: V^T carries a row of ones at dd = d (< 64): the denominator
// accumulates in O^T row d; otherwise a VALU row sum.
//
// Issue-slot economy of the tile body (what this kernel is about):
//  * the K tile is staged with its rows PERMUTED inside every group of 16 keys (row rho holds key rho with bits 2 and 3 swapped):
//    softmax does not care about key order, and with that order the 8 contraction slots of a lane half in a V^T P step are 8
//    CONSECUTIVE keys of V^T's (memory-order) tile -- one ds_read_b128 per V^T fragment instead of two ds_read_b64;
//  * the two LDS stages are two straight-line copies of the body (template SLOT): every LDS address is lane offset + immediate;
//  * -m lives in a register block that is the read-only C operand of the first K Q^T MFMA of a block (no per-tile accumulator
//    initialisation);
//  * the lazy-maximum check of the common tile does not compute a maximum at all: the probabilities are packed to bf16 anyway,
//    bf16 bit patterns of non-negative numbers order like integers, so a chain of packed 16-bit maxima over the 16 packed
//    registers (v_pk_max_u16) tells whether any probability exceeded 2^TAU.  Only then (rare, wave-uniform) the tile is
//    recomputed from LDS with the classic maximum / re-base sequence.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmax(uint32_t x, uint32_t y) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2_t, x), __builtin_bit_cast(us2_t, y)));
}
__host__ __device__ __forceinline__ int attn32_keyperm(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
// LDS-DMA piece with a uniform 64-bit base (SGPR pair) + a 32-bit lane offset: the pointers advance on the scalar unit
__device__ __forceinline__ void attn_glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
  // (the base is wave-uniform by construction; say so explicitly -- the divergence analysis does not always see it through a tile loop)
  const uint64_t pv = (uint64_t)(uintptr_t)sbase;
  const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pv), phi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pv >> 32));
  sbase = (const void*)(uintptr_t)(((uint64_t)phi << 32) | (uint64_t)plo);   // (readfirstlane returns int: widen as unsigned)
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr));
}
__device__ __forceinline__ float round_bf16(float x) { return __uint_as_float(pack2bf(x, 0.f) << 16); }

// MT ("m through the contraction", d % 8 == 0 and d < 16 KS): the running reference maximum m rides in the first padded k slot:
// K'[key][d] = 1 for every key (written into the LDS tile's padding ONCE -- the DMA never touches padded chunks), Q'[q][d] = -m
// (bf16; m is kept bf16-representable so that the subtraction is exact), so K' Q'^T = s - m with a ZERO C operand: no register
// block for -m, no per-tile accumulator initialisation -- 16 registers less, which is what lets three waves share a SIMD.
//
// Head dims 64 < d <= 80 (KS = 5, DB = 3: PixArt's d = 72, SD1.5's d = 80 level): the fifth k-step's 16 columns live in a second K
// sub-tile of 32-byte rows (K_B: key rho at rho * 32, two chunks), V^T carries a third block of 32 rows; the 16x16x32 kernel pads
// these heads to 96 on both products (25 % idle MFMA work), here K Q^T runs 80 wide and V^T P 96 rows.
template <int KS, int DB>
struct Attn32Lds {
  static constexpr int KA = 64 * 128, VT = DB * 32 * 128, KB = KS > 4 ? 64 * 32 : 0, STAGE = KA + VT + KB;
};
// (round 5: raising the wave's issue priority with s_setprio around the MFMA groups, with and without scheduling barriers, changed
// nothing: 912 vs 910 vs 918 us at B = 32, d = 40 -- profiles/r5_attn_setprio.txt; the variants were removed)
template <int KS, int DB, bool ONES, bool MT>
__global__ __launch_bounds__(256, (MT && ONES && KS < 5) ? 3 : 2) void attn_fwd32_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = Attn32Lds<KS, DB>;
  constexpr int KBYTES = L::KA, STAGEB = L::STAGE, KBOFF = L::KA + L::VT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, n = lane & 31;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh - b * a.H;
  const int hoff = hd * a.d;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int SP = attn_spad(a.Skv), DVP = attn_dvpad(a.d);
  const float sc = a.scale * 1.4426950408889634f;

  // padding of the K tiles (columns from d on) and V^T rows from dvpad(d) on, both stages: written once, never by the DMA
  for (int idx = tid; idx < 2 * (STAGEB / 16); idx += 256) {
    const int stg = idx / (STAGEB / 16), off = (idx - stg * (STAGEB / 16)) * 16;
    bool pad, one;
    if (off < KBYTES) {                 // K_A: row of 128 B, logical chunk = physical ^ swizzle
      const int row = off >> 7, c = ((off >> 4) & 7) ^ ((row >> 1) & 7);
      pad = c * 8 >= a.d;
      one = c * 8 == a.d;
    } else if (off < KBOFF) {           // V^T rows
      pad = ((off - KBYTES) >> 7) >= DVP;
      one = false;
    } else {                            // K_B: key rho at rho * 32, chunk cB = columns 64 + 8 cB ..
      const int cB = ((off - KBOFF) >> 4) & 1;
      pad = 64 + 8 * cB >= a.d;
      one = 64 + 8 * cB == a.d;
    }
    if (pad) {
      uint4 val = make_uint4(0, 0, 0, 0);
      if (MT && one) val.x = 0x3F80u;   // bf16 1.0 at column d
      *(uint4*)(smem + stg * STAGEB + off) = val;
    }
  }

  // Q fragments (B operand of K Q^T), scaled by scale * log2(e): the MFMA output is the exp2 argument
  union QF { bf16x8 v; uint4 u4; uint32_t u[4]; } qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks].u4 = make_uint4(0, 0, 0, 0);
    const int c = 16 * ks + 8 * h, row = q0 + n;
    if (row < a.Sq && c < a.d) qf[ks].u4 = *(const uint4*)(a.Q + ((int64_t)b * a.Sq + row) * a.ldq + hoff + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      qf[ks].u[e] = pack2bf(__uint_as_float(qf[ks].u[e] << 16) * sc, __uint_as_float(qf[ks].u[e] & 0xffff0000u) * sc);
  }
  f32x16 o[DB], cinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db][r] = 0.f;
    cinit[r] = 0.f;
  }
  float m = NEG_BIG, lsum = 0.f;
  auto set_m = [&](float mv) __attribute__((always_inline)) {   // publish the reference maximum to the next K Q^T
    m = mv;
    if (MT) {
      const uint32_t nb = pack2bf(-mv, 0.f) & 0xffffu;
      const bool mine = h == ((a.d >> 3) & 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (mine && ks == (a.d >> 4)) qf[ks].u[0] = (qf[ks].u[0] & 0xffff0000u) | nb;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[r] = -mv;
    }
  };

  // per-lane fragment offsets inside a tile (rows of 128 B; chunk c of row r sits at physical chunk c ^ ((r >> 1) & 7)); the second
  // 32-row block is the same offset + 4096 (the swizzle term only sees n)
  int kofs[KS], vofs[4];
  {
    const int sw = (n >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      kofs[ks] = ks < 4 ? n * 128 + (((2 * ks + h) ^ sw) * 16) : KBOFF + n * 32 + h * 16;   // (block kb: + 4096 kb in K_A, + 1024 kb in K_B)
#pragma unroll
    for (int st = 0; st < 4; ++st) vofs[st] = KBYTES + n * 128 + (((2 * st + h) ^ sw) * 16);   // keys 16 st + 8 h .. + 7
  }

  const int nt = (a.Skv + KVB - 1) / KVB;
  const bool ragged = (a.Skv % KVB) != 0;
  // ---- LDS-DMA staging: 4 pieces of 1 KiB per wave per tile (K rows 8 (wave + 4 i) .., V^T rows likewise); K rows permuted.  Uniform
  // tile bases advance on the scalar unit, the lane offsets are fixed; padded K chunks / V^T rows are not fetched (lanes off) ----
  const int drow = 8 * wave + (lane >> 3), dpc = lane & 7, dc = dpc ^ ((drow >> 1) & 7);   // (row + 32: same swizzle term)
  const bool kok = dc * 8 < a.d;
  unsigned koff[2], voff[DB];
#pragma unroll
  for (int i = 0; i < 2; ++i) koff[i] = (unsigned)((attn32_keyperm(drow + 32 * i) * a.ldk + dc * 8) * 2);
#pragma unroll
  for (int i = 0; i < DB; ++i) voff[i] = (unsigned)(((drow + 32 * i) * SP + dc * 8) * 2);
  // K_B (KS = 5): 64 keys x 32 B = two pieces, issued by waves 0 and 1: lane l -> LDS row 32 wave + l / 2, chunk l % 2
  const int brow = 32 * (wave & 1) + (lane >> 1), bc = lane & 1;
  const bool bok = KS > 4 && wave < 2 && 64 + 8 * bc < a.d;
  const unsigned kboff = (unsigned)((attn32_keyperm(brow) * a.ldk + 64 + 8 * bc) * 2);
  const char* kbase = (const char*)(a.K + ((int64_t)b * a.Skv) * a.ldk + hoff);
  const char* vbase = (const char*)(a.VT + (((int64_t)b * a.H + hd) * DVP) * SP);
  const int64_t kstep = (int64_t)KVB * a.ldk * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)((ATTN_LDS_AS char*)smem);
  auto dma_issue = [&](int t) __attribute__((always_inline)) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (t & 1) * STAGEB + wave * 1024);
    const bool last_ragged = ragged && t == nt - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool keyok = !last_ragged || t * KVB + attn32_keyperm(drow + 32 * i) < a.Skv;
      if (kok && keyok) attn_glds16_s(kbase, koff[i], base + i * 4096);
    }
#pragma unroll
    for (int i = 0; i < DB; ++i)
      if (8 * (wave + 4 * i) < DVP) attn_glds16_s(vbase, voff[i], base + KBYTES + i * 4096);
    if (KS > 4) {
      const bool keyok = !last_ragged || t * KVB + attn32_keyperm(brow) < a.Skv;
      if (bok && keyok) attn_glds16_s(kbase, kboff, base + KBOFF);     // (base already carries wave * 1024)
    }
    kbase += kstep;
    vbase += KVB * 2;
  };
  dma_issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  constexpr float TAU = 16.f;
  constexpr uint32_t TAU_BF16 = (uint32_t)(127 + 16) << 7;   // bf16 bits of 2^TAU
  // one S^T block (32 keys) of the tile at `base`: scores minus the reference maximum (MT: through the contraction; else the C operand)
  auto qk_block = [&](const char* base, int kb) __attribute__((always_inline)) -> f32x16 {
    bf16x8 kfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfr[ks] = *(const bf16x8*)(base + (ks < 4 ? kb * 4096 : kb * 1024) + kofs[ks]);
    f32x16 acc;
    if (MT) {
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      acc = MFMA32(kfr[0], qf[0].v, z);
    } else {
      acc = MFMA32(kfr[0], qf[0].v, cinit);
    }
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) acc = MFMA32(kfr[ks], qf[ks].v, acc);
    return acc;
  };
  auto qk = [&](auto slot_tag, f32x16 (&s)[2]) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    s[0] = qk_block(smem + SLOT * STAGEB, 0);
    s[1] = qk_block(smem + SLOT * STAGEB, 1);
  };
  auto pack = [&](const f32x16 (&s)[2], uint4 (&pb)[4]) __attribute__((always_inline)) {   // key step st = 2 kb + hs <- C registers 8 hs .. 8 hs + 7 of block kb
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int kb = st >> 1, r0 = 8 * (st & 1);
      pb[st] = make_uint4(pack2bf(s[kb][r0], s[kb][r0 + 1]), pack2bf(s[kb][r0 + 2], s[kb][r0 + 3]), pack2bf(s[kb][r0 + 4], s[kb][r0 + 5]),
                          pack2bf(s[kb][r0 + 6], s[kb][r0 + 7]));
    }
  };
  auto pv = [&](auto slot_tag, const uint4 (&pb)[4]) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const char* base = smem + SLOT * STAGEB;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      union { uint4 u; bf16x8 v; } p8;
      p8.u = pb[st];
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + vofs[st]), p8.v, o[db]);
    }
  };
  // classic body: row maximum, re-base m (FIRST: m := the maximum), exp2, row sum.  Used for tile 0, the ragged tail and the rare
  // tile whose scores run more than TAU past the reference maximum.  s arrives relative to the current m (0 before tile 0).
  auto classic = [&](auto first_tag, auto tail_tag, f32x16 (&s)[2], int t) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value, TAIL = decltype(tail_tag)::value;
    if (TAIL) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KVB + attn32_keyperm(32 * kb + 8 * (r >> 2) + 4 * h + (r & 3)) >= a.Skv) s[kb][r] = NEG_BIG;
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
    mx = fmaxf(mx, other32(mx));
    float up, alpha = 1.f;
    if (FIRST) {
      up = MT ? round_bf16(mx) : mx;
      set_m(up);
    } else {
      const float mnew = MT ? round_bf16(m + fmaxf(mx, 0.f)) : m + fmaxf(mx, 0.f);   // the reference only ever rises
      up = mnew - m;
      alpha = fast_exp2(-up);
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      set_m(mnew);
    }
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[kb][r] - up);
        s[kb][r] = p;
        if (!ONES) ps += p;
      }
    if (!ONES) lsum = lsum * alpha + ps;
  };
  // common body: no maximum -- exp2 straight away, pack, packed-integer overflow check; redo classically if it fires.  The tile is
  // processed as two halves of 32 keys (the two S^T blocks): both blocks' K Q^T MFMAs are issued first, then per half exp2 / pack /
  // check / its four V^T P MFMAs -- half 0's exponentials run under block 1's K Q^T, half 1's under half 0's V^T P.
  auto rebase_half = [&](const f32x16 sh) __attribute__((always_inline)) -> float {   // sh relative to the current m; returns the shift applied
    float mx = sh[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sh[r]);
    mx = fmaxf(mx, other32(mx));
    const float mnew = MT ? round_bf16(m + fmaxf(mx, 0.f)) : m + fmaxf(mx, 0.f);
    const float up = mnew - m, alpha = fast_exp2(-up);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    if (!ONES) lsum *= alpha;
    set_m(mnew);
    return up;
  };
  // one half: sh = this block's scores (exp2 arguments), snext = the other block's (adjusted if this half re-bases and KB == 0)
  auto half = [&](const char* base, auto kb_tag, f32x16 sh, f32x16& snext) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_tag)::value;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sh[r] = fast_exp2(sh[r]);
      if (!ONES) ps += sh[r];
    }
    uint4 pb0 = make_uint4(pack2bf(sh[0], sh[1]), pack2bf(sh[2], sh[3]), pack2bf(sh[4], sh[5]), pack2bf(sh[6], sh[7]));
    uint4 pb1 = make_uint4(pack2bf(sh[8], sh[9]), pack2bf(sh[10], sh[11]), pack2bf(sh[12], sh[13]), pack2bf(sh[14], sh[15]));
    const uint32_t c = pkmax(pkmax(pkmax(pb0.x, pb0.y), pkmax(pb0.z, pb0.w)), pkmax(pkmax(pb1.x, pb1.y), pkmax(pb1.z, pb1.w)));
    const bool over = (c & 0xffffu) > TAU_BF16 || (c >> 16) > TAU_BF16;
    if (__builtin_expect(__any(over), 0)) {
      asm volatile("" ::: "memory");   // keep the block conditional (no speculation into the common path)
      sh = qk_block(base, KB);         // recompute (relative to the m the block was issued with), re-base, redo the exponentials
      const float up = rebase_half(sh);
      ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sh[r] = fast_exp2(sh[r] - up);
        if (!ONES) ps += sh[r];
      }
      if (KB == 0) {   // block 1's scores were issued against the old reference
#pragma unroll
        for (int r = 0; r < 16; ++r) snext[r] -= up;
      }
      pb0 = make_uint4(pack2bf(sh[0], sh[1]), pack2bf(sh[2], sh[3]), pack2bf(sh[4], sh[5]), pack2bf(sh[6], sh[7]));
      pb1 = make_uint4(pack2bf(sh[8], sh[9]), pack2bf(sh[10], sh[11]), pack2bf(sh[12], sh[13]), pack2bf(sh[14], sh[15]));
    }
    if (!ONES) lsum += ps;
    union { uint4 u; bf16x8 v; } p0, p1;
    p0.u = pb0;
    p1.u = pb1;
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + vofs[2 * KB]), p0.v, o[db]);
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + vofs[2 * KB + 1]), p1.v, o[db]);
  };
  auto fast = [&](auto slot_tag, int t) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const char* base = smem + SLOT * STAGEB;
    f32x16 s0 = qk_block(base, 0);
    f32x16 s1 = qk_block(base, 1);
    half(base, std::integral_constant<int, 0>{}, s0, s1);
    half(base, std::integral_constant<int, 1>{}, s1, s1);
  };
  auto slow = [&](auto slot_tag, auto first_tag, auto tail_tag, int t) __attribute__((always_inline)) {
    f32x16 s[2];
    uint4 pb[4];
    qk(slot_tag, s);
    classic(first_tag, tail_tag, s, t);
    pack(s, pb);
    pv(slot_tag, pb);
  };

  auto stage = [&](int t) __attribute__((always_inline)) {
    __syncthreads();  // stage t & 1 is complete (and, before tile 0, the padding); every wave is done with the other stage
    if (t + 1 < nt) dma_issue(t + 1);
  };
  auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int nfull = ragged ? nt - 1 : nt;
  if (nfull > 0) {
    stage(0);
    slow(S0{}, std::true_type{}, std::false_type{}, 0);
    landed();
    int t = 1;
    for (; t + 1 < nfull; t += 2) {   // two tiles per trip: stage 1, then stage 0
      stage(t);
      fast(S1{}, t);
      landed();
      stage(t + 1);
      fast(S0{}, t + 1);
      landed();
    }
    if (t < nfull) {
      stage(t);
      fast(S1{}, t);
      landed();
    }
  }
  if (ragged) {
    stage(nt - 1);
    if (nt == 1) slow(S0{}, std::true_type{}, std::true_type{}, 0);
    else if ((nt - 1) & 1) slow(S1{}, std::false_type{}, std::true_type{}, nt - 1);
    else slow(S0{}, std::false_type{}, std::true_type{}, nt - 1);
  }
  // ---- epilogue: O[q][dd] = O^T[dd][q] / l ----
  float lt;
  if (ONES) {   // the denominator is O^T row d: block d / 32, register 4 ((d % 32) / 8) + (d % 4), lane half ((d % 8) / 4)
    const int dl = a.d & 31, reg = 4 * (dl >> 3) + (dl & 3), hh = (dl & 7) >> 2;
    float v = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (db == (a.d >> 5) && r == reg) v = o[db][r];
    lt = __shfl(v, hh * 32 + n, 64);
  } else {
    lt = lsum + other32(lsum);
  }
  const float inv = 1.f / lt;
  const int q = q0 + n;
  if (q < a.Sq) {
    if (a.lse && h == 0) a.lse[((int64_t)b * a.H + hd) * a.Sq + q] = m + log2f(lt);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int dd = 32 * db + 8 * i + 4 * h;
        if (dd < a.d) {
          uint2 pk;
          pk.x = pack2bf(o[db][4 * i] * inv, o[db][4 * i + 1] * inv);
          pk.y = pack2bf(o[db][4 * i + 2] * inv, o[db][4 * i + 3] * inv);
          *(uint2*)(a.out + ((int64_t)b * a.Sq + q) * a.ldout + hoff + dd) = pk;
        }
      }
  }
}

// =============================================================================================
// backward dQ on 32x32x16 MFMAs (round 3; head dims <= 80): the forward's structure without the online softmax -- the log-sum-exp is
// known, so P^T = exp2(K Q'^T - lse) directly (C operand = -lse).  Per 64-key tile: S^T and dP^T = V dO^T (V staged row-major with the
// same permuted key rows as K), dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T with K^T fragments as single b128 reads (the forward's
// V^T P product with K^T in V^T's place).  32 queries per wave, 128 per block, two LDS stages filled by LDS-DMA.
// Stage layout: [K_A 8K][V_A 8K][K^T DB x 4K][K_B 2K][V_B 2K] (the _B sub-tiles -- columns 64.. of K / V, 32-byte rows -- only for KS = 5)
template <int KS, int DB>
struct AttnDq32Lds {
  static constexpr int VOFF = 8192, TOFF = 16384, KBOFF = TOFF + DB * 4096, VBOFF = KBOFF + 2048, STAGE = KBOFF + (KS > 4 ? 4096 : 0);
};
template <int KS, int DB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq32_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = AttnDq32Lds<KS, DB>;
  constexpr int STAGEB = L::STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, n = lane & 31;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh - b * a.H;
  const int hoff = hd * a.d;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int SP = attn_spad(a.Skv), DVP = attn_dvpad(a.d);
  const float sc = a.scale * 1.4426950408889634f;

  // the whole LDS image starts as zeros: padded columns / rows are never fetched, and rows of a ragged last tile keep finite values
  for (int idx = tid; idx < 2 * STAGEB / 16; idx += 256) *(uint4*)(smem + idx * 16) = make_uint4(0, 0, 0, 0);

  union QF { bf16x8 v; uint4 u4; uint32_t u[4]; } qf[KS], dof[KS];
  const int qrow = q0 + n;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks].u4 = make_uint4(0, 0, 0, 0);
    dof[ks].u4 = make_uint4(0, 0, 0, 0);
    const int c = 16 * ks + 8 * h;
    if (qrow < a.Sq && c < a.d) {
      qf[ks].u4 = *(const uint4*)(a.Q + ((int64_t)b * a.Sq + qrow) * a.ldq + hoff + c);
      dof[ks].u4 = *(const uint4*)(a.dO + ((int64_t)b * a.Sq + qrow) * a.lddo + hoff + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)   // S comes out in exp2 units
      qf[ks].u[e] = pack2bf(__uint_as_float(qf[ks].u[e] << 16) * sc, __uint_as_float(qf[ks].u[e] & 0xffff0000u) * sc);
  }
  const bool qok = qrow < a.Sq;
  const float nlse = qok ? -a.lse[((int64_t)b * a.H + hd) * a.Sq + qrow] : -1.0e30f;
  const float dl = qok ? a.delta[((int64_t)b * a.H + hd) * a.Sq + qrow] : 0.f;
  f32x16 acc[DB], cinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int db = 0; db < DB; ++db) acc[db][r] = 0.f;
    cinit[r] = nlse;
  }

  int kofs[KS], tofs[4];
  {
    const int sw = (n >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kofs[ks] = ks < 4 ? n * 128 + (((2 * ks + h) ^ sw) * 16) : L::KBOFF + n * 32 + h * 16;
#pragma unroll
    for (int st = 0; st < 4; ++st) tofs[st] = L::TOFF + n * 128 + (((2 * st + h) ^ sw) * 16);
  }

  const int nt = (a.Skv + KVB - 1) / KVB;
  const bool ragged = (a.Skv % KVB) != 0;
  // ---- LDS-DMA staging (scalar tile bases, fixed lane offsets): K_A / V_A rows 8 (wave + 4 i) .. (permuted keys), K^T rows likewise,
  // K_B by waves 0 / 1 and V_B by waves 2 / 3 (64 keys x 32 B = two pieces each) ----
  const int drow = 8 * wave + (lane >> 3), dpc = lane & 7, dc = dpc ^ ((drow >> 1) & 7);
  const bool kok = dc * 8 < a.d;
  unsigned koff[2], voff[2], toff[DB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    koff[i] = (unsigned)((attn32_keyperm(drow + 32 * i) * a.ldk + dc * 8) * 2);
    voff[i] = (unsigned)((attn32_keyperm(drow + 32 * i) * a.ldv + dc * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < DB; ++i) toff[i] = (unsigned)(((drow + 32 * i) * SP + dc * 8) * 2);
  const int brow = 32 * (wave & 1) + (lane >> 1), bc = lane & 1;
  const bool bok = KS > 4 && 64 + 8 * bc < a.d;
  const unsigned boff = (unsigned)((attn32_keyperm(brow) * (wave < 2 ? a.ldk : a.ldv) + 64 + 8 * bc) * 2);
  const char* kbase = (const char*)(a.K + ((int64_t)b * a.Skv) * a.ldk + hoff);
  const char* vbase = (const char*)(a.V + ((int64_t)b * a.Skv) * a.ldv + hoff);
  const char* tbase = (const char*)(a.KT + (((int64_t)b * a.H + hd) * DVP) * SP);
  const int64_t kstep = (int64_t)KVB * a.ldk * 2, vstep = (int64_t)KVB * a.ldv * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)((ATTN_LDS_AS char*)smem);
  auto dma_issue = [&](int t) __attribute__((always_inline)) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (t & 1) * STAGEB + wave * 1024);
    const bool last_ragged = ragged && t == nt - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool keyok = !last_ragged || t * KVB + attn32_keyperm(drow + 32 * i) < a.Skv;
      if (kok && keyok) {
        attn_glds16_s(kbase, koff[i], base + i * 4096);
        attn_glds16_s(vbase, voff[i], base + L::VOFF + i * 4096);
      }
    }
#pragma unroll
    for (int i = 0; i < DB; ++i)
      if (8 * (wave + 4 * i) < DVP) attn_glds16_s(tbase, toff[i], base + L::TOFF + i * 4096);
    if (KS > 4) {
      const bool keyok = !last_ragged || t * KVB + attn32_keyperm(brow) < a.Skv;
      if (bok && keyok) {   // (base carries wave * 1024: waves 0 / 1 -> K_B rows 0.. / 32.., waves 2 / 3 -> V_B = K_B + 2048)
        if (wave < 2) attn_glds16_s(kbase, boff, base + L::KBOFF);
        else attn_glds16_s(vbase, boff, base + L::KBOFF);
      }
    }
    kbase += kstep;
    vbase += vstep;
    tbase += KVB * 2;
  };
  __syncthreads();   // the zero image is complete before the first piece lands
  dma_issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // one 32-key block of K Q'^T - lse (V = false: C = cinit) or of V dO^T (V = true: C = 0); the V sub-tiles sit at fixed distances
  // from the K ones (V_A = K_A + 8192, V_B = K_B + 2048)
  auto block = [&](const char* base, const QF (&bf)[KS], int kb, auto v_tag) __attribute__((always_inline)) -> f32x16 {
    constexpr bool V = decltype(v_tag)::value;
    bf16x8 fr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      fr[ks] = *(const bf16x8*)(base + (ks < 4 ? (V ? L::VOFF : 0) + kb * 4096 : (V ? 2048 : 0) + kb * 1024) + kofs[ks]);
    f32x16 r;
    if (!V) {
      r = MFMA32(fr[0], bf[0].v, cinit);
    } else {
      f32x16 z;
#pragma unroll
      for (int i = 0; i < 16; ++i) z[i] = 0.f;
      r = MFMA32(fr[0], bf[0].v, z);
    }
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) r = MFMA32(fr[ks], bf[ks].v, r);
    return r;
  };
  auto tile = [&](auto slot_tag, auto tail_tag, int t) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    const char* base = smem + SLOT * STAGEB;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = block(base, qf, kb, std::false_type{});
      dp[kb] = block(base, dof, kb, std::true_type{});
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KVB + attn32_keyperm(32 * kb + 8 * (r >> 2) + 4 * h + (r & 3)) >= a.Skv) s[kb][r] = NEG_BIG;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = fast_exp2(s[kb][r]) * (dp[kb][r] - dl);   // the softmax scale is applied once, in the epilogue
      union { uint4 u; bf16x8 v; } p0, p1;
      p0.u = make_uint4(pack2bf(s[kb][0], s[kb][1]), pack2bf(s[kb][2], s[kb][3]), pack2bf(s[kb][4], s[kb][5]), pack2bf(s[kb][6], s[kb][7]));
      p1.u = make_uint4(pack2bf(s[kb][8], s[kb][9]), pack2bf(s[kb][10], s[kb][11]), pack2bf(s[kb][12], s[kb][13]), pack2bf(s[kb][14], s[kb][15]));
#pragma unroll
      for (int db = 0; db < DB; ++db) acc[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + tofs[2 * kb]), p0.v, acc[db]);
#pragma unroll
      for (int db = 0; db < DB; ++db) acc[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + tofs[2 * kb + 1]), p1.v, acc[db]);
    }
  };
  auto stage = [&](int t) __attribute__((always_inline)) {
    __syncthreads();
    if (t + 1 < nt) dma_issue(t + 1);
  };
  auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int nfull = ragged ? nt - 1 : nt;
  int t = 0;
  for (; t + 1 < nfull; t += 2) {
    stage(t);
    tile(S0{}, std::false_type{}, t);
    landed();
    stage(t + 1);
    tile(S1{}, std::false_type{}, t + 1);
    landed();
  }
  if (t < nfull) {
    stage(t);
    tile(S0{}, std::false_type{}, t);
    landed();
  }
  if (ragged) {
    stage(nt - 1);
    if ((nt - 1) & 1) tile(S1{}, std::true_type{}, nt - 1);
    else tile(S0{}, std::true_type{}, nt - 1);
  }
  // ---- epilogue: dQ[q][dd] = scale * dQ^T[dd][q] ----
  if (qrow < a.Sq) {
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int dd = 32 * db + 8 * i + 4 * h;
        if (dd < a.d) {
          uint2 pk;
          pk.x = pack2bf(acc[db][4 * i] * a.scale, acc[db][4 * i + 1] * a.scale);
          pk.y = pack2bf(acc[db][4 * i + 2] * a.scale, acc[db][4 * i + 3] * a.scale);
          *(uint2*)(a.out + ((int64_t)b * a.Sq + qrow) * a.ldout + hoff + dd) = pk;
        }
      }
  }
}

// =============================================================================================
// backward dK / dV on 32x32x16 MFMAs (round 3; head dims <= 80).  Per wave 32 keys (their K / V fragments live in registers as the B
// operands), per block 128; loop over 64-query tiles staged by LDS-DMA: Q and dO row-major with permuted rows (the A operands of
// S = Q K'^T and dP = dO V^T), Q^T and dO^T (A operands of dK^T += Q^T dS and dV^T += dO^T P, single b128 fragment reads), and the
// tile's lse / delta rows (64 floats each, permuted like the rows, fetched by the dword LDS-DMA).  The row vectors never touch the VALU:
// K enters negated (x = lse - S with C = lse straight from LDS, P = exp2(-x) through the source modifier), V enters negated
// (y = delta - dP with C = delta, dS = P * (-y)).  Stage layout: [Q_A 8K][dO_A 8K][Q^T DB x 4K][dO^T DB x 4K][Q_B 2K][dO_B 2K][lse][delta]
template <int KS, int DB>
struct AttnDkv32Lds {
  static constexpr int DOOFF = 8192, QTOFF = 16384, DOTOFF = QTOFF + DB * 4096, QBOFF = DOTOFF + DB * 4096, LOFF = QBOFF + (KS > 4 ? 4096 : 0),
                       STAGE = LOFF + 1024;
};
__device__ __forceinline__ void attn_glds4_s(const void* sbase, unsigned voff, unsigned lds_addr) {   // 4 bytes per lane
  const uint64_t pv = (uint64_t)(uintptr_t)sbase;
  const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pv), phi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pv >> 32));
  sbase = (const void*)(uintptr_t)(((uint64_t)phi << 32) | (uint64_t)plo);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr));
}
// WDV / WDK: which of the two gradients this launch produces.  The 80-wide heads (KS = 5, DB = 3) need ~330 registers for both at once;
// they run as two launches (dV: S, P, dV^T; dK: S, P, dP, dS, dK^T: 27 instead of 22 MFMAs per 32 x 32 block, at two waves per SIMD
// without scratch) -- the fused form spilled 94 registers.
template <int KS, int DB, int NSTAGE, bool WDV, bool WDK>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv32_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = AttnDkv32Lds<KS, DB>;
  constexpr int STAGEB = L::STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, n = lane & 31;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh - b * a.H;
  const int hoff = hd * a.d;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const int SPq = attn_spad(a.Sq), DVP = attn_dvpad(a.d);
  const float sc = a.scale * 1.4426950408889634f;

  for (int idx = tid; idx < NSTAGE * STAGEB / 16; idx += 256) *(uint4*)(smem + idx * 16) = make_uint4(0, 0, 0, 0);

  // this wave's keys: -sc K and -V fragments (B operands)
  union QF { bf16x8 v; uint4 u4; uint32_t u[4]; } kf[KS], vf[KS];
  const int krow = k0 + n;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks].u4 = make_uint4(0, 0, 0, 0);
    vf[ks].u4 = make_uint4(0, 0, 0, 0);
    const int c = 16 * ks + 8 * h;
    if (krow < a.Skv && c < a.d) {
      kf[ks].u4 = *(const uint4*)(a.K + ((int64_t)b * a.Skv + krow) * a.ldk + hoff + c);
      vf[ks].u4 = *(const uint4*)(a.V + ((int64_t)b * a.Skv + krow) * a.ldv + hoff + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kf[ks].u[e] = pack2bf(__uint_as_float(kf[ks].u[e] << 16) * -sc, __uint_as_float(kf[ks].u[e] & 0xffff0000u) * -sc);
      if (WDK) vf[ks].u[e] ^= 0x80008000u;   // -V (sign bits)
    }
  }
  f32x16 adv[WDV ? DB : 1], adk[WDK ? DB : 1];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      if (WDV) adv[db][r] = 0.f;
      if (WDK) adk[db][r] = 0.f;
    }

  int qofs[KS], tofs[4];
  {
    const int sw = (n >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qofs[ks] = ks < 4 ? n * 128 + (((2 * ks + h) ^ sw) * 16) : L::QBOFF + n * 32 + h * 16;
#pragma unroll
    for (int st = 0; st < 4; ++st) tofs[st] = L::QTOFF + n * 128 + (((2 * st + h) ^ sw) * 16);
  }

  const int nt = (a.Sq + KVB - 1) / KVB;
  const bool ragged = (a.Sq % KVB) != 0;
  // ---- LDS-DMA staging: Q_A / dO_A rows 8 (wave + 4 i) .. (permuted queries), Q^T / dO^T rows likewise, Q_B by waves 0 / 1 and dO_B by
  // waves 2 / 3, lse by wave 0 and delta by wave 1 (one dword piece each, lane l <- query keyperm(l)) ----
  const int drow = 8 * wave + (lane >> 3), dpc = lane & 7, dc = dpc ^ ((drow >> 1) & 7);
  const bool cok = dc * 8 < a.d;
  unsigned qoff[2], dooff[2], toff[DB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    qoff[i] = (unsigned)((attn32_keyperm(drow + 32 * i) * a.ldq + dc * 8) * 2);
    dooff[i] = (unsigned)((attn32_keyperm(drow + 32 * i) * a.lddo + dc * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < DB; ++i) toff[i] = (unsigned)(((drow + 32 * i) * SPq + dc * 8) * 2);
  const int brow = 32 * (wave & 1) + (lane >> 1), bc = lane & 1;
  const bool bok = KS > 4 && 64 + 8 * bc < a.d;
  const unsigned boff = (unsigned)((attn32_keyperm(brow) * (wave < 2 ? a.ldq : a.lddo) + 64 + 8 * bc) * 2);
  const unsigned loff = (unsigned)(attn32_keyperm(lane) * 4);
  const char* qbase = (const char*)(a.Q + ((int64_t)b * a.Sq) * a.ldq + hoff);
  const char* dobase = (const char*)(a.dO + ((int64_t)b * a.Sq) * a.lddo + hoff);
  const char* qtbase = (const char*)(a.QT + (((int64_t)b * a.H + hd) * DVP) * SPq);
  const char* dotbase = (const char*)(a.dOT + (((int64_t)b * a.H + hd) * DVP) * SPq);
  const char* lbase = (const char*)(a.lse + ((int64_t)b * a.H + hd) * a.Sq);
  const char* dbase = (const char*)(a.delta + ((int64_t)b * a.H + hd) * a.Sq);
  const int64_t qstep = (int64_t)KVB * a.ldq * 2, dostep = (int64_t)KVB * a.lddo * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)((ATTN_LDS_AS char*)smem);
  auto dma_issue = [&](int t) __attribute__((always_inline)) {
    const unsigned sbase0 = __builtin_amdgcn_readfirstlane(lds0 + (NSTAGE > 1 ? (t & 1) * STAGEB : 0));
    const unsigned base = __builtin_amdgcn_readfirstlane(sbase0 + wave * 1024);
    const bool last_ragged = ragged && t == nt - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool rowok = !last_ragged || t * KVB + attn32_keyperm(drow + 32 * i) < a.Sq;
      if (cok && rowok) {
        attn_glds16_s(qbase, qoff[i], base + i * 4096);
        if (WDK) attn_glds16_s(dobase, dooff[i], base + L::DOOFF + i * 4096);
      }
    }
#pragma unroll
    for (int i = 0; i < DB; ++i)
      if (8 * (wave + 4 * i) < DVP) {
        if (WDK) attn_glds16_s(qtbase, toff[i], base + L::QTOFF + i * 4096);
        if (WDV) attn_glds16_s(dotbase, toff[i], base + L::DOTOFF + i * 4096);
      }
    if (KS > 4) {
      const bool rowok = !last_ragged || t * KVB + attn32_keyperm(brow) < a.Sq;
      if (bok && rowok) {   // (base carries wave * 1024: waves 0 / 1 -> Q_B, waves 2 / 3 -> dO_B = Q_B + 2048)
        if (wave < 2) attn_glds16_s(qbase, boff, base + L::QBOFF);
        else if (WDK) attn_glds16_s(dobase, boff, base + L::QBOFF);
      }
    }
    if (wave < 2) {
      const bool rowok = !last_ragged || t * KVB + attn32_keyperm(lane) < a.Sq;
      if (rowok) {
        if (wave == 0) attn_glds4_s(lbase, loff, sbase0 + L::LOFF);
        else if (WDK) attn_glds4_s(dbase, loff, sbase0 + L::LOFF + 256);
      }
    }
    qbase += qstep;
    dobase += dostep;
    qtbase += KVB * 2;
    dotbase += KVB * 2;
    lbase += KVB * 4;
    dbase += KVB * 4;
  };

  auto tile = [&](const char* base, auto tail_tag, int t) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const float* sL = (const float*)(base + L::LOFF);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // x = lse - S, y = delta - dP: the row vectors are the C operands (rows 8 i + 4 h + j of the block <-> registers 4 i + j)
      f32x16 x, y;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 lv = *(const float4*)(sL + 32 * qb + 8 * i + 4 * h);
        x[4 * i] = lv.x; x[4 * i + 1] = lv.y; x[4 * i + 2] = lv.z; x[4 * i + 3] = lv.w;
        if (WDK) {
          const float4 dv = *(const float4*)(sL + 64 + 32 * qb + 8 * i + 4 * h);
          y[4 * i] = dv.x; y[4 * i + 1] = dv.y; y[4 * i + 2] = dv.z; y[4 * i + 3] = dv.w;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (ks < 4 ? qb * 4096 : qb * 1024) + qofs[ks];
        x = MFMA32(*(const bf16x8*)(base + off), kf[ks].v, x);
        if (WDK) y = MFMA32(*(const bf16x8*)(base + off + (ks < 4 ? L::DOOFF : 2048)), vf[ks].v, y);
      }
      if (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KVB + attn32_keyperm(32 * qb + 8 * (r >> 2) + 4 * h + (r & 3)) >= a.Sq) x[r] = 1.0e30f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        x[r] = fast_exp2(-x[r]);                 // P
        if (WDK) y[r] = x[r] * -y[r];            // dS (the softmax scale is applied once, to dK, in the epilogue)
      }
      union { uint4 u; bf16x8 v; } p0, p1, s0, s1;
      p0.u = make_uint4(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7]));
      p1.u = make_uint4(pack2bf(x[8], x[9]), pack2bf(x[10], x[11]), pack2bf(x[12], x[13]), pack2bf(x[14], x[15]));
      if (WDK) {
        s0.u = make_uint4(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]), pack2bf(y[4], y[5]), pack2bf(y[6], y[7]));
        s1.u = make_uint4(pack2bf(y[8], y[9]), pack2bf(y[10], y[11]), pack2bf(y[12], y[13]), pack2bf(y[14], y[15]));
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if (WDV) adv[db] = MFMA32(*(const bf16x8*)(base + (L::DOTOFF - L::QTOFF) + db * 4096 + tofs[2 * qb]), p0.v, adv[db]);
        if (WDK) adk[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + tofs[2 * qb]), s0.v, adk[db]);
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if (WDV) adv[db] = MFMA32(*(const bf16x8*)(base + (L::DOTOFF - L::QTOFF) + db * 4096 + tofs[2 * qb + 1]), p1.v, adv[db]);
        if (WDK) adk[db] = MFMA32(*(const bf16x8*)(base + db * 4096 + tofs[2 * qb + 1]), s1.v, adk[db]);
      }
    }
  };
  auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  __syncthreads();   // the zero image is complete before the first piece lands
  if (NSTAGE > 1) {
    dma_issue(0);
    landed();
    for (int t = 0; t < nt; ++t) {
      __syncthreads();   // stage t & 1 complete; every wave is done with the other stage
      if (t + 1 < nt) dma_issue(t + 1);
      if (ragged && t == nt - 1) tile(smem + (t & 1) * STAGEB, std::true_type{}, t);
      else tile(smem + (t & 1) * STAGEB, std::false_type{}, t);
      landed();
    }
  } else {
    for (int t = 0; t < nt; ++t) {
      if (t) __syncthreads();   // every wave is done with the previous tile
      dma_issue(t);
      landed();
      __syncthreads();
      if (ragged && t == nt - 1) tile(smem, std::true_type{}, t);
      else tile(smem, std::false_type{}, t);
    }
  }
  // ---- epilogue: dV[key][dd] = dV^T[dd][key], dK[key][dd] = scale * dK^T[dd][key] ----
  if (krow < a.Skv) {
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int dd = 32 * db + 8 * i + 4 * h;
        if (dd < a.d) {
          uint2 pk;
          if (WDK) {
            pk.x = pack2bf(adk[db][4 * i] * a.scale, adk[db][4 * i + 1] * a.scale);
            pk.y = pack2bf(adk[db][4 * i + 2] * a.scale, adk[db][4 * i + 3] * a.scale);
            *(uint2*)(a.dK + ((int64_t)b * a.Skv + krow) * a.lddk + hoff + dd) = pk;
          }
          if (WDV) {
            pk.x = pack2bf(adv[db][4 * i], adv[db][4 * i + 1]);
            pk.y = pack2bf(adv[db][4 * i + 2], adv[db][4 * i + 3]);
            *(uint2*)(a.dV + ((int64_t)b * a.Skv + krow) * a.lddv + hoff + dd) = pk;
          }
        }
      }
  }
}

template <typename KernelT>
int set_smem(KernelT k, int bytes) {
  FDMI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

// 32x32x16 forward (round 3) for head dims < 64 and 64 < d <= 80; returns 1 when it took the problem, 0 when the 16x16x32 family should,
// < 0 on error.  Measured (profiles/r3_attn_ab.txt): wins for d < 64 (d = 40: +13..16 %) and 64 < d <= 80; the 16x16x32 kernel keeps
// d = 64 (+3..11 %).  Switch 11 != 0 (register staging experiments): 16x16x32 as well.
static int launch_attn_fwd32(const AttnArgs& a, hipStream_t st) {
  // A/B switch 26: 1 = the 16x16x32 kernels always, 2 = this kernel for every problem in its head-dim range (the parity tests flip both)
  if (fdmi_tune_get(26) == 1 || fdmi_tune_get(11) != 0) return 0;
  if (fdmi_tune_get(26) != 2 && a.Skv <= 128) return 0;   // one or two key tiles (cross-attention): 41 vs 45 us at Skv = 77, 73 vs 83 at 120
  if (!(a.d < 64 || (a.d > 64 && a.d <= 80))) return 0;   // (d = 64 on this kernel, KS = 4 with the C-operand block: 3..7 % slower)
  // (KS, DB, ONES, MT): d <= 40 -> (3, 2, *, true); d = 48 -> (3, 2, false, false) (no spare k slot, no spare V^T row); d = 56 -> (4, 2, *,
  // true); d = 72 -> (5, 3, *, true); d = 80 -> (5, 3, false, false)
#define ATTN32_ALL(F)                                                                                                             \
  F(3, 2, true, true) F(3, 2, false, true) F(3, 2, false, false) F(4, 2, true, true) F(4, 2, false, true) F(5, 3, true, true)      \
  F(5, 3, false, true) F(5, 3, false, false)
  static bool once32 = false;
  if (!once32) {
#define ATTN32_SET(KS_, DB_, ON_, MT_) \
  if (set_smem(attn_fwd32_kernel<KS_, DB_, ON_, MT_>, 2 * Attn32Lds<KS_, DB_>::STAGE)) return -2;
    ATTN32_ALL(ATTN32_SET)
#undef ATTN32_SET
    once32 = true;
  }
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_shape(2, (long long)a.B * a.H, a.Sq, a.Skv, a.d);
  if (prof) fdmi_prof_begin(st, PROF_ATTN_FWD, 4.0 * a.B * a.H * (double)a.Sq * a.Skv * a.d, 4.0 * a.B * a.H * (double)a.d * ((double)a.Sq + a.Skv));
  dim3 g32(cdiv(a.Sq, 128), a.B * a.H);
  const bool ones32 = a.vt_ones && a.d < attn_dvpad(a.d);   // (the transposer's spare padded row dd = d)
  const int ks32 = a.d <= 48 ? 3 : (a.d < 64 ? 4 : 5), db32 = a.d < 64 ? 2 : 3;
  const bool mt32 = a.d < 16 * ks32;                         // spare k slot for the running maximum
#define ATTN32_GO(KS_, DB_, ON_, MT_)                                                                                             \
  if (ks32 == KS_ && db32 == DB_ && ones32 == ON_ && mt32 == MT_)                                                                 \
    FDMI_KLAUNCH(prof, (attn_fwd32_kernel<KS_, DB_, ON_, MT_>), g32, dim3(256), (2 * Attn32Lds<KS_, DB_>::STAGE), st, a);
  ATTN32_ALL(ATTN32_GO)
#undef ATTN32_GO
#undef ATTN32_ALL
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 1;
}

template <int DK, int DV, int NF>
int fwd_t(const AttnArgs& a, hipStream_t st) {
  // LDS-DMA staged variant for 64-wide tiles (developer knob 11 != 0 falls back to the register-staged kernel)
  constexpr bool CAN_DMA = DK == 64 && DV <= 64;
  const bool dma = CAN_DMA && fdmi_tune_get(11) == 0;
  constexpr int smem_reg = 2 * (RowTile<DK>::BYTES + TrTile<DV>::BYTES);
  constexpr int smem_dma = 2 * 2 * 64 * 128;
  // developer knob 10: extra (unused) LDS bytes per block, to lower the occupancy for latency-vs-throughput experiments
  const int smem = (dma ? smem_dma : smem_reg) + (fdmi_tune_get(10) > 0 ? fdmi_tune_get(10) : 0);
  static int once = -1;
  if (once != smem) {
    if (set_smem(attn_fwd_kernel<DK, DV, NF, false, false>, smem)) return -2;
    if (set_smem(attn_fwd_kernel<DK, DV, NF, true, false>, smem)) return -2;
    if constexpr (CAN_DMA) {
      if (set_smem(attn_fwd_kernel<DK, DV, NF, false, true>, smem)) return -2;
      if (set_smem(attn_fwd_kernel<DK, DV, NF, true, true>, smem)) return -2;
    }
    once = smem;
  }
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_shape(2, (long long)a.B * a.H, a.Sq, a.Skv, a.d);
  if (prof) fdmi_prof_begin(st, PROF_ATTN_FWD, 4.0 * a.B * a.H * (double)a.Sq * a.Skv * a.d, 4.0 * a.B * a.H * (double)a.d * ((double)a.Sq + a.Skv));
  const bool ones = a.vt_ones && a.d < DV;
  dim3 grid(cdiv(a.Sq, 64 * NF), a.B * a.H);
  if constexpr (CAN_DMA) {
    if (dma) {
      if (ones) FDMI_KLAUNCH(prof, (attn_fwd_kernel<DK, DV, NF, true, true>), grid, dim3(256), smem, st, a);
      else FDMI_KLAUNCH(prof, (attn_fwd_kernel<DK, DV, NF, false, true>), grid, dim3(256), smem, st, a);
    }
  }
  if (!dma) {
    if (ones) FDMI_KLAUNCH(prof, (attn_fwd_kernel<DK, DV, NF, true, false>), grid, dim3(256), smem, st, a);
    else FDMI_KLAUNCH(prof, (attn_fwd_kernel<DK, DV, NF, false, false>), grid, dim3(256), smem, st, a);
  }
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 0;
}
// 32x32x16 backward-dQ for head dims <= 80; returns 1 when it took the problem.  A/B switch 35 = 1: the 16x16x32 kernels.
static int launch_attn_bwd_dq32(const AttnArgs& a, hipStream_t st) {
  if (fdmi_tune_get(35) != 0 || a.d > 80) return 0;
#define DQ32_ALL(F) F(3, 2) F(4, 2) F(5, 3)
  static bool once = false;
  if (!once) {
#define DQ32_SET(KS_, DB_) if (set_smem(attn_bwd_dq32_kernel<KS_, DB_>, (2 * AttnDq32Lds<KS_, DB_>::STAGE))) return -2;
    DQ32_ALL(DQ32_SET)
#undef DQ32_SET
    once = true;
  }
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_shape(3, (long long)a.B * a.H, a.Sq, a.Skv, a.d);
  if (prof) fdmi_prof_begin(st, PROF_ATTN_DQ, 6.0 * a.B * a.H * (double)a.Sq * a.Skv * a.d, 2.0 * a.B * a.H * (double)a.d * (3.0 * a.Sq + 2.0 * a.Skv));
  dim3 grid(cdiv(a.Sq, 128), a.B * a.H);
  const int ks = a.d <= 48 ? 3 : (a.d <= 64 ? 4 : 5), db = a.d <= 64 ? 2 : 3;
#define DQ32_GO(KS_, DB_) \
  if (ks == KS_ && db == DB_) FDMI_KLAUNCH(prof, (attn_bwd_dq32_kernel<KS_, DB_>), grid, dim3(256), (2 * AttnDq32Lds<KS_, DB_>::STAGE), st, a);
  DQ32_ALL(DQ32_GO)
#undef DQ32_GO
#undef DQ32_ALL
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 1;
}
// 32x32x16 backward-dK/dV for head dims <= 80; returns 1 when it took the problem.
static int launch_attn_bwd_dkv32(const AttnArgs& a, hipStream_t st) {
  // A/B switch 36: 1 = the 16x16x32 kernel always, 2 = this kernel for every problem with d <= 80 (the parity tests flip both)
  if (fdmi_tune_get(36) == 1 || a.d > 80) return 0;
  if (fdmi_tune_get(36) != 2 && (int64_t)cdiv(a.Skv, 128) * a.B * a.H < 512)
    return 0;   // few keys (cross-attention): the 64-key blocks of the 16x16x32 kernel fill the chip better (Skv = 120, d = 72: 401 vs 459 us)
#define DKV32_ALL(F) F(3, 2, 2, true, true) F(4, 2, 2, true, true) F(5, 3, 1, true, false) F(5, 3, 1, false, true)
  static bool once = false;
  if (!once) {
#define DKV32_SET(KS_, DB_, NS_, DV_, DK_) \
  if (set_smem(attn_bwd_dkv32_kernel<KS_, DB_, NS_, DV_, DK_>, (NS_ * AttnDkv32Lds<KS_, DB_>::STAGE))) return -2;
    DKV32_ALL(DKV32_SET)
#undef DKV32_SET
    once = true;
  }
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_shape(4, (long long)a.B * a.H, a.Sq, a.Skv, a.d);
  if (prof) fdmi_prof_begin(st, PROF_ATTN_DKV, 8.0 * a.B * a.H * (double)a.Sq * a.Skv * a.d, 2.0 * a.B * a.H * (double)a.d * (2.0 * a.Sq + 4.0 * a.Skv));
  dim3 grid(cdiv(a.Skv, 128), a.B * a.H);
  const int ks = a.d <= 48 ? 3 : (a.d <= 64 ? 4 : 5), db = a.d <= 64 ? 2 : 3;
#define DKV32_GO(KS_, DB_, NS_, DV_, DK_) \
  if (ks == KS_ && db == DB_)             \
    FDMI_KLAUNCH(prof, (attn_bwd_dkv32_kernel<KS_, DB_, NS_, DV_, DK_>), grid, dim3(256), (NS_ * AttnDkv32Lds<KS_, DB_>::STAGE), st, a);
  DKV32_ALL(DKV32_GO)
#undef DKV32_GO
#undef DKV32_ALL
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 1;
}
template <int DK, int DV, int NF>
int dq_t(const AttnArgs& a, hipStream_t st) {
  constexpr int smem = 2 * RowTile<DK>::BYTES + TrTile<DV>::BYTES;
  static bool once = false;
  if (!once) { if (set_smem(attn_bwd_dq_kernel<DK, DV, NF>, smem)) return -2; once = true; }
  dim3 grid(cdiv(a.Sq, 64 * NF), a.B * a.H);
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_shape(3, (long long)a.B * a.H, a.Sq, a.Skv, a.d);
  if (prof) fdmi_prof_begin(st, PROF_ATTN_DQ, 6.0 * a.B * a.H * (double)a.Sq * a.Skv * a.d, 2.0 * a.B * a.H * (double)a.d * (3.0 * a.Sq + 2.0 * a.Skv));
  FDMI_KLAUNCH(prof, (attn_bwd_dq_kernel<DK, DV, NF>), grid, dim3(256), smem, st, a);
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 0;
}
template <int DK, int DV, int NF>
int dkv_t(const AttnArgs& a, hipStream_t st) {
  constexpr int smem = 2 * RowTile<DK>::BYTES + 2 * TrTile<DV>::BYTES + 512;
  static bool once = false;
  if (!once) { if (set_smem(attn_bwd_dkv_kernel<DK, DV, NF>, smem)) return -2; once = true; }
  dim3 grid(cdiv(a.Skv, 64 * NF), a.B * a.H);
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_shape(4, (long long)a.B * a.H, a.Sq, a.Skv, a.d);
  if (prof) fdmi_prof_begin(st, PROF_ATTN_DKV, 8.0 * a.B * a.H * (double)a.Sq * a.Skv * a.d, 2.0 * a.B * a.H * (double)a.d * (2.0 * a.Sq + 4.0 * a.Skv));
  FDMI_KLAUNCH(prof, (attn_bwd_dkv_kernel<DK, DV, NF>), grid, dim3(256), smem, st, a);
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// head-dim dispatch: (d -> DK = ceil32, DV = ceil16)
#define ATTN_DISPATCH(FN, NF_SMALL, NF_BIG)                                        \
  ATTN_DISPATCH4(FN, NF_SMALL, NF_SMALL, NF_BIG)
#define ATTN_DISPATCH4(FN, NF_TINY, NF_SMALL, NF_BIG)                              \
  const int DKp = (a.d + 31) & ~31, DVp = attn_dvpad(a.d);                          \
  if (DKp == 32 && DVp == 16) return FN<32, 16, NF_TINY>(a, st);                   \
  if (DKp == 32 && DVp == 32) return FN<32, 32, NF_TINY>(a, st);                   \
  if (DKp == 64 && DVp == 48) return FN<64, 48, NF_TINY>(a, st);                   \
  if (DKp == 64 && DVp == 64) return FN<64, 64, NF_TINY>(a, st);                   \
  if (DKp == 96 && DVp == 80) return FN<96, 80, NF_SMALL>(a, st);                  \
  if (DKp == 96 && DVp == 96) return FN<96, 96, NF_SMALL>(a, st);                  \
  if (DKp == 128 && DVp == 128) return FN<128, 128, NF_BIG>(a, st);                \
  if (DKp == 160 && DVp == 160) return FN<160, 160, NF_BIG>(a, st);                \
  FDMI_CHECK(false, "attention: unsupported head dim " + std::to_string(a.d));

static int check_attn(const AttnArgs& a) {
  FDMI_CHECK(a.d % 8 == 0 && a.d <= 160, "attention: head dim must be a multiple of 8, <= 160");
  FDMI_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0, "attention: leading dims must be multiples of 8");
  FDMI_CHECK(a.Sq > 0 && a.Skv > 0, "attention: empty sequence");
  return 0;
}

int launch_attn_fwd(const AttnArgs& a, hipStream_t st) {
  if (check_attn(a)) return -1;
  if (const int rc = launch_attn_fwd32(a, st)) return rc < 0 ? rc : 0;
  // query fragments per wave: 2 for d <= 96 (the 96-wide instantiations -- d = 72 PixArt, d = 80 SD1.5 level 1 -- take 204 VGPRs,
  // still two waves per SIMD; every K / V^T fragment read from LDS then feeds twice the MFMAs: +17..24 % at d = 72 / 80,
  // profiles/r3_attn_ab.txt; A/B switch 27 = 1 restores one fragment), 1 for larger heads
  if (fdmi_tune_get(27) == 0) { ATTN_DISPATCH4(fwd_t, 2, 2, 1) }
  ATTN_DISPATCH4(fwd_t, 2, 1, 1)
}
int launch_attn_bwd_dq(const AttnArgs& a, hipStream_t st) {
  if (check_attn(a)) return -1;
  if (const int rc = launch_attn_bwd_dq32(a, st)) return rc < 0 ? rc : 0;
  if (fdmi_tune_get(34) == 0) { ATTN_DISPATCH4(dq_t, 2, 1, 1) }   // 96-wide heads: one fragment, two blocks per CU (switch 34 = 1: two, one)
  ATTN_DISPATCH(dq_t, 2, 1)
}
int launch_attn_bwd_dkv(const AttnArgs& a, hipStream_t st) {
  if (check_attn(a)) return -1;
  if (const int rc = launch_attn_bwd_dkv32(a, st)) return rc < 0 ? rc : 0;
  if (fdmi_tune_get(34) == 0) { ATTN_DISPATCH4(dkv_t, 2, 1, 1) }
  ATTN_DISPATCH(dkv_t, 2, 1)
}

int launch_transpose_heads(const bf16_t* X, int64_t ld, bf16_t* XT, int B, int H, int S, int d,
                           hipStream_t st, int ones_row) {
  FDMI_CHECK(d % 8 == 0 && ld % 8 == 0, "transpose_heads: d, ld must be multiples of 8");
  dim3 grid(attn_spad(S) / 64, cdiv(attn_dvpad(d), 64), B * H);
  hipLaunchKernelGGL(transpose_heads_kernel, grid, dim3(256), 0, st, X, ld, XT, B, H, S, d, ones_row);
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_attn_delta(const bf16_t* O, int64_t ldo, const bf16_t* dO, int64_t lddo, float* delta, int B,
                      int H, int S, int d, hipStream_t st) {
  const int64_t groups = (int64_t)B * S * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256), 0, st, O, ldo,
                     dO, lddo, delta, B, H, S, d);
  FDMI_HIP(hipGetLastError());
  return 0;
}
