// Large-tile bf16 MFMA GEMM / implicit-GEMM conv for gfx950: 256 x {128,160} x 64 tiles, 512 threads
// (8 waves as 4(M) x 2(N), wave tile 64 x BN/2), 3-stage LDS ring filled by LDS-DMA
// (global_load_lds 16 B/lane) with a COUNTED s_waitcnt vmcnt(N) across a raw s_barrier, so the next
// tile's loads stay in flight over the barrier (prefetch distance 2 tiles, one barrier per K tile).
//
// Why this shape: with 128x128 tiles the kernel needs ~38 TB/s of L2->LDS traffic at the MFMA peak
// (more than the 8 XCD L2s deliver); 256x160 needs ~25 TB/s.  BN = 160 divides every channel count of
// the SD/SDXL UNets (320, 640, 1280, 2560, 5120, 10240), BN = 128 covers GEGLU pairs and DiT widths.
// The im2col gather keeps one 64-bit source pointer per staged row that is re-derived only when the
// K loop crosses a filter tap (Cin % 64 == 0), so the steady-state loader is an add per row.
#include "gemm.h"

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
      if (n + r < a.N) v[r] += bf2f(rv[r]);
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
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += bf2f(rv[r]);
  }
}
__device__ __forceinline__ void epi_store8(const GemmArgs& a, int act, int64_t m, int nout, float v[8]) {
  if (a.residual) {
    const u16x8 t = *(const u16x8*)(a.residual + m * a.ldr + nout);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += bf2f(t[r]);
  }
  if (act == ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = silu_f(v[r]);
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

// LDS-DMA issued from inline asm: hipcc then keeps no scoreboard entry for it, so the ONLY waits on
// these loads are the counted ones placed by hand below (with the builtin, the waitcnt pass drained
// the ring with vmcnt(0) at every loop back-edge of the persistent loop).  M0 (the LDS destination
// base) is written and restored inside the statement; lds_addr is wave-uniform.
__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(lds_addr)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, int MODE>
__global__ __launch_bounds__(512) void gemm3_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int AR = 4;                       // A chunks per thread per tile (256 rows * 8 / 512)
  constexpr int WR = (BN * 8 + 511) / 512;    // W chunks per thread per tile (2 or 3)
  constexpr int WFULL = (BN * 8) / 512;       // chunks every thread issues
  constexpr bool WODD = WR != WFULL;          // BN = 160: waves 0-3 issue one more
  constexpr int MF = 4, NF = BN / 32;         // wave tile 64 x BN/2
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, j = lane & 15;

  // ---- persistent work loop: item = (tile, k-split); block b takes items b, b+G, b+2G, ... ----
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int sk = a.splitk > 1 ? a.splitk : 1;
  const int Wtot = tilesM * tilesN * sk;
  const int G = gridDim.x;
  const int ktiles = a.K >> 6, kper = (ktiles + sk - 1) / sk;
  auto remap = [&](int v) {  // XCD-aware (block b runs on XCD b % 8; G % 8 == 0 or G == Wtot)
    const int xcd = v & 7, q = Wtot >> 3, r = Wtot & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  };
  struct Item { int m0, n0, kbeg, nk, z; };
  auto item_of = [&](int v) {
    const int w = remap(v);
    const int tile = w / sk, z = w - tile * sk;
    const int tn = tile % tilesN, tm = tile / tilesN;
    Item it;
    it.m0 = tm * BM; it.n0 = tn * BN; it.z = z;
    it.kbeg = z * kper * 64;
    const int kend = min(a.K, it.kbeg + kper * 64);
    it.nk = max(0, (kend - it.kbeg) >> 6);
    return it;
  };

  // ---- loader (issue side) state ----
  const int p = tid & 7, lr = tid >> 3;      // slot, row within a 64-row group
  const int c8 = (p ^ ((lr >> 1) & 7)) * 8;  // logical k offset of this thread's chunk inside a tile
  const bf16_t* zero = (const bf16_t*)g_zero16b;
  const bf16_t* ap[AR];
  int ainc[AR];
  int aby[AR], abx[AR], apix[AR];
  bool aval[AR];
  int ky = 0, kx = 0, cc = 0;
  const bf16_t* wp[WR];
  int winc[WR];
  auto retap = [&]() {  // conv: derive the per-row source pointer for tap (ky, kx), channel offset cc
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      bool ok = aval[i];
      int sy = 0, sx = 0;
      if (a.dgrad) {
        const int ty = aby[i] - ky, tx = abx[i] - kx;
        ok = ok && ty >= 0 && tx >= 0;
        if (a.stride == 2) {
          ok = ok && (((ty | tx) & 1) == 0);
          sy = ty >> 1;
          sx = tx >> 1;
        } else {
          sy = ty;
          sx = tx;
        }
        ok = ok && sy < a.Hin && sx < a.Win;
      } else {
        const int iy = aby[i] + ky, ix = abx[i] + kx;
        ok = ok && iy >= 0 && ix >= 0 && iy < (a.Hin << a.ups) && ix < (a.Win << a.ups);
        sy = iy >> a.ups;
        sx = ix >> a.ups;
      }
      ap[i] = ok ? a.A + ((int64_t)(apix[i] + sy * a.Win + sx) * a.Cin + cc + c8) : zero;
      ainc[i] = ok ? 64 : 0;
    }
  };
  auto setup_issue = [&](const Item& it) {
    if (MODE == GEMM_CONV) {
      const int tap = it.kbeg / a.Cin;
      cc = it.kbeg - tap * a.Cin;
      ky = tap / a.KW;
      kx = tap - ky * a.KW;
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int m = it.m0 + lr + 64 * i;
      aval[i] = m < a.M;
      if (MODE == GEMM_ROW) {
        ap[i] = aval[i] ? a.A + (int64_t)m * a.lda + it.kbeg + c8 : zero;
        ainc[i] = aval[i] ? 64 : 0;
        aby[i] = abx[i] = apix[i] = 0;
      } else {
        const int hw = a.Hout * a.Wout;
        const int mm = aval[i] ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        apix[i] = b * a.Hin * a.Win;
        aby[i] = a.dgrad ? (oy + a.pad) : (oy * a.stride - a.pad);
        abx[i] = a.dgrad ? (ox + a.pad) : (ox * a.stride - a.pad);
      }
    }
    if (MODE == GEMM_CONV) retap();
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int n = it.n0 + lr + 64 * i;
      const bool ok = n < a.N && (lr + 64 * i) < BN;
      wp[i] = ok ? a.W + (int64_t)n * a.ldw + it.kbeg + c8 : zero;
      winc[i] = ok ? 64 : 0;
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);
  auto issue = [&](int slot) {
    const unsigned sa = __builtin_amdgcn_readfirstlane(lds0 + slot * STAGE + wave * 1024);
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      glds16(ap[i], sa + 512 * 16 * i);
      ap[i] += ainc[i];
    }
#pragma unroll
    for (int i = 0; i < WFULL; ++i) {
      glds16(wp[i], sa + BM * 128 + 512 * 16 * i);
      wp[i] += winc[i];
    }
    if (WODD) {
      if (wave < 4) {
        glds16(wp[WR - 1], sa + BM * 128 + 512 * 16 * (WR - 1));
        wp[WR - 1] += winc[WR - 1];
      }
    }
    if (MODE == GEMM_CONV) {
      cc += 64;
      if (cc >= a.Cin) {  // next tile starts a new filter tap (uniform: Cin % 64 == 0)
        cc = 0;
        if (++kx == a.KW) {
          kx = 0;
          ++ky;
        }
        retap();
      }
    }
  };
  // issue-side cursor over the flattened (item, k-tile) sequence
  int iv = blockIdx.x, ikt = 0, ink = 0, islot = 0;
  bool ihave = false;
  auto issue_next = [&]() -> bool {
    while (!ihave || ikt == ink) {
      if (ihave) iv += G;
      if (iv >= Wtot) return false;
      const Item it = item_of(iv);
      ihave = true;
      ikt = 0;
      ink = it.nk;
      if (ink > 0) setup_issue(it);
    }
    issue(islot);
    islot = islot == 2 ? 0 : islot + 1;
    ++ikt;
    return true;
  };

  f32x4 acc[NF][MF];
  // fragments of BOTH k-steps are requested right after the barrier, the next tile's LDS-DMA is issued
  // while they are in flight, then the 2 x NF x MF MFMAs run (a wave issues in order: anything placed
  // after the MFMAs would only start once they have all been issued)
  bf16x8 af[2][MF], wf[2][NF];
  auto load_frags = [&](int slot) {
    const char* sa = smem + slot * STAGE;
    const char* sw = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pc = (ks * 4 + g) ^ (j >> 1);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) af[ks][mf] = *(const bf16x8*)(sa + ((wm * 64 + mf * 16 + j) * 8 + pc) * 16);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) wf[ks][nf] = *(const bf16x8*)(sw + ((wn * (BN / 2) + nf * 16 + j) * 8 + pc) * 16);
    }
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nf], af[ks][mf], acc[nf][mf], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  auto epilogue = [&](const Item& it) {
    const int m0 = it.m0, n0 = it.n0;
    if (a.accum_atomic) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t m = m0 + wm * 64 + mf * 16 + j;
          const int n = n0 + wn * (BN / 2) + nf * 16 + g * 4;
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
    if (a.splitk > 1) {  // raw partial sums -> this split's fp32 slab (plain stores, deterministic)
      float* slab = a.ws + (int64_t)it.z * a.M * a.N;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t m = m0 + wm * 64 + mf * 16 + j;
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
      if constexpr (NF == 4) {
        const int Nout = a.N >> 1;
        if (wide) {
          // pair value fragments (0,2) and gate fragments (1,3): even lane groups get block q=0, odd q=1
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              SWAP16(acc[0][mf][r], acc[2][mf][r]);
              SWAP16(acc[1][mf][r], acc[3][mf][r]);
            }
            const int64_t m = m0 + wm * 64 + mf * 16 + j;
            const int q = g & 1;
            const int n = n0 + wn * 64 + q * 32 + (g >> 1) * 8;  // value cols n..n+7, gate cols n+16..n+23
            if (m < a.M && n < a.N) {
              float val[8], gate[8], o[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                val[r] = acc[0][mf][r]; val[4 + r] = acc[2][mf][r];
                gate[r] = acc[1][mf][r]; gate[4 + r] = acc[3][mf][r];
              }
              epi_terms8(a, m, n, val);
              epi_terms8(a, m, n + 16, gate);
              if (a.preact) {
                uint4 pk;
                pk.x = pack2bf(val[0], val[1]); pk.y = pack2bf(val[2], val[3]); pk.z = pack2bf(val[4], val[5]); pk.w = pack2bf(val[6], val[7]);
                *(uint4*)(a.preact + m * a.ldp + n) = pk;
                pk.x = pack2bf(gate[0], gate[1]); pk.y = pack2bf(gate[2], gate[3]); pk.z = pack2bf(gate[4], gate[5]); pk.w = pack2bf(gate[6], gate[7]);
                *(uint4*)(a.preact + m * a.ldp + n + 16) = pk;
              }
#pragma unroll
              for (int r = 0; r < 8; ++r) o[r] = val[r] * gelu_f(gate[r]);
              epi_store8(a, ACT_NONE, m, ((n0 + wn * 64) >> 1) + q * 16 + (g >> 1) * 8, o);
            }
          }
          return;
        }
#pragma unroll
        for (int q = 0; q < NF / 2; ++q)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int64_t m = m0 + wm * 64 + mf * 16 + j;
            const int n = n0 + wn * (BN / 2) + q * 32 + g * 4;
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
              epi_store3(a, ACT_NONE, m, ((n0 + wn * (BN / 2)) >> 1) + q * 16 + g * 4, Nout, o);
            }
          }
      }
      return;
    }
    if (wide) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = m0 + wm * 64 + mf * 16 + j;
#pragma unroll
        for (int pr = 0; pr < NF / 2; ++pr) {
          const int nf = 2 * pr;
#pragma unroll
          for (int r = 0; r < 4; ++r) SWAP16(acc[nf][mf][r], acc[nf + 1][mf][r]);
          const int n = n0 + wn * (BN / 2) + (nf + (g & 1)) * 16 + (g >> 1) * 8;
          if (m < a.M && n < a.N) {
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
          const int n = n0 + wn * (BN / 2) + (NF - 1) * 16 + g * 4;
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
        const int64_t m = m0 + wm * 64 + mf * 16 + j;
        const int n = n0 + wn * (BN / 2) + nf * 16 + g * 4;
        if (m < a.M && n < a.N) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[nf][mf][r];
          epi_terms3(a, m, n, v);
          epi_store3(a, a.act, m, n, a.N, v);
        }
      }
  };

  // ---- flattened 3-stage ring across items: counted vmcnt, one raw barrier per K tile ----
  int inflight = 0;
  if (issue_next()) ++inflight;
  if (issue_next()) ++inflight;
  int cslot = 0;
  for (int cv = blockIdx.x; cv < Wtot; cv += G) {
    const Item it = item_of(cv);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < it.nk; ++t) {
      // the epilogue of the previous item left ordinary loads/stores on the VM counter: drain once
      if (inflight >= 2 && t > 0) {
        if (WODD) {
          if (wave < 4) wait_vmcnt<AR + WR>(); else wait_vmcnt<AR + WFULL>();
        } else {
          wait_vmcnt<AR + WFULL>();
        }
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (issue_next()) ++inflight;   // (DMA issue first: the asm's memory clobber would drain pending ds_reads)
      load_frags(cslot);
      mma();
      cslot = cslot == 2 ? 0 : cslot + 1;
      --inflight;
    }
    epilogue(it);
    // make the waitcnt pass see an empty VM scoreboard at the back-edge: otherwise it protects the
    // epilogue's pending loads/stores with a vmcnt(0) inside every K iteration (draining the ring)
    __builtin_amdgcn_s_waitcnt(0);
  }
}

template <int BN, int MODE>
int launch3_t(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int smem = 3 * (256 + BN) * 128;
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)gemm3_kernel<BN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int items = cdiv(a.M, 256) * cdiv(a.N, BN) * (a.splitk > 1 ? a.splitk : 1);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    FDMI_HIP(hipGetDevice(&dev));
    FDMI_HIP(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount & ~7) : 256;
    if (ncu < 8) ncu = 8;
  }
  dim3 grid(items < ncu ? items : ncu, 1, 1);   // persistent: one 8-wave block per CU
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_begin(stream, PROF_GEMM3 + MODE * 2 + (BN == 160 ? 0 : 1), gemm_flops(a));
  hipLaunchKernelGGL((gemm3_kernel<BN, MODE>), grid, dim3(512), smem, stream, a);
  if (prof) fdmi_prof_end(stream);
  FDMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool gemm3_eligible(const GemmArgs& a) {
  if ((a.K & 63) != 0 || a.M < 256) return false;
  if (a.mode == GEMM_CONV && (a.Cin & 63) != 0) return false;
  if (a.act == ACT_GEGLU && (a.N % 128) != 0) return false;
  return a.N >= 128;
}
int gemm3_pick_bn(const GemmArgs& a) {
  if (a.act == ACT_GEGLU) return 128;
  if (a.N % 160 == 0) return 160;
  if (a.N % 128 == 0) return 128;
  return ((double)cdiv(a.N, 160) * 160 / a.N <= (double)cdiv(a.N, 128) * 128 / a.N) ? 160 : 128;
}
int launch_gemm3(const GemmArgs& a, int BN, hipStream_t stream) {
  if (a.mode == GEMM_ROW) return BN == 160 ? launch3_t<160, GEMM_ROW>(a, stream) : launch3_t<128, GEMM_ROW>(a, stream);
  return BN == 160 ? launch3_t<160, GEMM_CONV>(a, stream) : launch3_t<128, GEMM_CONV>(a, stream);
}
