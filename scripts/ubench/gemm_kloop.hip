// K-loop structure micro-benchmark (dev tool, round 6): out[M][N] = A[M][K] W[N][K]^T, bf16 in, fp32 accumulate,
// bf16 out, on the LDS-DMA ring of csrc/gemm4.hip generalised over the wave geometry --
//   V8: 8 waves (4 x 2), two per SIMD, wave tile 64 x BN/2      = the product kernel's structure (baseline of every comparison)
//   V4: 4 waves (2 x 2), ONE per SIMD, wave tile 128 x BN/2     (512 registers per lane: accumulators of a 128 x 192 tile fit)
// and over experiment switches (the `x` argument, bit field):
//   1  pieces of the next tile spread over every other MFMA group (instead of the first NP groups)
//   2  no s_setprio around the MFMA groups
//   4  A fragments of BOTH k-steps loaded at the tile head (V4 has the registers)
//   8  W ring 7 deep instead of 3 (V4 only)
//  16  MFMAs from inline asm with a register-CLASS constraint per accumulator fragment ("+a": the first 64 fragments = the 256
//      AGPRs, "+v" the rest): hipcc itself shuttles a > 256-register accumulator set between the two files inside the K loop
//      (200 v_accvgpr_write + 196 v_accvgpr_read + s_nop 5 per K tile in the V4 kernels: the disassembly)
// The epilogue is a plain per-fragment bf16 store (not what is being measured: use long K).  Results are checked on sampled
// elements against a host fp32 dot product of the bf16 inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_kloop.hip -o gemm_kloop && ./gemm_kloop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_addr));
}
__device__ __forceinline__ void glds16_nt(const void* gptr, unsigned lds_addr) {   // the same piece with the non-temporal policy (aux = 2)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_addr));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

struct Args { const bf16_t* A; const bf16_t* W; bf16_t* C; int M, N, K; int x; };

template <int NWM, int NWN, int BN, int X>
__global__ __launch_bounds__(NWM * NWN * 64) void kloop(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, NT = NWM * NWN * 64, NWAVE = NWM * NWN;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int RPP = NT / 8;                                  // rows covered by one piece index (8 threads per 128-byte row)
  constexpr int AR = BM / RPP, WR = BN / RPP, NP = AR + WR;    // pieces per thread per tile
  constexpr int MF = BM / NWM / 16, NF = BN / NWN / 16, NQ = 2 * NF;
  constexpr bool SPREAD = X & 1, NOPRIO = X & 2, AHEAD = X & 4, ASMM = X & 16;
  constexpr bool A3 = X & 32, NOA = X & 64, NOW = X & 128, ANT = X & 256, WNT = X & 512;   // 256 / 512: A / W pieces non-temporal   // timing ablations (WRONG results): A pieces on every third K tile only / never / W pieces never
  constexpr int WD = (X & 8) ? 7 : 3;                          // W ring look-ahead (ring size WD + 1)
  static_assert(BN % RPP == 0 && BM % RPP == 0, "pieces");
  static_assert(SPREAD ? (NP * 2 <= NQ) : (NP <= NQ), "one piece per MFMA group");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int g = lane >> 4, j = lane & 15;
  const int tilesN = a.N / BN, tilesM = a.M / BM, Wtot = tilesM * tilesN, G = gridDim.x, ktiles = a.K >> 6;
  auto remap = [&](int v) {
    const int xcd = v & 7, q = Wtot >> 3, r = Wtot & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  };
  const int p = tid & 7, lr = tid >> 3;
  const int c8 = (p ^ ((lr >> 1) & 7)) * 8;
  const bf16_t* abase = a.A; const bf16_t* wbase = a.W;
  const int64_t astep = (int64_t)RPP * a.K, wstep = (int64_t)RPP * a.K;
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);
  int iv = blockIdx.x, ikt = 0, islot = 0;
  bool idone = iv >= Wtot;
  auto setup = [&](int v) {
    const int w = remap(v), tn = w % tilesN, tm = w / tilesN;
    abase = a.A + (int64_t)(tm * BM + lr) * a.K + c8;
    wbase = a.W + (int64_t)(tn * BN + lr) * a.K + c8;
  };
  if (!idone) setup(iv);
  auto slot_base = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane(lds0 + islot * STAGE + wave * 1024); };
  auto piece = [&](int i, unsigned sa) {
    if (i < AR) { if (ANT) glds16_nt(abase + i * astep, sa + NWAVE * 1024 * i); else glds16(abase + i * astep, sa + NWAVE * 1024 * i); }
    else if (WNT) glds16_nt(wbase + (i - AR) * wstep, sa + BM * 128 + NWAVE * 1024 * (i - AR));
    else glds16(wbase + (i - AR) * wstep, sa + BM * 128 + NWAVE * 1024 * (i - AR));
  };
  auto finish = [&]() {   // advance the issue cursor by one K tile (next item when the current one is through)
    abase += 64; wbase += 64; islot ^= 1;
    if (++ikt == ktiles) {
      ikt = 0; iv += G;
      if (iv >= Wtot) { idone = true; abase = a.A; wbase = a.W; }   // parked: harmless re-reads of the first rows
      else setup(iv);
    }
  };
  f32x4 acc[NF][MF];
  bf16x8 af[2][MF], wq[WD + 1];
  auto lds_a = [&](int ks, int mf, int slot) -> bf16x8 {
    const int pc = (ks * 4 + g) ^ (j >> 1);
    return *(const bf16x8*)(smem + slot * STAGE + ((wm * (BM / NWM) + mf * 16 + j) * 8 + pc) * 16);
  };
  auto lds_w = [&](int q, int slot) -> bf16x8 {
    const int ks = q / NF, nf = q - ks * NF;
    const int pc = (ks * 4 + g) ^ (j >> 1);
    return *(const bf16x8*)(smem + slot * STAGE + BM * 128 + ((wn * (BN / NWN) + nf * 16 + j) * 8 + pc) * 16);
  };
  if (!idone) {
    const unsigned sa = slot_base();
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, sa);
    finish();
  }
  int cslot = 0;
  for (int cv = blockIdx.x; cv < Wtot; cv += G) {
    const int w = remap(cv), tn = w % tilesN, tm = w / tilesN;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < ktiles; ++t) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) af[0][mf] = lds_a(0, mf, cslot);
      if (AHEAD) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) af[1][mf] = lds_a(1, mf, cslot);
      }
#pragma unroll
      for (int q = 0; q < WD; ++q) wq[q] = lds_w(q, cslot);
      const unsigned sa = slot_base();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ks = q / NF, nf = q % NF;
        if (q + WD < NQ) wq[(q + WD) % (WD + 1)] = lds_w(q + WD, cslot);
        if (!AHEAD && q == (NF >= 6 ? 3 : 1)) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) af[1][mf] = lds_a(1, mf, cslot);
        }
        if (!NOPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          if (!ASMM) acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q % (WD + 1)], af[ks][mf], acc[nf][mf], 0, 0, 0);
          else if (nf * MF + mf < 64) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nf][mf]) : "v"(wq[q % (WD + 1)]), "v"(af[ks][mf]));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nf][mf]) : "v"(wq[q % (WD + 1)]), "v"(af[ks][mf]));
        }
        if (!NOPRIO) __builtin_amdgcn_s_setprio(0);
        const int pi = SPREAD ? ((q & 1) ? -1 : q / 2) : q;
        if (pi >= 0 && pi < NP && !(pi < AR && ((A3 && (t % 3) != 0) || NOA)) && !(pi >= AR && NOW)) {
          __builtin_amdgcn_sched_barrier(0);
          piece(pi, sa);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      finish();
      cslot ^= 1;
    }
    if (ASMM) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // (the asm MFMAs' results: hipcc pads nothing for them)
    // plain epilogue: lane (g, j) holds row j, columns g*4 .. g*4+3 of each 16 x 16 fragment
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = tm * BM + wm * (BM / NWM) + mf * 16 + j;
        const int n = tn * BN + wn * (BN / NWN) + nf * 16 + g * 4;
        uint2 pk;
        pk.x = pack2bf(acc[nf][mf][0], acc[nf][mf][1]);
        pk.y = pack2bf(acc[nf][mf][2], acc[nf][mf][3]);
        *(uint2*)(a.C + m * a.N + n) = pk;
      }
  }
  wait_vmcnt<0>();
}


// ---- round 6, second experiment: the same 8-wave 256 x 320 block fed 32 deep through an NSLOT-slot ring with
// COUNTED waits -- per K tile (40 MFMAs per wave) `s_waitcnt vmcnt((NSLOT - 2) * 5)` + one barrier, no full drain except at an item's
// first tile (the epilogue's stores share vmcnt and may complete out of order with the loads) -- against the product's 2-slot 64-deep
// ring (one vmcnt(0) + barrier per 80 MFMAs, look-ahead = one tile's MFMA time).  36 KB per slot: 4 slots = 144 KB (+ 4 KB dump area:
// the 576 rows of a tile are 4.5 pieces of 128 rows, waves 4 - 7 aim their fifth piece at it so that every wave counts 5 loads).
template <int NSLOT>
__global__ __launch_bounds__(512) void kloop32(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 320, STAGE = (BM + BN) * 64, P = 5, D = NSLOT - 1;
  constexpr int MF = 4, NF = 10;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  const int tilesN = a.N / BN, tilesM = a.M / BM, Wtot = tilesM * tilesN, G = gridDim.x, ktiles = a.K >> 5;
  auto remap = [&](int v) {
    const int xcd = v & 7, q = Wtot >> 3, r = Wtot & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  };
  const int p = tid & 3, lr = tid >> 2;            // chunk p of row lr of each 128-row piece
  // 64-byte rows: ds_read_b128 is served in four groups of 16 lanes holding every j once, with k-chunk g = 0 for j in {0-3, 12-15}
  // and 1 for j in {4-11} (or 1 / 0; 2 / 3; 3 / 2): rows j, j+4, j+8, j+12 share their banks, so the chunk swizzle f(j >> 2) must make
  // {f0, f1^1, f2^1, f3} and {f0^1, f1, f2, f3^1} both permutations of 0..3: f = (0, 2, 3, 1)
  const int c8 = (p ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3)) * 8;
  const bf16_t* abase = a.A; const bf16_t* wbase = a.W;
  const int64_t step = (int64_t)128 * a.K;
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);
  const unsigned dump = lds0 + NSLOT * STAGE;
  int iv = blockIdx.x, ikt = 0, islot = 0;
  bool idone = iv >= Wtot;
  auto setup = [&](int v) {
    const int w = remap(v), tn = w % tilesN, tm = w / tilesN;
    abase = a.A + (int64_t)(tm * BM + lr) * a.K + c8;
    wbase = a.W + (int64_t)(tn * BN + lr) * a.K + c8;
  };
  if (!idone) setup(iv);
  auto piece = [&](int i) {
    const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + islot * STAGE + wave * 1024);
    if (i < 2) glds16(abase + i * step, sa + 8192 * i);
    else if (i < 4) glds16(wbase + (i - 2) * step, sa + 8192 * i);
    else if (wave < 4) glds16(wbase + 2 * step, sa + 8192 * 4);
    else glds16(a.W, (unsigned)__builtin_amdgcn_readfirstlane(dump + (wave - 4) * 1024));
  };
  auto finish = [&]() {
    abase += 32; wbase += 32;
    islot = islot + 1 == NSLOT ? 0 : islot + 1;
    if (++ikt == ktiles) {
      ikt = 0; iv += G;
      if (iv >= Wtot) { idone = true; abase = a.A; wbase = a.W; }
      else setup(iv);
    }
  };
  f32x4 acc[NF][MF];
  bf16x8 af[MF], wq[4];
  const int pc = g ^ ((0x78 >> (2 * ((j >> 2) & 3))) & 3);
  auto lds_a = [&](int mf, int slot) -> bf16x8 { return *(const bf16x8*)(smem + slot * STAGE + ((wm * 64 + mf * 16 + j) * 4 + pc) * 16); };
  auto lds_w = [&](int nf, int slot) -> bf16x8 { return *(const bf16x8*)(smem + slot * STAGE + BM * 64 + ((wn * 160 + nf * 16 + j) * 4 + pc) * 16); };
  // prologue: D tiles in flight (a parked cursor re-reads the first rows: harmless)
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int i = 0; i < P; ++i) piece(i);
    finish();
  }
  int cslot = 0;
  for (int cv = blockIdx.x; cv < Wtot; cv += G) {
    const int w = remap(cv), tn = w % tilesN, tm = w / tilesN;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < ktiles; ++t) {
      if (t == 0) wait_vmcnt<0>();            // (the previous item's stores are in the count)
      else wait_vmcnt<(D - 1) * P>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) af[mf] = lds_a(mf, cslot);
#pragma unroll
      for (int q = 0; q < 3; ++q) wq[q] = lds_w(q, cslot);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        if (q + 3 < NF) wq[(q + 3) & 3] = lds_w(q + 3, cslot);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[q][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q & 3], af[mf], acc[q][mf], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (q < P) {
          __builtin_amdgcn_sched_barrier(0);
          piece(q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      finish();
      cslot = cslot + 1 == NSLOT ? 0 : cslot + 1;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = tm * BM + wm * 64 + mf * 16 + j;
        const int n = tn * BN + wn * 160 + nf * 16 + g * 4;
        uint2 pk;
        pk.x = pack2bf(acc[nf][mf][0], acc[nf][mf][1]);
        pk.y = pack2bf(acc[nf][mf][2], acc[nf][mf][3]);
        *(uint2*)(a.C + m * a.N + n) = pk;
      }
  }
  wait_vmcnt<0>();
}


// ---- round 6, third experiment: the 32-deep 4-slot ring with the block's two wave groups HALF A UNIT OUT OF STEP.  Waves w and w + 4
// share a SIMD (a workgroup's waves go to the SIMDs cyclically); X = waves 0-3 walks the classic sequence per 32-deep unit k --
// `s_waitcnt`, barrier B_k, fragment reads, 40 MFMAs -- while Y = waves 4-7 runs one unit behind and meets B_k in the MIDDLE of its
// unit k - 1 (between MFMA groups 4 and 5): the data of unit k - 1 was published by B_{k-1}, so Y never waits for data at a unit
// boundary (after B_k it pre-reads the first fragments of unit k into a second A register set), and while X stands at its boundary
// (wait + barrier + LDS read latency, ~300 clocks in the lock-step kernels: that is what made the plain 32-deep ring 10 - 15 %
// slower) Y's MFMAs keep the SIMD's pipe busy, and vice versa.  Slot of unit k + 2 (= slot of unit k - 2) is refilled after B_k:
// X finished unit k - 2 two phases ago, Y finished it before the middle of its unit k - 1.  Every wave: one barrier per unit, 5
// pieces per barrier, `vmcnt(5)` before a barrier (vmcnt(0) when the epilogue's stores are in the count).
template <int DUMMY>
__global__ __launch_bounds__(512) void kloop32s(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSLOT = 4, BM = 256, BN = 320, STAGE = (BM + BN) * 64, P = 5;
  constexpr int MF = 4, NF = 10;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool isY = wave >= 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  const int tilesN = a.N / BN, tilesM = a.M / BM, Wtot = tilesM * tilesN, G = gridDim.x, ktiles = a.K >> 5;
  auto remap = [&](int v) {
    const int xcd = v & 7, q = Wtot >> 3, r = Wtot & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  };
  const int p = tid & 3, lr = tid >> 2;
  const int c8 = (p ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3)) * 8;
  const bf16_t* abase = a.A; const bf16_t* wbase = a.W;
  const int64_t step = (int64_t)128 * a.K;
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);
  const unsigned dump = lds0 + NSLOT * STAGE;
  int iv = blockIdx.x, ikt = 0, islot = 0;
  bool idone = iv >= Wtot;
  auto setup = [&](int v) {
    const int w = remap(v), tn = w % tilesN, tm = w / tilesN;
    abase = a.A + (int64_t)(tm * BM + lr) * a.K + c8;
    wbase = a.W + (int64_t)(tn * BN + lr) * a.K + c8;
  };
  if (!idone) setup(iv);
  auto piece = [&](int i) {
    const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + islot * STAGE + wave * 1024);
    if (i < 2) glds16(abase + i * step, sa + 8192 * i);
    else if (i < 4) glds16(wbase + (i - 2) * step, sa + 8192 * i);
    else if (wave < 4) glds16(wbase + 2 * step, sa + 8192 * 4);
    else glds16(a.W, (unsigned)__builtin_amdgcn_readfirstlane(dump + (wave - 4) * 1024));
  };
  auto finish = [&]() {
    abase += 32; wbase += 32;
    islot = (islot + 1) & 3;
    if (++ikt == ktiles) {
      ikt = 0; iv += G;
      if (iv >= Wtot) { idone = true; abase = a.A; wbase = a.W; }
      else setup(iv);
    }
  };
  f32x4 acc[NF][MF];
  bf16x8 af[MF], afn[MF], wq[5];   // (W ring of 5: 10 groups per unit, so the ring position is the same at every unit start)
  const int pc = g ^ ((0x78 >> (2 * ((j >> 2) & 3))) & 3);
  auto lds_a = [&](int mf, int slot) -> bf16x8 { return *(const bf16x8*)(smem + slot * STAGE + ((wm * 64 + mf * 16 + j) * 4 + pc) * 16); };
  auto lds_w = [&](int nf, int slot) -> bf16x8 { return *(const bf16x8*)(smem + slot * STAGE + BM * 64 + ((wn * 160 + nf * 16 + j) * 4 + pc) * 16); };
  auto store_tile = [&](int tm, int tn) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t m = tm * BM + wm * 64 + mf * 16 + j;
        const int n = tn * BN + wn * 160 + nf * 16 + g * 4;
        uint2 pk;
        pk.x = pack2bf(acc[nf][mf][0], acc[nf][mf][1]);
        pk.y = pack2bf(acc[nf][mf][2], acc[nf][mf][3]);
        *(uint2*)(a.C + m * a.N + n) = pk;
      }
  };
  // prologue: units 0 and 1 in flight
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int i = 0; i < P; ++i) piece(i);
    finish();
  }
  int cslot = 0;
  if (!isY) {
    // ================= X: barrier at the unit boundary =================
    for (int cv = blockIdx.x; cv < Wtot; cv += G) {
      const int w = remap(cv), tn = w % tilesN, tm = w / tilesN;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < ktiles; ++t) {
        if (t == 0) wait_vmcnt<0>();
        else wait_vmcnt<P>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) af[mf] = lds_a(mf, cslot);
#pragma unroll
        for (int q = 0; q < 3; ++q) wq[q] = lds_w(q, cslot);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NF; ++q) {
          if (q + 3 < NF) wq[(q + 3) % 5] = lds_w(q + 3, cslot);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc[q][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q % 5], af[mf], acc[q][mf], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          if (q < P) {
            __builtin_amdgcn_sched_barrier(0);
            piece(q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        finish();
        cslot = (cslot + 1) & 3;
      }
      store_tile(tm, tn);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // (pairs with Y's last mid-unit barrier)
  } else {
    // ================= Y: one unit behind, barrier in the middle of the unit =================
    wait_vmcnt<P>();
    __builtin_amdgcn_s_barrier();   // B_0
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < P; ++i) piece(i);   // unit 2
    finish();
    bool fresh = true;                      // the unit's first fragments are not pre-read (first unit of an item)
    for (int cv = blockIdx.x; cv < Wtot; cv += G) {
      const int w = remap(cv), tn = w % tilesN, tm = w / tilesN;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < ktiles; ++t) {
        if (fresh) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) af[mf] = lds_a(mf, cslot);
#pragma unroll
          for (int q = 0; q < 3; ++q) wq[q] = lds_w(q, cslot);
        } else {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) af[mf] = afn[mf];
        }
        const bool pre = t + 1 < ktiles;    // pre-read the next unit of THIS item after the barrier
        const int nslot = (cslot + 1) & 3;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NF; ++q) {
          if (q == 5) {
            if (t == 0) wait_vmcnt<0>();    // (the previous item's stores are in the count)
            else wait_vmcnt<P>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (pre) {
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) afn[mf] = lds_a(mf, nslot);
            }
          }
          if (q + 3 < NF) wq[(q + 3) % 5] = lds_w(q + 3, cslot);
          else if (pre) wq[(q + 3) % 5] = lds_w(q + 3 - NF, nslot);     // q = 7, 8, 9 -> the next unit's W fragments 0, 1, 2
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc[q][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q % 5], af[mf], acc[q][mf], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          if (q >= 5) {
            __builtin_amdgcn_sched_barrier(0);
            piece(q - 5);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        finish();
        cslot = nslot;
        fresh = !pre;
      }
      store_tile(tm, tn);
    }
    wait_vmcnt<0>();
  }
}

static float bf2f_h(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

template <int NWM, int NWN, int BN, int X>
static double run(const char* name, const Args& a, const std::vector<bf16_t>& hA, const std::vector<bf16_t>& hW, int reps) {
  constexpr int smem = 2 * (256 + BN) * 128;
  if (a.N % BN || a.M % 256 || a.K % 64) return 0;
  hipFuncSetAttribute((const void*)kloop<NWM, NWN, BN, X>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int items = (a.M / 256) * (a.N / BN);
  const dim3 grid(items < 256 ? items : 256);
  hipMemset(a.C, 0, (size_t)a.M * a.N * 2);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((kloop<NWM, NWN, BN, X>), grid, dim3(NWM * NWN * 64), smem, 0, a);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return 0; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kloop<NWM, NWN, BN, X>), grid, dim3(NWM * NWN * 64), smem, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, tf = 2.0 * a.M * a.N * a.K / us / 1e6;
  // sampled check
  std::vector<bf16_t> hC((size_t)a.M * a.N);
  hipMemcpy(hC.data(), a.C, hC.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 64; ++s) {
    const int m = (int)(((uint64_t)s * 2654435761u + 17) % a.M), n = (int)(((uint64_t)s * 40503u + 5) % a.N);
    double ref = 0;
    for (int k = 0; k < a.K; ++k) ref += (double)bf2f_h(hA[(size_t)m * a.K + k]) * bf2f_h(hW[(size_t)n * a.K + k]);
    const double err = fabs(bf2f_h(hC[(size_t)m * a.N + n]) - ref) / (fabs(ref) + 1.0);
    worst = err > worst ? err : worst;
  }
  printf("  %-34s %9.1f us %8.1f TFLOP/s   check %.2e %s\n", name, us, tf, worst, worst < 2e-2 ? "ok" : "MISMATCH");
  fflush(stdout);
  return tf;
}

template <int NSLOT>
static double run32(const char* name, const Args& a, const std::vector<bf16_t>& hA, const std::vector<bf16_t>& hW, int reps, bool stag = false) {
  constexpr int smem = NSLOT * (256 + 320) * 64 + 4096;
  void (*kern)(const Args) = stag ? kloop32s<0> : kloop32<NSLOT>;
  if (a.N % 320 || a.M % 256 || a.K % 64) return 0;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int items = (a.M / 256) * (a.N / 320);
  const dim3 grid(items < 256 ? items : 256);
  hipMemset(a.C, 0, (size_t)a.M * a.N * 2);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), smem, 0, a);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return 0; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), smem, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, tf = 2.0 * a.M * a.N * a.K / us / 1e6;
  std::vector<bf16_t> hC((size_t)a.M * a.N);
  hipMemcpy(hC.data(), a.C, hC.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 64; ++s) {
    const int m = (int)(((uint64_t)s * 2654435761u + 17) % a.M), n = (int)(((uint64_t)s * 40503u + 5) % a.N);
    double ref = 0;
    for (int k = 0; k < a.K; ++k) ref += (double)bf2f_h(hA[(size_t)m * a.K + k]) * bf2f_h(hW[(size_t)n * a.K + k]);
    const double err = fabs(bf2f_h(hC[(size_t)m * a.N + n]) - ref) / (fabs(ref) + 1.0);
    worst = err > worst ? err : worst;
  }
  printf("  %-34s %9.1f us %8.1f TFLOP/s   check %.2e %s\n", name, us, tf, worst, worst < 2e-2 ? "ok" : "MISMATCH");
  fflush(stdout);
  return tf;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int fill = argc > 2 ? atoi(argv[2]) : 0;   // 0: uniform random; 1: zeros; 2: constant 1.0 / 0.05 (no toggling, non-zero)
  struct Shape { int M, N, K; };
  const int only32 = argc > 3 ? atoi(argv[3]) : 0;   // 1: only the product structure and the 32-deep ring variants
  const Shape shapes0[] = {{16384, 3840, 4096}, {131072, 320, 2880}, {32768, 1920, 1152}, {131072, 320, 320}, {131072, 2560, 320}, {131072, 320, 1280}};   // long-K square-ish, conv-like, DiT-like, short-K
  // only32 == 4: the transformer denoisers' own widths (SD3-medium D = 1536 at 4096 + 333 tokens x B = 4 ... 8, PixArt D = 1152): which N tile?
  const Shape shapes4[] = {{16384, 4608, 1536}, {16384, 1536, 1536}, {16384, 6144, 1536}, {16384, 1536, 6144}, {32768, 4608, 1152}, {32768, 1152, 4608}, {32768, 3456, 1152}};
  std::vector<Shape> shapes(only32 == 4 ? std::begin(shapes4) : std::begin(shapes0), only32 == 4 ? std::end(shapes4) : std::end(shapes0));
  for (const Shape& s : shapes) {
    std::vector<bf16_t> hA((size_t)s.M * s.K), hW((size_t)s.N * s.K);
    srand(1);
    for (auto& v : hA) v = fill == 1 ? 0 : f2bf_h(fill == 2 ? 1.f : (float)rand() / RAND_MAX * 2.f - 1.f);
    for (auto& v : hW) v = fill == 1 ? 0 : f2bf_h(fill == 2 ? 0.05f : ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f);
    Args a{};
    hipMalloc((void**)&a.A, hA.size() * 2); hipMalloc((void**)&a.W, hW.size() * 2); hipMalloc((void**)&a.C, (size_t)s.M * s.N * 2);
    hipMemcpy((void*)a.A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy((void*)a.W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    a.M = s.M; a.N = s.N; a.K = s.K;
    printf("M=%d N=%d K=%d fill=%d\n", s.M, s.N, s.K, fill);
    for (int round = 0; round < 2; ++round) {   // interleaved rounds (run-to-run noise, clock state)
      if (only32 == 4) {   // N tile of the product structure for widths 320 does not divide
        run<4, 2, 192, 0>("V8 256x192 (product structure)", a, hA, hW, reps);
        run<4, 2, 256, 0>("V8 256x256 (product structure)", a, hA, hW, reps);
        run<4, 2, 128, 0>("V8 256x128 (product structure)", a, hA, hW, reps);
        run<4, 2, 384, 0>("V8 256x384 (product structure)", a, hA, hW, reps);
        continue;
      }
      run<4, 2, 320, 0>("V8 256x320 (product structure)", a, hA, hW, reps);
      if (only32 == 3) {   // cache policy of the LDS-DMA pieces
        run<4, 2, 320, 256>("V8 256x320 A pieces nt", a, hA, hW, reps);
        run<4, 2, 320, 768>("V8 256x320 A and W pieces nt", a, hA, hW, reps);
        continue;
      }
      if (only32 == 2) {   // fill-traffic ablations of the product structure (WRONG results: the check column says MISMATCH)
        run<4, 2, 320, 32>("V8 256x320 A staged every 3rd tile", a, hA, hW, reps);
        run<4, 2, 320, 64>("V8 256x320 A never staged", a, hA, hW, reps);
        run<4, 2, 320, 128>("V8 256x320 W never staged", a, hA, hW, reps);
        run<4, 2, 320, 192>("V8 256x320 nothing staged", a, hA, hW, reps);
        continue;
      }
      run32<3>("V8 256x320 32-deep, 3 slots, counted", a, hA, hW, reps);
      run32<4>("V8 256x320 32-deep, 4 slots, counted", a, hA, hW, reps);
      run32<4>("V8 256x320 32-deep, 4 slots, STAGGERED halves", a, hA, hW, reps, true);
      if (only32) continue;
      run<4, 2, 320, 1>("V8 256x320 spread pieces", a, hA, hW, reps);
      run<4, 2, 320, 2>("V8 256x320 no setprio", a, hA, hW, reps);
      run<2, 2, 320, 0>("V4 256x320", a, hA, hW, reps);
      run<2, 2, 320, 18>("V4 256x320 asm-class MFMA", a, hA, hW, reps);
      run<2, 2, 320, 22>("V4 256x320 asm, A ahead", a, hA, hW, reps);
      run<2, 2, 320, 30>("V4 256x320 asm, A ahead, W ring 7", a, hA, hW, reps);
      run<2, 2, 384, 18>("V4 256x384 asm-class MFMA", a, hA, hW, reps);
      run<2, 2, 384, 30>("V4 256x384 asm, A ahead, W ring 7", a, hA, hW, reps);
      run<2, 2, 256, 2>("V4 256x256 builtin (acc = 256 regs)", a, hA, hW, reps);
      run<2, 2, 256, 18>("V4 256x256 asm-class", a, hA, hW, reps);
      run<4, 2, 192, 0>("V8 256x192 (product structure)", a, hA, hW, reps);
    }
    hipFree((void*)a.A); hipFree((void*)a.W); hipFree(a.C);
  }
  return 0;
}
