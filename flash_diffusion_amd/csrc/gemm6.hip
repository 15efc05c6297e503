// 128 x 320 x 64 bf16 MFMA GEMM / implicit-GEMM conv tile for gfx950 with a DEFERRED epilogue (round 4).
//
// Why: measured on the 256 x 320 kernel (scripts/rowbench.py dev, profiles/r4_rowbench_*.txt) the short-K row GEMMs of the UNets
// spend 50 - 65 % of their time in the epilogue -- M = 131072, N = K = 320 with a residual: K loop 27 us, epilogue 42 us -- and the
// two phases never overlap: all eight waves of a block sit in the epilogue (stores + residual reads at the access pattern's
// 3.7 - 5 TB/s, MFMA pipe idle), then all in the K loop (MFMA, no stores).  One block per CU, identical work everywhere, so the
// whole chip alternates.  The 160 accumulator registers of a 64 x 160 wave tile leave no room to keep a finished tile around.
//
// This kernel halves the wave tile (8 waves as 4(M) x 2(N) over 128 x 320: 32 x 160 = 80 accumulator registers) and keeps the
// FINISHED tile beside the running one: when an item's K loop ends its accumulators are rounded to bf16 in the store layout
// (`prev`, 40 registers: two fp32 sets plus the fragments of the K loop do not fit 256 registers -- the first version spilled
// > 100 of them into the loop), and the epilogue of `prev` -- residual add, store -- is woven into the first K tiles of the NEXT
// item, four 16-byte steps per tile placed behind the tile's ring pieces, where their VALU work issues in the shadow of the MFMAs
// and their memory traffic mixes with the K loop's reads.  With a residual the sum is therefore rounded twice (bf16(bf16(acc +
// bias) + residual)): what PyTorch's bf16 autocast does with `to_out(x) + residual`, one rounding more than the fused epilogue of
// gemm4.hip.
//
//  * Everything that touches VMEM in the loop is inline asm (ring pieces, residual loads, stores), so the compiler's waitcnt
//    pass sees none of it and the only waits are the counted ones placed here: VMEM returns in order, so `vmcnt(n)` with n = the
//    operations issued behind the one waited for.  A tile hand-over waits for the ring pieces and leaves the epilogue's stores
//    and next residual chunks in flight (`tail`); an epilogue step waits for its residual chunk, issued one tile earlier.
//  * Bias and the per-sample row vector (time embedding) are not epilogue terms here: their 320-column slice is DMA'd into LDS
//    with the item's first tile and becomes the accumulators' initial value (fp32: acc = bias + rowvec + sum; the general
//    epilogue adds them after the sum -- same value up to fp32 rounding order).
//  * 2-slot LDS ring as gemm4.hip (2 x 56 KB), same source-side swizzle, same persistent XCD-aware item order.
// Restrictions (gemm6_eligible): whole tiles (M % 128, N % 320, K % 64), bf16 output, alpha = 1, no activation, no split-K /
// atomics / GroupNorm sums / pre-activation save, a row vector only when a 128-row tile lies inside one sample.
#include "gemm_tile.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void g6_load16(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void g6_store16(void* p, const u32x4& v) {   // (s_nop: VMEM store data hazard, the asm is invisible to hipcc)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v));
}
template <int N>
__device__ __forceinline__ void g6_wait_use(u32x4& v) {   // counted wait; every later use of v is ordered behind it
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}
__device__ __forceinline__ void g6_wait_n(int n) {   // tile hand-over: n = VMEM operations issued behind the last ring piece
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 12: wait_vmcnt<12>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void gemm6_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 128, BN = 320;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int AR = BM / 64, WR = BN / 64, NP = AR + WR;   // LDS-DMA pieces (1 KiB per wave) per thread per tile
  constexpr int MF = 2, NF = BN / 32, NQ = 2 * NF;          // NQ MFMA groups (k-step, W fragment) of MF MFMAs per tile
  constexpr int NS = (NF / 2) * MF;                         // epilogue steps of a wave tile: (column pair, row fragment), 16 B per lane
  constexpr int SPT = 4;                                    // steps woven into one K tile
  constexpr int NT = (NS + SPT - 1) / SPT;                  // K tiles an epilogue is spread over
  constexpr int VEC_OFF = 2 * STAGE, ROWV_OFF = VEC_OFF + BN * 4;   // LDS: bias slice (fp32), row-vector slice (bf16)
  constexpr int RB_OFF = VEC_OFF + 2048;                            // LDS: residual chunks, SPT x 1 KiB per wave (16 B per lane)
  static_assert(NP + 1 <= 8 && NQ >= 18, "pieces at groups 0 .. NP, epilogue steps at groups 8, 11, 14, 17");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  const bool RES = a.residual != nullptr;

  // ---- persistent work loop: item = one 128 x 320 tile; block b takes items b, b+G, b+2G, ... ----
  const int tilesN = a.N / BN, tilesM = a.M / BM;
  const int Wtot = tilesM * tilesN;
  const int G = gridDim.x;
  const int ktiles = a.K >> 6;
  auto remap = [&](int v) {  // XCD-aware (block b runs on XCD b % 8; G % 8 == 0 or G == Wtot)
    const int xcd = v & 7, q = Wtot >> 3, r = Wtot & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  };
  struct Item { int m0, n0; };
  auto item_of = [&](int v) {
    const int w = remap(v);
    Item it;
    it.n0 = (w % tilesN) * BN;
    it.m0 = (w / tilesN) * BM;
    return it;
  };

  // ---- loader state (as gemm4.hip, two A pieces): thread t fills chunk t&7 of row t>>3 of each 64-row group; the chunk holds
  // the logical k-chunk (t&7) ^ ((row>>1)&7) ----
  const int p = tid & 7, lr = tid >> 3;
  const int c8 = (p ^ ((lr >> 1) & 7)) * 8;
  const bf16_t* zero = (const bf16_t*)g_zero16b;
  const bf16_t* ap[AR];
  unsigned aok = 0;
  int ayx[AR], apix[AR];
  int ky = 0, kx = 0, cc = 0;
  const bf16_t* abase = zero;
  int64_t astep = 0;
  int akpos = 0, arow = 0;
  const bf16_t* wbase = zero;
  int64_t wstep = 0;
  auto retap = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int by = (ayx[i] >> 16) - 256, bx = (ayx[i] & 0xffff) - 256;
      bool ok = true;
      int sy = 0, sx = 0;
      if (a.dgrad) {
        const int ty = by - ky, tx = bx - kx;
        ok = ty >= 0 && tx >= 0;
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
        const int iy = by + ky, ix = bx + kx;
        ok = iy >= 0 && ix >= 0 && iy < (a.Hin << a.ups) && ix < (a.Win << a.ups);
        sy = iy >> a.ups;
        sx = ix >> a.ups;
      }
      ap[i] = ok ? a.A + ((int64_t)(apix[i] + sy * a.Win + sx) * a.Cin + cc + c8) : zero;
      aok = ok ? (aok | (1u << i)) : (aok & ~(1u << i));
    }
  };
  auto setup_issue = [&](const Item& it) {
    if (MODE == GEMM_ROW) {
      abase = a.A + (int64_t)(it.m0 + lr) * a.lda + c8;
      astep = 64 * a.lda;
      akpos = 0;
      arow = it.m0 + lr;
    } else {
      cc = 0;
      ky = 0;
      kx = 0;
      const int hw = a.Hout * a.Wout;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int m = it.m0 + lr + 64 * i;
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        apix[i] = b * a.Hin * a.Win;
        const int by = a.dgrad ? (oy + a.pad) : (oy * a.stride - a.pad);
        const int bx = a.dgrad ? (ox + a.pad) : (ox * a.stride - a.pad);
        ayx[i] = ((by + 256) << 16) | (bx + 256);
      }
      retap();
    }
    wbase = a.W + (int64_t)(it.n0 + lr) * a.ldw + c8;
    wstep = 64 * a.ldw;
  };
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);
  int iv = blockIdx.x, ikt = 0, islot = 0;
  bool ihave = false, idone = false;
  bool inew = false;          // the tile being issued is the first of its item: its bias / row-vector slices ride along
  int in0 = 0, im0 = 0;       // that item's column / row origin
  auto issue_prepare = [&]() -> bool {  // position the cursor on the next k-tile; false when none is left
    if (idone) return false;
    if (!ihave || ikt == ktiles) {
      if (ihave) iv += G;
      if (iv >= Wtot) {
        idone = true;
        return false;
      }
      const Item it = item_of(iv);
      ihave = true;
      inew = true;
      in0 = it.n0;
      im0 = it.m0;
      ikt = 0;
      setup_issue(it);
    }
    return true;
  };
  auto park_issue = [&]() {  // past the last tile: the pieces read a zero page (keeps one instruction stream)
    abase = zero; astep = 0;
    wbase = zero; wstep = 0;
    aok = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) ap[i] = zero;
  };
  auto slot_base = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane(lds0 + islot * STAGE + wave * 1024); };
  auto piece = [&](int i, unsigned sa) {
    if (i < AR) {
      if (MODE == GEMM_ROW) {
        glds16(abase + i * astep, sa + 512 * 16 * i);
      } else {
        glds16(ap[i], sa + 512 * 16 * i);
        ap[i] += ((aok >> i) & 1u) << 6;
      }
    } else {
      glds16(wbase + (i - AR) * wstep, sa + BM * 128 + 512 * 16 * (i - AR));
    }
  };
  // the item's bias (fp32, 320 values = 80 x 16 B: waves 0, 1) and row-vector (bf16, 40 x 16 B: wave 2) slices, one more "piece"
  // of the item's first tile.  Only some waves issue it and only on an item's first tile: the counted waits below (per wave)
  // assume it was NOT issued -- a wave that did issue it then waits for one operation more than it must, never for one less
  auto vec_piece = [&]() {
    const bool b_on = inew && a.bias != nullptr && tid < BN / 4;
    const bool v_on = inew && a.rowvec != nullptr && tid >= 128 && tid < 128 + BN / 8;
    const char* src = (const char*)zero;
    unsigned dst = lds0 + VEC_OFF;
    if (b_on) {
      src = (const char*)(a.bias + in0 + 4 * tid);
      dst = lds0 + VEC_OFF + (wave ? 1024 : 0);
    } else if (v_on) {
      src = (const char*)(a.rowvec + (int64_t)(im0 / a.rows_per_batch) * a.rowvec_ld + in0 + 8 * (tid - 128));
      dst = lds0 + ROWV_OFF;
    }
    const unsigned d = (unsigned)__builtin_amdgcn_readfirstlane(dst);
    if (b_on || v_on) glds16(src, d);
  };
  auto issue_finish = [&]() {
    if (MODE == GEMM_ROW) {
      abase += astep ? 64 : 0;
      akpos += 64;
      if (a.A2 && akpos == a.K1 && astep) {   // the next tile is the first of the second segment (uniform)
        abase = a.A2 + (int64_t)arow * a.lda2 + c8;
        astep = 64 * a.lda2;
      }
    } else {
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
    wbase += wstep ? 64 : 0;
    islot ^= 1;
    ++ikt;
    inew = false;
  };

  // ---- fragments and the two accumulator sets ----
  f32x4 acc[NF][MF];
  u32x4 prev[NS];   // the finished tile: step i = (pair i / MF, row fragment i % MF), 8 bf16 = this lane's 16 bytes of the store
  constexpr int WD = 6;   // W fragments in flight ahead of their MFMAs (two MFMAs per fragment: the LDS latency needs more groups than gemm4's 3)
  bf16x8 af[2][MF], wq[8];
  // LDS byte offsets of this lane's fragments inside a ring slot: k-step 0 of row fragment 0 (A) / W fragment 0; a row or W
  // fragment further on is + 2048, k-step 1 is ^ 64 ((ks * 4 + g) ^ (j >> 1) with g < 4).  Every tile body re-derives its four
  // base registers from these two through an opaque move, so that the 24 fragment addresses of a tile are base + immediate and
  // nothing address-shaped is hoisted out of the tile loop (with four copies of the body that cost > 100 spilled registers, and a
  // spill reload inside the loop drains the LDS-DMA queue: hipcc waits vmcnt(0) for its own scratch loads)
  const unsigned a_off = ((wm * 32 + j) * 8 + (g ^ (j >> 1))) * 16;
  const unsigned w_off = BM * 128 + ((wn * (BN / 2) + j) * 8 + (g ^ (j >> 1))) * 16;
  unsigned fa0 = 0, fa1 = 0, fw0 = 0, fw1 = 0;
  auto frag_bases = [&](int slot) {
    unsigned xa = a_off + slot * STAGE, xw = w_off + slot * STAGE;
    asm volatile("" : "+v"(xa), "+v"(xw));
    fa0 = xa; fa1 = xa ^ 64u; fw0 = xw; fw1 = xw ^ 64u;
  };
  auto lds_a = [&](int ks, int mf) -> bf16x8 { return *(const bf16x8*)(smem + (ks ? fa1 : fa0) + mf * 2048); };
  auto lds_w = [&](int q) -> bf16x8 {  // q = ks * NF + nf
    const int ks = q / NF, nf = q - ks * NF;
    return *(const bf16x8*)(smem + (ks ? fw1 : fw0) + nf * 2048);
  };

  // ---- deferred epilogue of `prev`: lane (g, j) owns, per (pair pr, row fragment mf), row mf*16 + j and 8 columns ----
  bf16_t* pc_ = nullptr;          // C + row * ldc + first column of pair 0
  const bf16_t* pr_ = nullptr;    // residual likewise
  const int64_t cstep = 16 * a.ldc, rstep = RES ? 16 * a.ldr : 0;
  // The residual chunks of the next SPT steps travel by LDS-DMA into a per-wave page (lane l's 16 bytes at l * 16) and are read
  // back with ds_read_b128 when their step comes up.  (A first version loaded them into registers from inline asm: hipcc knows
  // nothing about an asm output arriving LATER -- it copied or reused those registers before the data landed.  Through LDS no
  // register is written behind the compiler's back; the LDS-DMA pieces already rely on that.)
  const unsigned rb_lds = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + RB_OFF + wave * (SPT * 1024));
  auto epi_loads = [&](int first) {   // residual chunks of steps first .. first + SPT - 1 (static `first`); returns how many
    int n = 0;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      const int i = first + k;
      if (i < NS) {
        const bf16_t* q = pr_;
        asm volatile("" : "+v"(q));   // (opaque: the ten step addresses are otherwise hoisted out of the tile loop, 40 registers)
        glds16(q + (i % MF) * rstep + 32 * (i / MF), rb_lds + k * 1024);
        ++n;
      }
    }
    return n;
  };
  auto epi_step = [&](int i, int k, auto WAITC) {   // step i (static), chunk k of the page, residual wait count WAITC::value
    const int pr = i / MF, mf = i % MF;
    u32x4 pk = prev[i];
    asm volatile("" : "+v"(pk));   // (opaque: the unpacked fp32 values of `prev` are loop-invariant and would be hoisted -- and spilled)
    if (RES) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(WAITC)::value) : "memory");
      const u32x4 rr = *(const u32x4*)(smem + RB_OFF + wave * (SPT * 1024) + k * 1024 + lane * 16);
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[2 * r] = __uint_as_float(pk[r] << 16) + __uint_as_float(rr[r] << 16);
        v[2 * r + 1] = __uint_as_float(pk[r] & 0xffff0000u) + __uint_as_float(rr[r] & 0xffff0000u);
      }
      pk[0] = pack2bf(v[0], v[1]); pk[1] = pack2bf(v[2], v[3]); pk[2] = pack2bf(v[4], v[5]); pk[3] = pack2bf(v[6], v[7]);
    }
    bf16_t* q = pc_;
    asm volatile("" : "+v"(q));
    g6_store16(q + mf * cstep + 32 * pr, pk);
  };
  // everything of `prev` from step `first` on, synchronously (short K loops; the last item).  The residual chunks of steps
  // first .. first + SPT - 1 are in flight already
  auto epi_flush = [&](auto FIRST) {
    constexpr int first = decltype(FIRST)::value;
#pragma unroll
    for (int b = first; b < NS; b += SPT) {
      if (b > first && RES) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the page's previous chunks have been read
        epi_loads(b);
      }
#pragma unroll
      for (int k = 0; k < SPT; ++k)
        if (b + k < NS) epi_step(b + k, k, std::integral_constant<int, 0>{});
    }
  };

  // ---- one K tile: MFMAs into acc, next tile's ring pieces, and (TS >= 0) steps TS*SPT .. of prev's epilogue ----
  int cslot = 0;
  int tail = 0;   // VMEM operations issued behind the last ring piece of the tile waited for next
  auto tile_body = [&](auto TSC) {
    constexpr int TS = decltype(TSC)::value;
    constexpr int NSTEP = TS < 0 ? 0 : (NS - TS * SPT < SPT ? NS - TS * SPT : SPT);            // steps woven into this tile
    constexpr int NLOAD = TS < 0 ? 0 : (NS - (TS + 1) * SPT <= 0 ? 0 : (NS - (TS + 1) * SPT < SPT ? NS - (TS + 1) * SPT : SPT));
    frag_bases(cslot);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) af[0][mf] = lds_a(0, mf);
#pragma unroll
    for (int q = 0; q < WD; ++q) wq[q] = lds_w(q);
    const bool have = issue_prepare();
    if (!have) park_issue();
    const unsigned sa = slot_base();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ks = q / NF, nf = q % NF;
      if (q + WD < NQ) wq[(q + WD) & 7] = lds_w(q + WD);
      if (q == 3) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) af[1][mf] = lds_a(1, mf);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
        acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q & 7], af[ks][mf], acc[nf][mf], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (q < NP) {
        __builtin_amdgcn_sched_barrier(0);
        piece(q, sa);
      } else if (q == NP) {
        __builtin_amdgcn_sched_barrier(0);
        vec_piece();
      } else {
        if constexpr (TS >= 0) {
          if (q >= 8 && (q - 8) % 3 == 0 && (q - 8) / 3 < NSTEP) {
            // residual chunk k of this tile's batch (NSTEP chunks, issued one tile ago); behind it in this wave's queue: the rest
            // of the batch, this tile's NP pieces (+ the vector piece, not counted: see vec_piece) and this tile's k earlier stores
            __builtin_amdgcn_sched_barrier(0);
            epi_step(TS * SPT + (q - 8) / 3, (q - 8) / 3, std::integral_constant<int, NSTEP - 1 + NP>{});
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (have) issue_finish();
    int t = NSTEP;
    if (TS >= 0 && RES && NLOAD > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's steps have read their chunks: the page may be refilled
      t += epi_loads((TS + 1) * SPT);
    }
    tail = t;
    cslot ^= 1;
  };

  // prologue: tile 0 of the first item -> slot 0
  if (issue_prepare()) {
    const unsigned sa = slot_base();
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, sa);
    vec_piece();
    issue_finish();
  }

  // hand-over at the top of a tile: this wave's pieces of the tile have landed (the epilogue operations issued behind them may
  // stay in flight) and its reads of the other slot are done; an item's first tile also seeds the accumulators from the item's
  // bias (+ row vector) slice, which landed with that tile
  auto tile_top = [&](bool first) {
    g6_wait_n(tail);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // (compiler) no LDS read of the new tile may be scheduled above the barrier
    if (first) {
      const int cw = wn * (BN / 2) + g * 4;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias) b = *(const f32x4*)(smem + VEC_OFF + (cw + nf * 16) * 4);
        if (a.rowvec) {
          const u16x4 rv = *(const u16x4*)(smem + ROWV_OFF + (cw + nf * 16) * 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) b[r] += bf2f(rv[r]);
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = b;
      }
    }
  };
  static_assert(NT == 3, "the woven K loop below is written out for three epilogue tiles");
  // The item loop is ROTATED: an iteration runs the tail of item i (tiles NT ..), moves its accumulators to `prev`, issues the
  // first residual chunks, and then the head of item i + 1 (tiles 0 .. NT - 1) with the epilogue of `prev` woven in -- so every
  // asm-loaded register is written and consumed inside ONE straight-line stretch.  (Loaded at the end of an iteration and
  // consumed at the top of the next, the residual ring crossed the loop's back edge, hipcc placed register copies there -- copies
  // of registers whose data had not arrived yet: NaNs in the first four steps of every tile.)  K >= 64 NT (gemm6_eligible).
  int cv = blockIdx.x;
  if (cv < Wtot) {
    Item it = item_of(cv);
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {    // head of the block's first item: nothing to finish yet
      tile_top(t == 0);
      tile_body(std::integral_constant<int, -1>{});
    }
    while (true) {
#pragma unroll 1
      for (int t = NT; t < ktiles; ++t) {
        tile_top(false);
        tile_body(std::integral_constant<int, -1>{});
      }
      if (a.dev & 32) {   // (timing ablation of scripts/rowbench.py: the K loop alone, WRONG results)
        cv += G;
        if (cv >= Wtot) break;
        it = item_of(cv);
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
          tile_top(t == 0);
          tile_body(std::integral_constant<int, -1>{});
        }
        continue;
      }
      // this item's accumulators become `prev`: lane (g, j) of a fragment pair gets 8 consecutive columns (v_permlane16_swap), bf16
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int nf = 2 * (i / MF), mf = i % MF;
#pragma unroll
        for (int r = 0; r < 4; ++r) SWAP16(acc[nf][mf][r], acc[nf + 1][mf][r]);
        prev[i][0] = pack2bf(acc[nf][mf][0], acc[nf][mf][1]);
        prev[i][1] = pack2bf(acc[nf][mf][2], acc[nf][mf][3]);
        prev[i][2] = pack2bf(acc[nf + 1][mf][0], acc[nf + 1][mf][1]);
        prev[i][3] = pack2bf(acc[nf + 1][mf][2], acc[nf + 1][mf][3]);
      }
      {
        const int64_t row = it.m0 + wm * 32 + j;
        const int col = it.n0 + wn * (BN / 2) + (g & 1) * 16 + (g >> 1) * 8;
        pc_ = (bf16_t*)a.C + row * a.ldc + col;
        pr_ = RES ? a.residual + row * a.ldr + col : nullptr;
      }
      if (RES) tail += epi_loads(0);
      cv += G;
      if (cv >= Wtot) break;
      it = item_of(cv);
      // head of the next item, the epilogue of `prev` riding on its first three tiles
      tile_top(true);
      tile_body(std::integral_constant<int, 0>{});
      tile_top(false);
      tile_body(std::integral_constant<int, 1>{});
      tile_top(false);
      tile_body(std::integral_constant<int, 2>{});
    }
    if (!(a.dev & 32)) epi_flush(std::integral_constant<int, 0>{});   // the block's last item (its first residual chunks are in flight)
  }
  wait_vmcnt<0>();  // no LDS-DMA (or store) may still be in flight when the workgroup ends
}

template <int MODE>
int launch6_t(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int smem = 2 * (128 + 320) * 128 + 2048 + 8 * 4 * 1024;   // ring + bias / row-vector slices + residual pages
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)gemm6_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int items = (a.M / 128) * (a.N / 320);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    FDMI_HIP(hipGetDevice(&dev));
    FDMI_HIP(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount & ~7) : 256;
    if (ncu < 8) ncu = 8;
  }
  dim3 grid(items < ncu ? items : ncu, 1, 1);  // persistent: one 8-wave block per CU
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_begin(stream, PROF_GEMM4 + MODE, gemm_flops(a));
  FDMI_KLAUNCH(prof, (gemm6_kernel<MODE>), grid, dim3(512), smem, stream, a);
  if (prof) fdmi_prof_end(stream);
  FDMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool gemm6_eligible(const GemmArgs& a) {
  if ((a.K & 63) != 0 || a.K < 192 || (a.M & 127) != 0 || (a.N % 320) != 0) return false;   // (K >= 192: an epilogue rides on three K tiles)
  if (a.act != ACT_NONE || a.out_f32 || a.accum_atomic || a.splitk > 1 || a.preact || a.gn_stats || a.alpha != 1.f) return false;
  if ((a.ldc & 7) || (a.residual && (a.ldr & 7)) || ((uintptr_t)a.C & 15) || ((uintptr_t)a.residual & 15)) return false;
  if (a.bias && ((uintptr_t)a.bias & 15)) return false;
  if (a.rowvec && (a.rowvec_mul || (a.rowvec_ld & 7) || ((uintptr_t)a.rowvec & 15) || (a.rows_per_batch & 127))) return false;
  if (a.mode == GEMM_CONV && ((a.Cin & 63) != 0 || a.Hout > 2048 || a.Wout > 2048)) return false;
  if (a.mode == GEMM_ROW && a.A2 && (a.K1 & 63)) return false;
  return true;
}
int launch_gemm6(const GemmArgs& a, hipStream_t stream) {
  FDMI_CHECK(gemm6_eligible(a), "gemm6: the 128 x 320 deferred-epilogue tile is not applicable to this problem");
  return a.mode == GEMM_ROW ? launch6_t<GEMM_ROW>(a, stream) : launch6_t<GEMM_CONV>(a, stream);
}
