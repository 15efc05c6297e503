// 256 x 320 x 64 bf16 MFMA GEMM / implicit-GEMM conv tile for gfx950 (512 threads = 8 waves as 4(M) x 2(N),
// wave tile 64 x 160 = 4 x 10 accumulator fragments).
//
// Why a second large-tile kernel: the 256 x 160 kernel (gemm3.hip) moves 53 KB through the CU's vector
// L1 per 64-deep K tile (98 flop/B).  One CU's L1 -> LDS path sustains ~42 B/clk (measured: the fill
// time of a tile does not change from 8 to 256 active CUs), the MFMA pipe ~19 clk per 16x16x32, so
// fill (~1300 clk) and MFMA (~1500 clk) of a 256 x 160 tile are both near saturation and every
// imperfection of their overlap shows (measured 2300 clk per tile).  256 x 320 needs 72 KB per
// 2 x 256 x 320 x 64 flop = 142 flop/B: the fill is 57% of the MFMA time.  N = 320 divides every
// channel count of the SD / SDXL UNets, M = B*H*W is a multiple of 256 at every level but the deepest.
//
// Pipeline: 2-slot LDS ring (2 x 72 KB).  While tile k is multiplied from slot k%2, tile k+1 is
// fetched into the other slot by LDS-DMA pieces issued one per MFMA group over the first half of the
// tile; one `s_waitcnt vmcnt(0) lgkmcnt(0)` + `s_barrier` per tile hands the slots over.  W fragments
// are streamed through a 4-deep register ring (10 W fragments per k-step would not fit beside 160
// accumulator registers at 2 waves per SIMD); both k-steps' A fragments stay resident.
#include "gemm_tile.h"

namespace {

// LDS-DMA piece with a wave-uniform 64-bit base (SGPR pair) + a 32-bit lane offset (the 256 x 384 instantiations: their 192 accumulator
// registers leave no room for two 64-bit per-lane pointers plus a 64-bit temporary per piece; the bases advance on the scalar unit)
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
  const uint64_t pv = (uint64_t)(uintptr_t)sbase;
  const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pv), phi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pv >> 32));
  sbase = (const void*)(uintptr_t)(((uint64_t)phi << 32) | (uint64_t)plo);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr));
}

// BN = 320 is the tile described above.  BN = 192 is the same pipeline for widths that 320 does not divide but 192 does --
// the transformer denoisers' 1152 / 1536 / 4608 / 6144 (PixArt, SD3): 56 KB per tile, 110 flop/B, wave tile 64 x 96.
// BN = 384 (round 6; row problems only) is the wide tile for the same widths: 80 KB per tile -- the ring takes the CU's whole 160 KB --,
// 154 flop/B, wave tile 64 x 192 = 192 accumulator registers of the 256 a wave has at two per SIMD (the compiler keeps a handful of
// loop-invariant values in scratch OUTSIDE the K loop: scripts/kloop_spill_audit.py).  scripts/ubench/gemm_kloop.hip (mode 4): +9 ... +16 % over
// the 256 x 192 structure on the SD3 / PixArt widths (profiles/r6_kloop_dit_widths.txt); 256 x 256 quantises worse than 192 and loses.
// GN: the epilogue also accumulates the consumer's GroupNorm statistics (GemmArgs::gn_stats); a separate instantiation, so the
// default kernels are instruction-identical with and without the feature
// DEV: developer instantiation (scripts/rowbench.py, scripts/kbench.py through knob 40 = GemmArgs::dev) -- timing ablations with
// WRONG results (16: A re-read from its first tile; 32: no epilogue) and experiments (0x800: conv K order (channel chunk, tap)
// instead of (tap, channel chunk); bits 16-18 / 20-27: start the blocks in 2 ... 7 phase groups, group i delayed by i * n us: the
// K = 320 GEGLU launch gains 13 % in isolation, 0.1 - 0.4 ms on the C2 step -- profiles/r5_knob_ab_geglu_stagger.txt -- not adopted).
// The production instantiations (DEV = false) contain none of it.
// SKR: in-launch split-K reduction (GemmArgs::sk_tickets, gemm_tile.h::splitk_last_arriver) -- its own instantiation: the second
// epilogue's live ranges pushed a spill into the production conv kernel's K loop when it was compiled into it
// R32: the fp32-residual-stream epilogue (GemmArgs::residual32 / C32, gemm_tile.h::tile_epilogue_r32) -- its own instantiation too
template <int MODE, bool GEGLU, int BN, bool GN = false, bool DEV = false, bool SKR = false, bool R32 = false>
__global__ __launch_bounds__(512) void gemm4_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256;
  static_assert(BN % 64 == 0 && BN >= 128, "two waves along N, 16-wide fragments in pairs, 64-row LDS-DMA pieces");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int AR = 4, WR = BN / 64, NP = AR + WR;  // LDS-DMA pieces (1 KiB per wave) per thread per tile
  constexpr int MF = 4, NF = BN / 32, NQ = 2 * NF;   // NQ MFMA groups (k-step, W fragment) per tile
  static_assert(NP <= NQ && NQ >= 8, "one piece per MFMA group; the W ring runs 3 groups ahead");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  if constexpr (DEV) {   // phase stagger: desynchronise the blocks' K-loop / epilogue phases across the chip
    const int ph = (a.dev >> 16) & 7, dl = (a.dev >> 20) & 0xff;
    if (ph > 1) {
      const int mine = (int)((blockIdx.x >> 3) % ph) * dl;
      for (int i = 0; i < mine; ++i) __builtin_amdgcn_s_sleep(32);   // ~2048 clocks ~ 1 us
    }
  }
  if constexpr (SKR) {
    if (tid == 0) ((volatile SkList*)(smem + 2 * STAGE))->n = 0;   // (ordered before its first use by the K loops' barriers)
  }
  const bool ctap = DEV && MODE == GEMM_CONV && (a.dev & 0x800);   // K tile t = (channel chunk t / taps, tap t % taps)
  int tapi = 0;                 // (ctap) the tap of the tile to issue next
  const bf16_t* wrow = nullptr; // (ctap) W row n0+lr at k = 0

  // ---- persistent work loop: item = (tile, k-split); block b takes items b, b+G, b+2G, ... ----
  const int tilesN = a.N / BN, tilesM = a.M / BM;
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

  // ---- loader state.  Thread t fills chunk t&7 of row t>>3 of each 64-row group; the chunk holds the
  // logical k-chunk (t&7) ^ ((row>>1)&7) (source-side swizzle: conflict-free ds_read_b128 below) ----
  const int p = tid & 7, lr = tid >> 3;
  const int c8 = (p ^ ((lr >> 1) & 7)) * 8;
  const bf16_t* zero = (const bf16_t*)g_zero16b;
  const bf16_t* ap[AR];      // CONV: one source pointer per staged row (re-derived per filter tap)
  unsigned aok = 0;          // CONV: bit i = row i reads real data (advances along Cin)
  int ayx[AR], apix[AR];     // CONV: (y << 16 | x) window origin (biased by +256), batch pixel base
  int ky = 0, kx = 0, cc = 0;
  const bf16_t* abase = zero;  // ROW: row m0+lr; piece i is 64*i rows further
  int64_t astep = 0;
  int akpos = 0, arow = 0;     // ROW with a second A segment (GemmArgs::A2): k of the tile to issue next, this thread's row
  const bf16_t* wbase = zero;  // W row n0+lr; piece i is 64*i rows further
  int64_t wstep = 0;
  // WIDE (BN = 384, row problems without a second A segment): uniform bases + 32-bit lane offsets instead of the per-lane pointers above
  constexpr bool WIDE = BN == 384;
  const char* asb = (const char*)zero;   // A row m0, column kbeg of the tile to issue next (uniform)
  const char* wsb = (const char*)zero;   // W row n0, column kbeg (uniform)
  unsigned aoff = 0u, woff = 0u;         // this lane's row / chunk inside a 64-row group (bytes); 0 while parked: every lane reads the zero page
  const unsigned astepb = WIDE ? (unsigned)(128 * a.lda) : 0u, wstepb = WIDE ? (unsigned)(128 * a.ldw) : 0u;   // 64 rows further (bytes; uniform: added to the BASE on the scalar unit)
  unsigned pmask = 0u;                   // 0 while parked
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
    if (WIDE) {
      asb = (const char*)(a.A + (int64_t)it.m0 * a.lda + it.kbeg);
      wsb = (const char*)(a.W + (int64_t)it.n0 * a.ldw + it.kbeg);
      aoff = (unsigned)((lr * a.lda + c8) * 2);
      woff = (unsigned)((lr * a.ldw + c8) * 2);
      pmask = ~0u;
      return;
    }
    if (MODE == GEMM_ROW) {
      if (a.A2 && it.kbeg >= a.K1) {   // (a k-split that starts behind the seam)
        abase = a.A2 + (int64_t)(it.m0 + lr) * a.lda2 + (it.kbeg - a.K1) + c8;
        astep = 64 * a.lda2;
      } else {
        abase = a.A + (int64_t)(((DEV && (a.dev & 16)) ? 0 : it.m0) + lr) * a.lda + it.kbeg + c8;   // (dev & 16: timing ablation, A from L2)
        astep = 64 * a.lda;
      }
      akpos = it.kbeg;
      arow = it.m0 + lr;
    } else {
      int tap = it.kbeg / a.Cin;
      cc = it.kbeg - tap * a.Cin;
      if (ctap) {
        const int t0 = it.kbeg >> 6, ntap = a.KH * a.KW;
        tap = t0 % ntap;
        cc = (t0 / ntap) << 6;
        tapi = tap;
      }
      ky = tap / a.KW;
      kx = tap - ky * a.KW;
      const int hw = a.Hout * a.Wout;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int m = ((DEV && (a.dev & 16)) ? 0 : it.m0) + lr + 64 * i;   // (dev & 16: timing ablation, every tile gathers the first 256 pixels: A from L2)
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        apix[i] = b * a.Hin * a.Win;
        const int by = a.dgrad ? (oy + a.pad) : (oy * a.stride - a.pad);
        const int bx = a.dgrad ? (ox + a.pad) : (ox * a.stride - a.pad);
        ayx[i] = ((by + 256) << 16) | (bx + 256);
      }
      retap();
    }
    wbase = a.W + (int64_t)(it.n0 + lr) * a.ldw + it.kbeg + c8;
    wstep = 64 * a.ldw;
    if (ctap) {
      wrow = a.W + (int64_t)(it.n0 + lr) * a.ldw + c8;
      wbase = wrow + tapi * a.Cin + cc;
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);
  int iv = blockIdx.x, ikt = 0, ink = 0, islot = 0;
  bool ihave = false, idone = false;
  auto issue_prepare = [&]() -> bool {  // position the cursor on the next k-tile; false when none is left
    if (idone) return false;
    while (!ihave || ikt == ink) {
      if (ihave) iv += G;
      if (iv >= Wtot) {
        idone = true;
        return false;
      }
      const Item it = item_of(iv);
      ihave = true;
      ikt = 0;
      ink = it.nk;
      if (ink > 0) setup_issue(it);
    }
    return true;
  };
  auto park_issue = [&]() {  // past the last tile: the pieces read a zero page (keeps one instruction stream)
    if (WIDE) {
      asb = wsb = (const char*)zero;
      aoff = woff = 0u;
      pmask = 0u;
      return;
    }
    abase = zero; astep = 0;
    wbase = zero; wstep = 0;
    aok = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) ap[i] = zero;
  };
  auto slot_base = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane(lds0 + islot * STAGE + wave * 1024); };
  auto piece = [&](int i, unsigned sa) {
    if (WIDE) {
      if (i < AR) glds16_s(asb + (size_t)(i * (astepb & pmask)), aoff, sa + 512 * 16 * i);
      else glds16_s(wsb + (size_t)((i - AR) * (wstepb & pmask)), woff, sa + BM * 128 + 512 * 16 * (i - AR));
      return;
    }
    if (i < AR) {
      if (MODE == GEMM_ROW) {
        glds16(abase + i * astep, sa + 512 * 16 * i);
      } else {
        glds16(ap[i], sa + 512 * 16 * i);
        if (!ctap) ap[i] += ((aok >> i) & 1u) << 6;
      }
    } else {
      glds16(wbase + (i - AR) * wstep, sa + BM * 128 + 512 * 16 * (i - AR));
    }
  };
  auto issue_finish = [&]() {
    if (WIDE) {
      asb += pmask & 128u;
      wsb += pmask & 128u;
      islot ^= 1;
      ++ikt;
      return;
    }
    if (MODE == GEMM_ROW) {
      abase += astep ? 64 : 0;
      akpos += 64;
      if (a.A2 && akpos == a.K1 && astep) {   // the next tile is the first of the second segment (uniform)
        abase = a.A2 + (int64_t)arow * a.lda2 + c8;
        astep = 64 * a.lda2;
      }
    } else if (ctap) {   // next tile: the next tap of the same channel chunk, then the next chunk
      if (++tapi == a.KH * a.KW) {
        tapi = 0;
        cc += 64;
      }
      ky = tapi / a.KW;
      kx = tapi - ky * a.KW;
      retap();
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
    if (ctap) wbase = wstep ? wrow + tapi * a.Cin + cc : wbase;
    else wbase += wstep ? 64 : 0;
    islot ^= 1;
    ++ikt;
  };

  // ---- fragments ----
  f32x4 acc[NF][MF];
  bf16x8 af[2][MF], wq[4];
  auto lds_a = [&](int ks, int mf, int slot) -> bf16x8 {
    const int pc = (ks * 4 + g) ^ (j >> 1);
    return *(const bf16x8*)(smem + slot * STAGE + ((wm * 64 + mf * 16 + j) * 8 + pc) * 16);
  };
  auto lds_w = [&](int q, int slot) -> bf16x8 {  // q = ks * NF + nf
    const int ks = q / NF, nf = q - ks * NF;
    const int pc = (ks * 4 + g) ^ (j >> 1);
    return *(const bf16x8*)(smem + slot * STAGE + BM * 128 + ((wn * (BN / 2) + nf * 16 + j) * 8 + pc) * 16);
  };

  // prologue: tile 0 -> slot 0
  if (issue_prepare()) {
    const unsigned sa = slot_base();
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, sa);
    issue_finish();
  }

  int cslot = 0;
  for (int cv = blockIdx.x; cv < Wtot; cv += G) {
    const Item it = item_of(cv);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < it.nk; ++t) {
      // hand-over: this wave's pieces of tile k have landed and its reads of the other slot are done
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");  // (compiler) no LDS read of the new tile may be scheduled above the barrier
      // fragments of k-step 0 and the head of the W ring
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) af[0][mf] = lds_a(0, mf, cslot);
#pragma unroll
      for (int q = 0; q < 3; ++q) wq[q] = lds_w(q, cslot);
      const bool have = issue_prepare();
      if (!have) park_issue();
      const unsigned sa = slot_base();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ks = q / NF, nf = q % NF;
        if (q + 3 < NQ) wq[(q + 3) & 3] = lds_w(q + 3, cslot);
        if (q == 3) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) af[1][mf] = lds_a(1, mf, cslot);
        }
        FDMI_SETPRIO(1);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[q & 3], af[ks][mf], acc[nf][mf], 0, 0, 0);
        FDMI_SETPRIO(0);
        if (q < NP) {
          __builtin_amdgcn_sched_barrier(0);
          piece(q, sa);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (have) issue_finish();
      cslot ^= 1;
    }
    // the next item's first tile is landing meanwhile
    if constexpr (R32)
      tile_epilogue_r32<NF, MF, BN == 384>(a, it.m0 + wm * 64, it.n0 + wn * (BN / 2), it.z, acc, g, j);
    else if (!(DEV && (a.dev & 32)))   // (timing ablation of scripts/rowbench.py: the K loop alone)
      tile_epilogue<NF, MF, GEGLU ? 1 : 0, GN, true, MODE == GEMM_ROW>(a, it.m0 + wm * 64, it.n0 + wn * (BN / 2), it.z, acc, g, j);   // (whole tiles only)
    if constexpr (SKR)   // in-launch split-K reduction: this item's slab is stored -- hand off (gemm_tile.h)
      splitk_arrive(a, (it.m0 / BM) * tilesN + it.n0 / BN, it.m0, it.n0, smem + 2 * STAGE);
  }
  wait_vmcnt<0>();  // no LDS-DMA may still be in flight when the workgroup's LDS is released
  if constexpr (SKR) {   // the tiles whose last slab this block wrote: sum the slabs in slab order, run the real epilogue
    __syncthreads();
    const SkList* l = (const SkList*)(smem + 2 * STAGE);
    const int nred = l->n;
    if (nred > 0) {
      if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      for (int i = 0; i < nred; ++i) {
        const int mw = l->m0[i] + wm * 64, nw = l->n0[i] + wn * (BN / 2);
        splitk_sum_slabs<NF, MF, true>(a, mw, nw, acc, g, j);
        tile_epilogue<NF, MF, 0, false, true, MODE == GEMM_ROW, true>(a, mw, nw, 0, acc, g, j);
      }
    }
  }
}

template <int MODE, bool GEGLU, int BN, bool GN = false, bool DEV = false, bool SKR = false, bool R32 = false>
int launch4_t(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int smem = 2 * (256 + BN) * 128 + (SKR ? SK_LDS_BYTES : 0);   // (+ the split-K reducer's tile list behind the ring)
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)gemm4_kernel<MODE, GEGLU, BN, GN, DEV, SKR, R32>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int items = (a.M / 256) * (a.N / BN) * (a.splitk > 1 ? a.splitk : 1);
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
  if (prof) gemm_prof_shape(a);
  if (prof) fdmi_prof_begin(stream, (BN != 320 ? PROF_GEMM4_192 : PROF_GEMM4) + MODE, gemm_flops(a), gemm_bytes(a), (MODE == GEMM_ROW && BN == 320 && gemm_hbm_side(a)) ? PROF_GEMM4_ROW_HBM : -1);
  FDMI_KLAUNCH(prof, (gemm4_kernel<MODE, GEGLU, BN, GN, DEV, SKR, R32>), grid, dim3(512), smem, stream, a);
  if (prof) fdmi_prof_end(stream);
  FDMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool gemm4_eligible(const GemmArgs& a, int BN) {
  if (BN != 320 && BN != 192 && BN != 384) return false;
  if ((a.K & 63) != 0 || (a.M & 255) != 0 || (a.N % BN) != 0) return false;
  if (BN != 320 && a.act == ACT_GEGLU) return false;
  if (BN == 384) {   // the wide tile: ONE A operand, and what its only epilogue (tile_epilogue_r32 + a run-time tanh-GELU) serves
    if (a.mode != GEMM_ROW || a.A2 || (a.act != ACT_NONE && a.act != ACT_GELU_TANH)) return false;
    GemmArgs b = a;
    b.act = ACT_NONE;
    if (!gemm_r32_ok(b)) return false;
  }
  if (a.mode == GEMM_CONV && ((a.Cin & 63) != 0 || a.Hout > 2048 || a.Wout > 2048)) return false;
  if (a.act == ACT_GEGLU && (a.mode != GEMM_ROW || a.accum_atomic || fdmi_tune_get(9))) return false;
  if ((a.residual32 || a.C32) && BN == 320) return false;   // (the fp32 residual stream: the transformer denoisers' 256 x 192 / 256 x 384 tiles only)
  return true;
}
int launch_gemm4(const GemmArgs& a, hipStream_t stream, int BN) {
  FDMI_CHECK((BN == 320 || BN == 192 || BN == 384) && (a.M & 255) == 0 && (a.N % BN) == 0 && (a.K & 63) == 0, "gemm4: whole 256 x BN x 64 tiles only (its epilogue has no bounds checks)");
  if (BN == 384) {   // the wide tile has ONE instantiation (the fp32-residual-stream epilogue, which also serves plain problems and tanh-GELU)
    FDMI_CHECK(gemm4_eligible(a, 384) && !a.sk_tickets, "gemm4: the 256 x 384 tile takes row problems with one A operand, no activation but tanh-GELU, bf16 output");
    return launch4_t<GEMM_ROW, false, 384, false, false, false, true>(a, stream);
  }
  if (a.residual32 || a.C32) {   // the fp32 residual stream: its own instantiation
    FDMI_CHECK(BN == 192 && a.mode == GEMM_ROW && !a.sk_tickets, "gemm4: the fp32 residual stream runs on the 256 x 192 / 256 x 384 row kernels");
    return launch4_t<GEMM_ROW, false, 192, false, false, false, true>(a, stream);
  }
  if (a.sk_tickets) {   // in-launch split-K reduction: the SKR instantiations (launch_gemm sets the tickets only for plain problems)
    FDMI_CHECK(a.splitk > 1 && a.act != ACT_GEGLU && !a.gn_stats, "gemm4: in-launch split-K reduction needs a plain split problem");
    if (BN == 192) return a.mode == GEMM_ROW ? launch4_t<GEMM_ROW, false, 192, false, false, true>(a, stream)
                                             : launch4_t<GEMM_CONV, false, 192, false, false, true>(a, stream);
    return a.mode == GEMM_ROW ? launch4_t<GEMM_ROW, false, 320, false, false, true>(a, stream)
                              : launch4_t<GEMM_CONV, false, 320, false, false, true>(a, stream);
  }
  if (BN == 192) {
    FDMI_CHECK(a.act != ACT_GEGLU, "gemm4: the 256 x 192 tile has no GEGLU epilogue");
    return a.mode == GEMM_ROW ? launch4_t<GEMM_ROW, false, 192>(a, stream) : launch4_t<GEMM_CONV, false, 192>(a, stream);
  }
  // developer instantiations: only the plain 256 x 320 row / conv / GEGLU kernels, only when a developer bit asks for one
  const bool dev = (a.dev & (16 | 32 | 0x800)) != 0 || ((a.dev >> 16) & 7) > 1;
  if (a.act == ACT_GEGLU) {
    FDMI_CHECK(a.mode == GEMM_ROW && a.splitk <= 1 && !a.accum_atomic, "gemm4: GEGLU needs a plain row GEMM");
    if (dev) return launch4_t<GEMM_ROW, true, 320, false, true>(a, stream);
    return launch4_t<GEMM_ROW, true, 320>(a, stream);
  }
  if (dev && !a.gn_stats)
    return a.mode == GEMM_ROW ? launch4_t<GEMM_ROW, false, 320, false, true>(a, stream) : launch4_t<GEMM_CONV, false, 320, false, true>(a, stream);
  if (a.gn_stats)
    return a.mode == GEMM_ROW ? launch4_t<GEMM_ROW, false, 320, true>(a, stream) : launch4_t<GEMM_CONV, false, 320, true>(a, stream);
  return a.mode == GEMM_ROW ? launch4_t<GEMM_ROW, false, 320>(a, stream) : launch4_t<GEMM_CONV, false, 320>(a, stream);
}
