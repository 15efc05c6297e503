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
#include "gemm_tile.h"

namespace {

// LDS-DMA issued from inline asm: hipcc then keeps no scoreboard entry for it, so the ONLY waits on
// these loads are the counted ones placed by hand below (with the builtin, the waitcnt pass drained
// the ring with vmcnt(0) at every loop back-edge of the persistent loop).  M0 (the LDS destination
// base) is written and restored inside the statement; lds_addr is wave-uniform.
__device__ __forceinline__ void glds16m(const void* gptr, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(lds_addr)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// GN: the epilogue also accumulates the consumer's GroupNorm statistics (GemmArgs::gn_stats); separate instantiations
// SKR: in-launch split-K reduction (GemmArgs::sk_tickets, gemm_tile.h::splitk_last_arriver), its own instantiation (see gemm4.hip)
// R32: the fp32-residual-stream epilogue (GemmArgs::residual32 / C32, gemm_tile.h::tile_epilogue_r32), its own instantiation
template <int BN, int MODE, bool GN = false, bool SKR = false, bool R32 = false>
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
  if constexpr (SKR) {
    if (tid == 0) ((volatile SkList*)(smem + 3 * (BM + BN) * 128))->n = 0;   // (ordered before its first use by the K loops' barriers)
  }

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
  int akpos = 0, am0 = 0;   // ROW with a second A segment: k of the tile to issue next, the item's first row
  auto setup_issue = [&](const Item& it) {
    akpos = it.kbeg;
    am0 = it.m0;
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
        if (a.A2 && it.kbeg >= a.K1) ap[i] = aval[i] ? a.A2 + (int64_t)m * a.lda2 + (it.kbeg - a.K1) + c8 : zero;
        else ap[i] = aval[i] ? a.A + (int64_t)m * a.lda + it.kbeg + c8 : zero;
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
      glds16m(ap[i], sa + 512 * 16 * i);
      ap[i] += ainc[i];
    }
#pragma unroll
    for (int i = 0; i < WFULL; ++i) {
      glds16m(wp[i], sa + BM * 128 + 512 * 16 * i);
      wp[i] += winc[i];
    }
    if (WODD) {
      if (wave < 4) {
        glds16m(wp[WR - 1], sa + BM * 128 + 512 * 16 * (WR - 1));
        wp[WR - 1] += winc[WR - 1];
      }
    }
    if (MODE == GEMM_ROW) {
      akpos += 64;
      if (a.A2 && akpos == a.K1) {   // the next tile is the first of the second A segment (GemmArgs::A2; uniform)
#pragma unroll
        for (int i = 0; i < AR; ++i)
          if (ainc[i]) ap[i] = a.A2 + (int64_t)(am0 + lr + 64 * i) * a.lda2 + c8;
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
    FDMI_SETPRIO(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nf], af[ks][mf], acc[nf][mf], 0, 0, 0);
    FDMI_SETPRIO(0);
  };

  auto epilogue = [&](const Item& it) {
    if constexpr (R32) tile_epilogue_r32<NF, MF>(a, it.m0 + wm * 64, it.n0 + wn * (BN / 2), it.z, acc, g, j);
    else tile_epilogue<NF, MF, 2, GN, false, MODE == GEMM_ROW>(a, it.m0 + wm * 64, it.n0 + wn * (BN / 2), it.z, acc, g, j);
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
          if (wave < 4) wait_vm<AR + WR>(); else wait_vm<AR + WFULL>();
        } else {
          wait_vm<AR + WFULL>();
        }
      } else {
        wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (issue_next()) ++inflight;   // (DMA issue first: the asm's memory clobber would drain pending ds_reads)
      load_frags(cslot);
      mma();
      cslot = cslot == 2 ? 0 : cslot + 1;
      --inflight;
    }
    epilogue(it);
    if constexpr (SKR)   // in-launch split-K reduction: this item's slab is stored -- hand off (gemm_tile.h)
      splitk_arrive(a, (it.m0 / BM) * tilesN + it.n0 / BN, it.m0, it.n0, smem + 3 * STAGE);
    // make the waitcnt pass see an empty VM scoreboard at the back-edge: otherwise it protects the
    // epilogue's pending loads/stores with a vmcnt(0) inside every K iteration (draining the ring)
    __builtin_amdgcn_s_waitcnt(0);
  }
  if constexpr (SKR) {   // the tiles whose last slab this block wrote: sum the slabs in slab order, run the real epilogue
    __syncthreads();
    const SkList* l = (const SkList*)(smem + 3 * STAGE);
    const int nred = l->n;
    if (nred > 0) {
      if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      for (int i = 0; i < nred; ++i) {
        const int mw = l->m0[i] + wm * 64, nw = l->n0[i] + wn * (BN / 2);
        splitk_sum_slabs<NF, MF, false>(a, mw, nw, acc, g, j);
        tile_epilogue<NF, MF, 2, false, false, MODE == GEMM_ROW, true>(a, mw, nw, 0, acc, g, j);
      }
    }
  }
}

template <int BN, int MODE, bool GN = false, bool SKR = false, bool R32 = false>
int launch3_t(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int smem = 3 * (256 + BN) * 128 + (SKR ? SK_LDS_BYTES : 0);   // (+ the split-K reducer's tile list behind the ring)
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)gemm3_kernel<BN, MODE, GN, SKR, R32>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
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
  if (prof) gemm_prof_shape(a);
  if (prof) fdmi_prof_begin(stream, PROF_GEMM3 + MODE * 2 + (BN == 160 ? 0 : 1), gemm_flops(a), gemm_bytes(a), (MODE == GEMM_ROW && gemm_hbm_side(a)) ? PROF_GEMM3_ROW_HBM : -1);
  FDMI_KLAUNCH(prof, (gemm3_kernel<BN, MODE, GN, SKR, R32>), grid, dim3(512), smem, stream, a);
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
  if (a.residual32 || a.C32) {   // the fp32 residual stream: its own instantiation (row GEMMs only: gemm_r32_ok)
    FDMI_CHECK(a.mode == GEMM_ROW && !a.sk_tickets, "gemm3: the fp32 residual stream needs a row GEMM");
    return BN == 160 ? launch3_t<160, GEMM_ROW, false, false, true>(a, stream) : launch3_t<128, GEMM_ROW, false, false, true>(a, stream);
  }
  if (a.sk_tickets) {   // in-launch split-K reduction: the SKR instantiations
    FDMI_CHECK(a.splitk > 1 && !a.gn_stats, "gemm3: in-launch split-K reduction needs a split problem without GroupNorm sums");
    if (a.mode == GEMM_ROW) return BN == 160 ? launch3_t<160, GEMM_ROW, false, true>(a, stream) : launch3_t<128, GEMM_ROW, false, true>(a, stream);
    return BN == 160 ? launch3_t<160, GEMM_CONV, false, true>(a, stream) : launch3_t<128, GEMM_CONV, false, true>(a, stream);
  }
  if (a.gn_stats) {
    if (a.mode == GEMM_ROW) return BN == 160 ? launch3_t<160, GEMM_ROW, true>(a, stream) : launch3_t<128, GEMM_ROW, true>(a, stream);
    return BN == 160 ? launch3_t<160, GEMM_CONV, true>(a, stream) : launch3_t<128, GEMM_CONV, true>(a, stream);
  }
  if (a.mode == GEMM_ROW) return BN == 160 ? launch3_t<160, GEMM_ROW>(a, stream) : launch3_t<128, GEMM_ROW>(a, stream);
  return BN == 160 ? launch3_t<160, GEMM_CONV>(a, stream) : launch3_t<128, GEMM_CONV>(a, stream);
}
