// TN product for the LoRA weight gradients:  C[n1][n2] += sum_m X[m][n1] * Y[m][n2]   (fp32 atomics into the flat gradient)
//
// The LoRA gradients contract over the ROWS of two row-major activations (dB = dY^T t, dA = dt^T x): both operands have the
// reduction index slow.  The NT kernels (gemm*.hip) want it fast, so the plan used to materialise four transposed copies per
// LoRA linear (transpose2d: 449 launches and ~10 GB of HBM traffic per C2 step at 1.8 TB/s, round-1 profile) before two
// split-K GEMMs.  This kernel reads X and Y as they are: a 64-row slab of each is loaded with coalesced 16-byte reads, transposed
// in registers in 8 x 8 blocks and written as whole 16-byte chunks into exactly the [index][64 k] LDS image with the
// 16-byte-chunk XOR swizzle that gemm.hip's fragment reads expect; the MFMA loop is that kernel's.
//
// grid (N2 tiles of 128, N1 tiles of 64, row splits); 256 threads = 4 waves as 2 (n1) x 2 (n2), wave tile 32 x 64.
#include "ops.h"
#include "gemm_tile.h"   // glds16 (LDS-DMA from inline asm), wait_vmcnt, g_zero16b
#include "wgrad.h"

namespace {

constexpr int TN_BN1 = 64, TN_BN2 = 128, TN_BK = 64;

struct TnArgs {
  const bf16_t* X; int64_t ldx;
  const bf16_t* Y; int64_t ldy;
  int64_t M; int N1, N2;
  float* C; int64_t ldc;
  int rows_per_split;   // multiple of TN_BK
  int ct;               // 1: the result is stored transposed, C[n2][n1] (the launcher swapped the operands: see launch_wgrad_tn)
};

// byte offset of element (row, k) of an [rows][64] bf16 image: 16-byte chunk k>>3 stored at slot (k>>3) ^ ((row>>1)&7)
__device__ __forceinline__ int img_off(int row, int k) { return ((row * 8 + ((k >> 3) ^ ((row >> 1) & 7))) << 4) + ((k & 7) << 1); }

__global__ __launch_bounds__(256) void wgrad_tn_kernel(TnArgs a) {
  __shared__ __attribute__((aligned(16))) char sx[TN_BN1 * 128];
  __shared__ __attribute__((aligned(16))) char sy[TN_BN2 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w1 = wave >> 1, w2 = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  const int n2_0 = blockIdx.x * TN_BN2, n1_0 = blockIdx.y * TN_BN1;
  const int64_t m_beg = (int64_t)blockIdx.z * a.rows_per_split;
  const int64_t m_end = min(a.M, m_beg + a.rows_per_split);
  if (m_beg >= m_end) return;
  const int nk = (int)((m_end - m_beg + TN_BK - 1) / TN_BK);

  // staging (round 4): a thread owns ONE 8 (rows m) x 8 (columns n) block of the slab -- wave 0 the 64 blocks of X's 64 x 64
  // slab, waves 1 / 2 the two 64-column halves of Y's 64 x 128 slab (wave 3 only multiplies; with several blocks per CU the
  // SIMDs even out).  It loads the block's 8 rows with 16-byte reads (a wave instruction = 8 rows x 128 B), transposes the 8 x 8
  // 16-bit elements in registers (two ALU ops per output dword) and writes 8 whole 16-byte chunks of the image -- chunk
  // (index n, k-chunk br) holds k = 8 br .. 8 br + 7 of index n -- instead of 64 2-byte scatter writes whose bank conflicts
  // bound the kernel (C4: 586 launches x 92 us per step at 0.8 TB/s).  Lane -> (br = lane & 7, column block lane >> 3): the 16
  // lanes of an LDS write group then cover two image rows with all eight swizzled slots each, i.e. 2 x 128 B without conflicts.
  const int sw = wave;
  const int br = lane & 7, bc = (sw == 2 ? 8 : 0) + (lane >> 3);
  const bf16_t* sbase = sw == 0 ? a.X : a.Y;
  const int64_t sld = sw == 0 ? a.ldx : a.ldy;
  const int sn0 = (sw == 0 ? n1_0 : n2_0) + bc * 8;
  const bool scol_ok = sw < 3 && sn0 < (sw == 0 ? a.N1 : a.N2);   // (widths are multiples of 8: a column block is in or out as a whole)
  char* const simg = sw == 0 ? sx : sy;
  uint4 rr[8];
  auto load = [&](int kt) {
    const int64_t m0 = m_beg + (int64_t)kt * TN_BK + br * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      rr[i] = (scol_ok && m0 + i < m_end) ? *(const uint4*)(sbase + (m0 + i) * sld + sn0) : make_uint4(0, 0, 0, 0);
  };
  auto store = [&]() {
    if (sw == 3) return;
    uint32_t w[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[i][0] = rr[i].x; w[i][1] = rr[i].y; w[i][2] = rr[i].z; w[i][3] = rr[i].w; }
#pragma unroll
    for (int e = 0; e < 8; ++e) {            // column e of the block -> image row n, its 8 k values = rows 0 .. 7 of the block
      uint32_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t lo = w[2 * q][e >> 1], hi = w[2 * q + 1][e >> 1];
        o[q] = (e & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
      }
      const int n = bc * 8 + e;
      *(uint4*)(simg + ((n * 8 + (br ^ ((n >> 1) & 7))) << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int f1 = 0; f1 < 2; ++f1)
#pragma unroll
    for (int f2 = 0; f2 < 4; ++f2) acc[f1][f2] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();            // every wave is done reading the previous slab
    store();
    __syncthreads();
    if (kt + 1 < nk) load(kt + 1);   // in flight during the MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pc = (ks * 4 + g) ^ (j >> 1);
      bf16x8 xf[2], yf[4];
#pragma unroll
      for (int f = 0; f < 2; ++f) xf[f] = *(const bf16x8*)(sx + ((w1 * 32 + f * 16 + j) * 8 + pc) * 16);
#pragma unroll
      for (int f = 0; f < 4; ++f) yf[f] = *(const bf16x8*)(sy + ((w2 * 64 + f * 16 + j) * 8 + pc) * 16);
#pragma unroll
      for (int f1 = 0; f1 < 2; ++f1)
#pragma unroll
        for (int f2 = 0; f2 < 4; ++f2)
          acc[f1][f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[f1], yf[f2], acc[f1][f2], 0, 0, 0);
    }
  }
  // lane (g, j): first-operand index (n1) g*4 + r, second-operand index (n2) j of each 16 x 16 fragment
#pragma unroll
  for (int f1 = 0; f1 < 2; ++f1)
#pragma unroll
    for (int f2 = 0; f2 < 4; ++f2) {
      const int n2 = n2_0 + w2 * 64 + f2 * 16 + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n1 = n1_0 + w1 * 32 + f1 * 16 + g * 4 + r;
        if (n1 < a.N1 && n2 < a.N2) atomicAdd(a.ct ? a.C + (int64_t)n2 * a.ldc + n1 : a.C + (int64_t)n1 * a.ldc + n2, acc[f1][f2][r]);
      }
    }
}


// ---- round 5: the same product as a streaming TN GEMM whose staging does no arithmetic at all -------------------------------------
// Both operands are copied AS THEY LIE (row-major, the reduction index m slow) into an LDS ring by LDS-DMA
// (`global_load_lds_dwordx4`, gemm_tile.h::glds16: no registers, no ds_write pass); the transposition the MFMA operand layout
// needs is done by the LDS read itself -- gfx950's `ds_read_b64_tr_b16` hands lane c of a 16-lane group column c of a
// [4 rows][16 columns] block, i.e. four consecutive reduction indices of ONE output index, which is half of a 16x16x32 fragment.
// The old kernel (above) loads through registers, transposes 8 x 8 blocks with ALU ops and writes the NT kernels' image with
// two block-wide barriers per 64-row slab: 27.8 us per launch on the C2 step (2.1 TB/s) and 92 us on C4's (0.8 TB/s).
//
// Tile: BN1 (64 | 128: the FIRST operand, the launcher puts the narrow one = the LoRA rank there) x 128 columns of the second
// operand x 64 rows per ring stage; 256 threads = 4 waves as W1 x W2 (BN1 = 128: 2 x 2, wave tile 64 x 64; BN1 = 64: 1 x 4,
// 64 x 32).  grid (column tiles of the second operand, column tiles of the first, row splits); every block streams its rows once
// through an NSTAGE-deep ring (one raw s_barrier + one counted s_waitcnt per stage) and adds its partial tile with fp32 atomics.
//
// LDS image of a stage: the operand's 64 rows, row r at byte r * RB (RB = 2 * columns = 128 | 256), its 32-byte granules
// (16 columns = one fragment's width) XOR-swizzled: granule q of row r sits at slot q ^ f(r), f(r) = (r * RB / 256) mod (RB / 32).
// The swizzle is applied on the SOURCE side (a lane of a DMA piece picks which 16 bytes of the row it fetches; the LDS side of
// a piece is 1 KB lane-linear).  A `ds_read_b64_tr_b16` is serviced in two 32-lane halves; the reduction-index assignment below
// makes a half touch 8 CONSECUTIVE rows of one granule column, which the swizzle spreads over all 64 banks: conflict-free.
//
// Reduction-index assignment (free, as long as both operands use the same one): MFMA k-slot 8 g + 4 h + j of k-step ks (lane group
// g, tr-read h in {0, 1}, element j) is row 32 ks + 16 h + 4 g + j of the stage.
template <int BN1>
struct Tn2 {
  static constexpr int BN2 = 128, BK = 64, NSTAGE = BN1 == 64 ? 3 : 4;   // 72 KB (two blocks per CU) | 128 KB
  static constexpr int RB1 = BN1 * 2, RB2 = BN2 * 2;
  static constexpr int XBYTES = BK * RB1, YBYTES = BK * RB2, STAGE = XBYTES + YBYTES;
  static constexpr int XP = XBYTES / 1024 / 4, YP = YBYTES / 1024 / 4, PPW = XP + YP;   // DMA pieces per wave per stage
  static constexpr int W1 = BN1 / 64, W2 = 4 / W1;
  static constexpr int F1 = 4, F2 = BN2 / W2 / 16;                                       // fragments per wave
  static constexpr int SMEM = NSTAGE * STAGE;
};
typedef __attribute__((ext_vector_type(4))) short s16x4;

// one block's work: tile (bx = column tile of the second operand, by = of the first), row split bz of problem `a`
template <int BN1, bool CT>
__device__ __forceinline__ void tn2_block(const TnArgs& a, const int bx, const int by, const int bz) {
  using T = Tn2<BN1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w1 = wave / T::W2, w2 = wave % T::W2;
  const int g = lane >> 4, c = lane & 15;
  const int n2_0 = bx * T::BN2, n1_0 = by * BN1;
  const int64_t m_beg = (int64_t)bz * a.rows_per_split;
  const int64_t m_end = min(a.M, m_beg + a.rows_per_split);
  if (m_beg >= m_end) return;
  const int nk = (int)((m_end - m_beg + T::BK - 1) / T::BK);
  const bf16_t* zero = (const bf16_t*)g_zero16b;
  const unsigned lds0 = (unsigned)(uintptr_t)((LDS_AS char*)smem);

  // ---- DMA bookkeeping: piece i of this wave covers rows (wave + 4 i) * RPP .. of the operand's stage image; lane t fetches the
  // 16 bytes that belong at (row, slot) = (t / SLOTS, t % SLOTS) of the piece ----
  auto src_chunk = [](int r, int slot, int RB) {   // logical 16-byte chunk of row r stored at `slot`
    const int f = RB == 256 ? (r & 7) : ((r >> 1) & 3);
    return (((slot >> 1) ^ f) << 1) | (slot & 1);
  };
  constexpr int S1 = T::RB1 / 16, S2 = T::RB2 / 16;          // slots per row
  constexpr int RPP1 = 64 / S1, RPP2 = 64 / S2;              // rows per piece
  int xr[T::XP], xc[T::XP], yr[T::YP], yc[T::YP];            // row inside the stage, first column (global) of the lane's chunk
#pragma unroll
  for (int i = 0; i < T::XP; ++i) {
    xr[i] = (wave + 4 * i) * RPP1 + lane / S1;
    xc[i] = n1_0 + src_chunk(xr[i], lane % S1, T::RB1) * 8;
  }
#pragma unroll
  for (int i = 0; i < T::YP; ++i) {
    yr[i] = (wave + 4 * i) * RPP2 + lane / S2;
    yc[i] = n2_0 + src_chunk(yr[i], lane % S2, T::RB2) * 8;
  }
  auto issue = [&](int kt, int slot) {   // every call issues exactly PPW pieces (real data, or the zero page past the last row / column)
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + slot * T::STAGE + wave * 1024);
    const int64_t m0 = m_beg + (int64_t)kt * T::BK;
#pragma unroll
    for (int i = 0; i < T::XP; ++i) {
      const int64_t m = m0 + xr[i];
      const bool ok = kt < nk && m < m_end && xc[i] < a.N1;
      glds16(ok ? a.X + m * a.ldx + xc[i] : zero, sb + 4096 * i);
    }
#pragma unroll
    for (int i = 0; i < T::YP; ++i) {
      const int64_t m = m0 + yr[i];
      const bool ok = kt < nk && m < m_end && yc[i] < a.N2;
      glds16(ok ? a.Y + m * a.ldy + yc[i] : zero, sb + T::XBYTES + 4096 * i);
    }
  };

  // ---- fragment addressing: lane (g, c) supplies the 8 bytes at row 4 g + (c >> 2) (+ 32 ks + 16 h), granule nf, quarter c & 3 ----
  const int br = 4 * g + (c >> 2);
  const int fl1 = T::RB1 == 256 ? (br & 7) : ((br >> 1) & 3), fl2 = br & 7;   // (RB2 = 256)
  const int xoff = br * T::RB1 + (c & 3) * 8, yoff = T::XBYTES + br * T::RB2 + (c & 3) * 8;
  auto frag = [&](const char* st, int off, int RB, int nf, int fl, int ks) -> bf16x8 {
    const char* p = st + off + (32 * ks) * RB + ((nf ^ fl) << 5);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + 16 * RB));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  f32x4 acc[T::F1][T::F2];
#pragma unroll
  for (int f1 = 0; f1 < T::F1; ++f1)
#pragma unroll
    for (int f2 = 0; f2 < T::F2; ++f2) acc[f1][f2] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < T::NSTAGE - 1; ++s) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed for this wave (NSTAGE - 2 younger stages may still be in flight) and this wave's reads of stage kt - 1
    // are done; behind the barrier that holds for every wave: stage kt is readable, slot (kt - 1) % NSTAGE is free
    wait_vmcnt<(T::NSTAGE - 2) * T::PPW>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + T::NSTAGE - 1, (kt + T::NSTAGE - 1) % T::NSTAGE);
    const char* st = smem + (kt % T::NSTAGE) * T::STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[T::F1], yf[T::F2];
#pragma unroll
      for (int f = 0; f < T::F1; ++f) xf[f] = frag(st, xoff, T::RB1, w1 * T::F1 + f, fl1, ks);
#pragma unroll
      for (int f = 0; f < T::F2; ++f) yf[f] = frag(st, yoff, T::RB2, w2 * T::F2 + f, fl2, ks);
#pragma unroll
      for (int f1 = 0; f1 < T::F1; ++f1)
#pragma unroll
        for (int f2 = 0; f2 < T::F2; ++f2)
          acc[f1][f2] = CT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[f2], xf[f1], acc[f1][f2], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[f1], yf[f2], acc[f1][f2], 0, 0, 0);
    }
  }
  wait_vmcnt<0>();   // no LDS-DMA may be in flight when the workgroup's LDS is released
  // lane (g, c) of a fragment: the MFMA's first operand's index g * 4 + r, the second's c.  CT = false: first = X (n1), the
  // result C[n1][n2] -- lanes c run along n2, contiguous.  CT = true (the launcher swapped the operands: the result is
  // C[n2][n1] with ldc): the MFMA's first operand is Y, lanes c run along n1 -- contiguous again (round 4's swap stored with a
  // stride and lost more than the fuller tiles won)
#pragma unroll
  for (int f1 = 0; f1 < T::F1; ++f1)
#pragma unroll
    for (int f2 = 0; f2 < T::F2; ++f2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n1 = n1_0 + (w1 * T::F1 + f1) * 16 + (CT ? c : g * 4 + r);
        const int n2 = n2_0 + (w2 * T::F2 + f2) * 16 + (CT ? g * 4 + r : c);
        if (n1 < a.N1 && n2 < a.N2) atomicAdd(CT ? a.C + (int64_t)n2 * a.ldc + n1 : a.C + (int64_t)n1 * a.ldc + n2, acc[f1][f2][r]);
      }
    }
}

template <int BN1, bool CT>
__global__ __launch_bounds__(256) void wgrad_tn2_kernel(TnArgs a) {
  tn2_block<BN1, CT>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- several products in ONE launch (round 5, the "grouped LoRA launches" of VERDICT r4): the weight gradients of one LoRA-carrying
// linear (dB, dA) or of a fused q/k/v projection (3 x (dB, dA)) are independent, ~20 us each, and each alone cannot both fill the
// chip and keep its atomics small (every row split adds N1 x N2 fp32 atomics: 86 splits to reach 256 blocks at M = 65536, 320 x
// 128).  As one launch the group fills the chip TOGETHER: the row split is chosen for the group (equal rows per block across the
// problems, about one block per CU in total), so each problem runs with several times fewer splits.  blockIdx.z runs over the
// problems' row splits back to back (zend = exclusive prefix ends); the grid's x / y are the largest tile counts of the group and
// a block outside its problem's tile range leaves at once.  The problems share BN1 (the narrow operand's tile); the operand
// order of the MFMA / the store (`ct`) is per problem (a uniform branch over two instantiations of the block body).
constexpr int TN_GROUP_MAX = 6;
struct TnGroup {
  TnArgs p[TN_GROUP_MAX];
  int zend[TN_GROUP_MAX], t1[TN_GROUP_MAX], t2[TN_GROUP_MAX];
  int n;
};

template <int BN1>
__global__ __launch_bounds__(256) void wgrad_tn2_group_kernel(TnGroup gr) {
  const int z = blockIdx.z;
  int p = 0;
#pragma unroll
  for (int i = 0; i + 1 < TN_GROUP_MAX; ++i)
    if (i + 1 < gr.n && z >= gr.zend[i]) p = i + 1;
  p = __builtin_amdgcn_readfirstlane(p);
  const int zb = z - (p ? gr.zend[p - 1] : 0);
  if ((int)blockIdx.x >= gr.t2[p] || (int)blockIdx.y >= gr.t1[p]) return;
  const TnArgs& a = gr.p[p];
  if (a.ct) tn2_block<BN1, true>(a, blockIdx.x, blockIdx.y, zb);
  else tn2_block<BN1, false>(a, blockIdx.x, blockIdx.y, zb);
}

template <int BN1>
int launch_tn2_group(const TnGroup& gr, dim3 grid, double flops, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)wgrad_tn2_group_kernel<BN1>, hipFuncAttributeMaxDynamicSharedMemorySize, Tn2<BN1>::SMEM));
    attr_set = true;
  }
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_begin(st, PROF_WGRAD_TN, flops);
  FDMI_KLAUNCH(prof, (wgrad_tn2_group_kernel<BN1>), grid, dim3(256), Tn2<BN1>::SMEM, st, gr);
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 0;
}

template <int BN1, bool CT>
int launch_tn2(const TnArgs& a, dim3 grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    FDMI_HIP(hipFuncSetAttribute((const void*)wgrad_tn2_kernel<BN1, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, Tn2<BN1>::SMEM));
    attr_set = true;
  }
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_begin(st, PROF_WGRAD_TN, 2.0 * (double)a.M * a.N1 * a.N2);
  FDMI_KLAUNCH(prof, (wgrad_tn2_kernel<BN1, CT>), grid, dim3(256), Tn2<BN1>::SMEM, st, a);
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_wgrad_tn(const bf16_t* X, int64_t ldx, const bf16_t* Y, int64_t ldy, int64_t M, int N1, int N2, float* C,
                    int64_t ldc, hipStream_t st) {
  FDMI_CHECK(X && Y && C && M > 0 && N1 > 0 && N2 > 0, "wgrad_tn: empty problem / null operand");
  FDMI_CHECK((N1 % 8) == 0 && (N2 % 8) == 0 && (ldx % 8) == 0 && (ldy % 8) == 0, "wgrad_tn: widths and leading dims must be multiples of 8");
  FDMI_CHECK(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0, "wgrad_tn: operands must be 16-B aligned");
  if (!fdmi_tune_get(46)) {   // round 5: the streaming kernel (knob 46 = 1: the register-transposing kernel above, the A/B side)
    // the narrow operand (the LoRA rank) first: its tile then covers all of its columns, the wide operand is read exactly once
    int ct = 0;
    if (N2 < N1) {
      const bf16_t* tp = X; X = Y; Y = tp;
      const int64_t tl = ldx; ldx = ldy; ldy = tl;
      const int tn = N1; N1 = N2; N2 = tn;
      ct = 1;
    }
    const int bn1 = N1 <= 64 ? 64 : 128;
    const int t1 = cdiv(N1, bn1), t2 = cdiv(N2, 128);
    const int64_t slabs = (M + 63) / 64;
    // row splits: about one block per CU, but at least 512 rows (8 ring stages) per block -- every further split costs another
    // N1 x N2 floats of atomics, and those bound the kernel before the streaming does (measured, profiles/r5_wgrad_tn_rates.txt:
    // M = 16384, 640 x 128: 64 splits of 256 rows 25.3 us, 26 splits of 640 rows 15.7 us; M = 65536, 320 x 128: 171 / 86 / 43
    // splits 39.7 / 25.5 / 24.7 us; the wide products, 36 column tiles at M = 32768: 4 splits 78 us, 15 splits 66 us)
    const int64_t tiles = (int64_t)t1 * t2;
    int64_t want = fdmi_tune_get(45) > 0 ? fdmi_tune_get(45) : 256;
    if (fdmi_det()) want = 1;   // deterministic mode: ONE row split per tile (a tile's atomics then have a single contributor)
    // the 64-wide kernel fits two blocks per CU: 512 blocks when each still streams >= 2048 rows (the widest products: M = 32768,
    // 4608 x 64: 15 splits 66.7 us, 8 splits 78.9 us; at 1152 x 64 it is the other way round: 57 splits 28.3 us, 29 splits 23.5 us)
    if (!fdmi_tune_get(45) && !fdmi_det() && bn1 == 64 && slabs * 64 / ((512 + tiles - 1) / tiles) >= 2048) want = 512;
    int64_t splits = (want + tiles - 1) / tiles;
    if (splits > (slabs + 7) / 8) splits = (slabs + 7) / 8;
    if (splits < 1) splits = 1;
    TnArgs a{X, ldx, Y, ldy, M, N1, N2, C, ldc, (int)(((slabs + splits - 1) / splits) * 64), ct};
    const int nz = (int)((M + a.rows_per_split - 1) / a.rows_per_split);
    const dim3 grid(t2, t1, nz);
    if (bn1 == 64) return ct ? launch_tn2<64, true>(a, grid, st) : launch_tn2<64, false>(a, grid, st);
    return ct ? launch_tn2<128, true>(a, grid, st) : launch_tn2<128, false>(a, grid, st);
  }
  int ct = 0;
  // The tile is 64 (first operand) x 128 (second).  A product with a narrow SECOND operand (dB = dY^T t: N2 = a LoRA rank of 64)
  // leaves half of every tile empty; running it with the operands swapped (t^T dY, result stored transposed) fills the tiles but
  // turns the epilogue's atomics into a strided pattern, and that costs more than the empty half: 115 vs 42 us at M = 32768,
  // N1 = 1152, N2 = 64 (profiles/r4_wgrad_tn_rates.txt).  The swap therefore stays a developer switch (knob 43 = 1), off by default.
  if (N2 <= TN_BN1 && N1 >= TN_BN2 && fdmi_tune_get(43)) {
    const bf16_t* tp = X; X = Y; Y = tp;
    const int64_t tl = ldx; ldx = ldy; ldy = tl;
    const int tn = N1; N1 = N2; N2 = tn;
    ct = 1;
  }
  const int t1 = cdiv(N1, TN_BN1), t2 = cdiv(N2, TN_BN2);
  const int64_t slabs = (M + TN_BK - 1) / TN_BK;
  // ~4 blocks per CU: enough row splits to fill the chip, at least 4 slabs each so the atomics stay a small tail
  const int64_t want = fdmi_det() ? 1 : fdmi_tune_get(45) > 0 ? fdmi_tune_get(45) : 1024;   // blocks to aim for (developer knob 45: scripts/wgrad_rates.py; deterministic mode: one split)
  int64_t splits = (want + (int64_t)t1 * t2 - 1) / ((int64_t)t1 * t2);   // (512 ... 2048 blocks measure the same on C2; 256 is slower)
  if (splits > (slabs + 3) / 4) splits = (slabs + 3) / 4;
  if (splits < 1) splits = 1;
  TnArgs a{X, ldx, Y, ldy, M, N1, N2, C, ldc, (int)(((slabs + splits - 1) / splits) * TN_BK), ct};
  const int nz = (int)((M + a.rows_per_split - 1) / a.rows_per_split);
  const bool prof = fdmi_prof_on();
  if (prof) fdmi_prof_begin(st, PROF_WGRAD_TN, 2.0 * (double)M * N1 * N2);
  FDMI_KLAUNCH(prof, wgrad_tn_kernel, dim3(t2, t1, nz), dim3(256), 0, st, a);
  if (prof) fdmi_prof_end(st);
  FDMI_HIP(hipGetLastError());
  return 0;
}

int launch_wgrad_tn_group(const WgradProblem* pr, int n, hipStream_t st) {
  FDMI_CHECK(pr && n >= 1 && n <= WGRAD_GROUP_MAX, "wgrad_tn_group: 1 ... 6 problems");
  static_assert(WGRAD_GROUP_MAX == TN_GROUP_MAX, "group sizes");
  bool group = n > 1 && !fdmi_tune_get(46) && fdmi_tune_get(47) != 1;
  TnGroup gr{};
  int bn1 = 0;
  double flops = 0;
  for (int i = 0; i < n && group; ++i) {
    const WgradProblem& q = pr[i];
    FDMI_CHECK(q.X && q.Y && q.C && q.M > 0 && q.N1 > 0 && q.N2 > 0, "wgrad_tn: empty problem / null operand");
    FDMI_CHECK((q.N1 % 8) == 0 && (q.N2 % 8) == 0 && (q.ldx % 8) == 0 && (q.ldy % 8) == 0, "wgrad_tn: widths and leading dims must be multiples of 8");
    FDMI_CHECK(((uintptr_t)q.X % 16) == 0 && ((uintptr_t)q.Y % 16) == 0, "wgrad_tn: operands must be 16-B aligned");
    TnArgs& a = gr.p[i];
    if (q.N2 < q.N1) a = TnArgs{q.Y, q.ldy, q.X, q.ldx, q.M, q.N2, q.N1, q.C, q.ldc, 0, 1};   // the narrow operand first (launch_wgrad_tn)
    else a = TnArgs{q.X, q.ldx, q.Y, q.ldy, q.M, q.N1, q.N2, q.C, q.ldc, 0, 0};
    const int b = a.N1 <= 64 ? 64 : 128;
    if (i == 0) bn1 = b;
    else if (b != bn1) group = false;
    gr.t1[i] = cdiv(a.N1, b);
    gr.t2[i] = cdiv(a.N2, 128);
    flops += 2.0 * (double)q.M * q.N1 * q.N2;
  }
  if (group) {
    // products of very different widths do not share a launch well: (dB, dA) of PixArt's feed-forward projections -- 36 against 9
    // column tiles -- measured 105 us as a group against 94.5 us one by one, while equal or nearly equal widths gain 1.05 - 2.1x
    // (profiles/r5_wgrad_group_rates.txt)
    int64_t tmin = INT64_MAX, tmax = 0;
    for (int i = 0; i < n; ++i) {
      const int64_t t = (int64_t)gr.t1[i] * gr.t2[i];
      tmin = t < tmin ? t : tmin;
      tmax = t > tmax ? t : tmax;
    }
    if (tmax >= 4 * tmin) group = false;
  }
  if (!group) {
    for (int i = 0; i < n; ++i) {
      const int rc = launch_wgrad_tn(pr[i].X, pr[i].ldx, pr[i].Y, pr[i].ldy, pr[i].M, pr[i].N1, pr[i].N2, pr[i].C, pr[i].ldc, st);
      if (rc) return rc;
    }
    return 0;
  }
  // ONE row split for the group: the smallest number of 64-row slabs per block (>= 8: the ring's fill and the tile's atomics must
  // stay a tail) with which the whole group is at most one resident set of blocks -- 256 (the 128-wide tile: one block per CU) or
  // 512 (the 64-wide one: two per CU); a few blocks more than that and the last ones would run alone
  const int64_t want = fdmi_det() ? 1 : fdmi_tune_get(45) > 0 ? fdmi_tune_get(45) : (bn1 == 64 ? 512 : 256);   // (deterministic mode: every tile one block)
  int64_t work = 0;
  for (int i = 0; i < n; ++i) work += (int64_t)gr.t1[i] * gr.t2[i] * ((gr.p[i].M + 63) / 64);
  int64_t spb = (work + want - 1) / want;
  if (spb < 8) spb = 8;
  auto blocks_at = [&](int64_t s) {
    int64_t b = 0;
    for (int i = 0; i < n; ++i) b += (int64_t)gr.t1[i] * gr.t2[i] * (((gr.p[i].M + 63) / 64 + s - 1) / s);
    return b;
  };
  int64_t max_slabs = 0;
  for (int i = 0; i < n; ++i) max_slabs = (gr.p[i].M + 63) / 64 > max_slabs ? (gr.p[i].M + 63) / 64 : max_slabs;
  while (spb < max_slabs && blocks_at(spb) > want) spb += spb / 32 > 0 ? spb / 32 : 1;   // (at max_slabs every tile is one block: the floor)
  int gx = 1, gy = 1, z = 0;
  for (int i = 0; i < n; ++i) {
    gr.p[i].rows_per_split = (int)(spb * 64);
    z += (int)((gr.p[i].M + gr.p[i].rows_per_split - 1) / gr.p[i].rows_per_split);
    gr.zend[i] = z;
    if (gr.t2[i] > gx) gx = gr.t2[i];
    if (gr.t1[i] > gy) gy = gr.t1[i];
  }
  gr.n = n;
  FDMI_CHECK(z <= 65535, "wgrad_tn_group: too many row splits");
  const dim3 grid(gx, gy, z);
  return bn1 == 64 ? launch_tn2_group<64>(gr, grid, flops, st) : launch_tn2_group<128>(gr, grid, flops, st);
}
