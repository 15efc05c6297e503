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

}  // namespace

int launch_wgrad_tn(const bf16_t* X, int64_t ldx, const bf16_t* Y, int64_t ldy, int64_t M, int N1, int N2, float* C,
                    int64_t ldc, hipStream_t st) {
  FDMI_CHECK(X && Y && C && M > 0 && N1 > 0 && N2 > 0, "wgrad_tn: empty problem / null operand");
  FDMI_CHECK((N1 % 8) == 0 && (N2 % 8) == 0 && (ldx % 8) == 0 && (ldy % 8) == 0, "wgrad_tn: widths and leading dims must be multiples of 8");
  FDMI_CHECK(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0, "wgrad_tn: operands must be 16-B aligned");
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
  const int64_t want = fdmi_tune_get(45) > 0 ? fdmi_tune_get(45) : 1024;   // blocks to aim for (developer knob 45: scripts/wgrad_rates.py)
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
