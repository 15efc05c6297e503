// TN product for the LoRA weight gradients:  C[n1][n2] += sum_m X[m][n1] * Y[m][n2]   (fp32 atomics into the flat gradient)
//
// The LoRA gradients contract over the ROWS of two row-major activations (dB = dY^T t, dA = dt^T x): both operands have the
// reduction index slow.  The NT kernels (gemm*.hip) want it fast, so the plan used to materialise four transposed copies per
// LoRA linear (transpose2d: 449 launches and ~10 GB of HBM traffic per C2 step at 1.8 TB/s, round-1 profile) before two
// split-K GEMMs.  This kernel reads X and Y as they are: a 64-row slab of each is loaded with coalesced 16-byte reads and
// scattered into LDS TRANSPOSED (2-byte writes) into exactly the [index][64 k] image with the 16-byte-chunk XOR swizzle that
// gemm.hip's fragment reads expect; the MFMA loop is that kernel's.  Developer knob 16 until its first GPU run.
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

  // loader: chunk q = tid + 256 * i of a [64 m][BN / 8 chunks] slab; X: 2 chunks per thread, Y: 4
  constexpr int XI = TN_BK * (TN_BN1 / 8) / 256, YI = TN_BK * (TN_BN2 / 8) / 256;
  uint4 xr[XI], yr[YI];
  auto load = [&](int kt) {
    const int64_t m0 = m_beg + (int64_t)kt * TN_BK;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int q = tid + 256 * i, m = q / (TN_BN1 / 8), c = q % (TN_BN1 / 8);
      const int n = n1_0 + c * 8;
      xr[i] = (m0 + m < m_end && n < a.N1) ? *(const uint4*)(a.X + (m0 + m) * a.ldx + n) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < YI; ++i) {
      const int q = tid + 256 * i, m = q / (TN_BN2 / 8), c = q % (TN_BN2 / 8);
      const int n = n2_0 + c * 8;
      yr[i] = (m0 + m < m_end && n < a.N2) ? *(const uint4*)(a.Y + (m0 + m) * a.ldy + n) : make_uint4(0, 0, 0, 0);
    }
  };
  auto scatter = [&](char* img, const uint4& v, int row0, int k) {   // 8 values of one source row -> 8 image rows, column k
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 8; ++e)
      *(bf16_t*)(img + img_off(row0 + e, k)) = (bf16_t)(e & 1 ? w[e >> 1] >> 16 : w[e >> 1] & 0xffffu);
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int q = tid + 256 * i;
      scatter(sx, xr[i], (q % (TN_BN1 / 8)) * 8, q / (TN_BN1 / 8));
    }
#pragma unroll
    for (int i = 0; i < YI; ++i) {
      const int q = tid + 256 * i;
      scatter(sy, yr[i], (q % (TN_BN2 / 8)) * 8, q / (TN_BN2 / 8));
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
  // The tile is 64 (first operand) x 128 (second): a product with a narrow SECOND operand (dB = dY^T t: N2 = the LoRA rank) would
  // fill half of every tile with zeros and walk the wide operand in 64-column tiles.  It runs with the operands swapped --
  // t^T dY, the wide operand in 128-column tiles -- and stores its result transposed (round 4; C4: 586 launches, 54 ms per step).
  if (N2 <= TN_BN1 && N1 >= TN_BN2 && !fdmi_tune_get(43)) {
    const bf16_t* tp = X; X = Y; Y = tp;
    const int64_t tl = ldx; ldx = ldy; ldy = tl;
    const int tn = N1; N1 = N2; N2 = tn;
    ct = 1;
  }
  const int t1 = cdiv(N1, TN_BN1), t2 = cdiv(N2, TN_BN2);
  const int64_t slabs = (M + TN_BK - 1) / TN_BK;
  // ~4 blocks per CU: enough row splits to fill the chip, at least 4 slabs each so the atomics stay a small tail
  int64_t splits = (1024 + (int64_t)t1 * t2 - 1) / ((int64_t)t1 * t2);   // (512 ... 2048 blocks measure the same on C2; 256 is slower)
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
