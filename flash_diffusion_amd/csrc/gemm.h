// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950.  out[M,N] = A[M,K] * W[N,K]^T (+epilogue)
#pragma once
#include "common.h"

enum { GEMM_ROW = 0, GEMM_CONV = 1 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GEGLU = 2, ACT_RELU = 3 /* conv + ReLU of the VGG16 feature stack (LPIPS) */,
       ACT_GELU = 4, ACT_GELU_TANH = 5 /* exact (erf) GELU: the MLP of OpenCLIP-style text encoders */ };

struct GemmArgs {
  int M = 0, N = 0, K = 0;
  const bf16_t* A = nullptr; int64_t lda = 0;   // ROW: A[M][lda]; CONV: NHWC activation base
  // ROW only: a second segment of the reduction index -- columns k >= K1 of the (virtual) [M][K] operand are A2[m][k - K1]
  // (row stride lda2), columns k < K1 are A[m][k].  K1 % 64 == 0: a 64-deep K tile never straddles the seam.  This is how a
  // channel concatenation [h | skip] feeds a GEMM without being materialised, and how a LoRA up-projection rides along as extra
  // K tiles of the base GEMM: y = [x | t] [W | B]^T.  Only the 256-row / 128-row LDS-DMA kernels implement it (gemm_a2_ok).
  const bf16_t* A2 = nullptr; int64_t lda2 = 0; int K1 = 0;
  const bf16_t* W = nullptr; int64_t ldw = 0;   // W[N][ldw], reduction index contiguous
  int mode = GEMM_ROW;
  // CONV geometry.  forward: X[B,Hin,Win,Cin] (virtually 2x nearest-upsampled when ups) ->
  // Y[B,Hout,Wout,N].  dgrad: "X" is dY of a forward conv with the same KH/KW/stride/pad and
  // Y is dX (gather form of the transposed convolution).
  int Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, KH = 1, KW = 1, stride = 1, pad = 0;
  int ups = 0, dgrad = 0;
  // epilogue:  v = alpha*acc + bias[n] + rowvec[m / rows_per_batch][n] + residual[m][n]; act
  const float* bias = nullptr;
  const bf16_t* rowvec = nullptr; int64_t rowvec_ld = 0; int rows_per_batch = 1;
  int rowvec_mul = 0;             // 1: the row vector MULTIPLIES: v = (alpha*acc + bias[n]) * rowvec[m / rows_per_batch][n] + residual[m][n]
                                  // (the adaLN gate of the transformer denoisers; not with ACT_GEGLU)
  const bf16_t* residual = nullptr; int64_t ldr = 0;
  // fp32 residual stream (round 5: the transformer denoisers' hidden state; the reference's bf16-mixed run keeps it in fp32 -- the
  // fp32 position table / scale_shift_table promote it -- and rounding it to bf16 after each of the ~56 residual adds of a 28-block
  // DiT cost 4x the reference's own bf16 deviation): residual32[m][n] (fp32) is ADDED like `residual`, and the fp32 value of the
  // result is ALSO stored to C32[m][n] beside the bf16 store to C (the shadow the next GEMM's A operand reads).  ACT_NONE only,
  // N, ldr32, ldc32 multiples of 8; run by the R32 instantiations of gemm3 / gemm4<192>, the small-tile kernel and the finalize
  // kernel (gemm_r32_ok).
  const float* residual32 = nullptr; int64_t ldr32 = 0;
  float* C32 = nullptr; int64_t ldc32 = 0;
  int act = ACT_NONE;             // ACT_GEGLU: N pre-activation columns in 16-wide (value|gate)
                                  // interleave -> N/2 output columns
  bf16_t* preact = nullptr; int64_t ldp = 0;   // optional save of the pre-activation (GEGLU bwd)
  void* C = nullptr; int64_t ldc = 0; int out_f32 = 0;
  float alpha = 1.f;
  int splitk = 1; float* ws = nullptr;  // splitk>1: raw f32 partial sums accumulate into ws[M][N]
  // set by launch_gemm (round 5): one zeroed counter per output tile -- the 256-row kernels then reduce the split-K slabs INSIDE the
  // launch (the block that writes a tile's last slab sums all of them in slab order and runs the epilogue; gemm_tile.h::
  // splitk_last_arriver) instead of leaving them to gemm_finalize_kernel
  int* sk_tickets = nullptr;
  int accum_atomic = 0;           // C is f32 and receives atomicAdd(alpha*acc) (wgrad accumulation)
  int force_tile = 0;             // 0 auto; else (BM<<16 | BN)
  int use_glds = 1;               // LDS-DMA staging (1) or register staging (0)
  // developer bits (0 = the launcher's default, knob 40).  16 / 32: timing ablations of scripts/rowbench.py on the gemm4 row
  // kernels (WRONG results: A re-read from its first tile / no epilogue); 64: the general epilogue instead of the lean one
  // (gemm_tile.h, A/B switch).  Round 4 also tried, measured and removed: L2 prefetch touches of the A / residual tiles and
  // burst issue of the ring pieces (profiles/r4_rowbench_dev.txt: no gain -- the row kernels were epilogue-instruction-bound)
  int dev = 0;
  // optional: GroupNorm statistics of the tensor this GEMM produces, accumulated by the epilogue into
  // gn_stats[m / gn_rows][gn_G][2] += (sum, sum of squares) over the group's gn_cpg channels (pre-zeroed by the caller) so
  // that the consuming GroupNorm skips its reduction pass.  Only honoured when gemm_gn_ok(args): ask first.
  float* gn_stats = nullptr; int gn_rows = 0, gn_cpg = 0, gn_G = 0;
  // fp32 validation mode (ref32.hip, launch_gemm32): every `bf16_t*` operand above then points to float data and the
  // output is float.  Operands may be strided along the reduction index and batched over a (b1, b2) grid -- the attention
  // products and the TN weight gradients are this same kernel.  The bf16 kernels ignore these fields (f32 must be 0).
  int f32 = 0;
  int64_t a_sk = 1, w_sk = 1;          // element stride along k (lda / ldw stay the row strides)
  int nb1 = 1, nb2 = 1;                // batch grid
  int64_t a_b1 = 0, a_b2 = 0, w_b1 = 0, w_b2 = 0, c_b1 = 0, c_b2 = 0;   // element offsets per batch index
};

int launch_gemm(const GemmArgs& a, hipStream_t stream);
int launch_gemm32(const GemmArgs& a, hipStream_t stream);   // a.f32 == 1 (launch_gemm forwards to it)
// kernel / tile / split-K selection (shared by the launcher and by callers that must size `ws`)
struct GemmPlan { int big = 0;  /* 0: gemm.hip tiles, 1: gemm3 (256 x BN), 2: gemm4 (256 x BN, BN = 320 | 192 | 384), 3: gemm5 (128 x 320, two blocks per CU) */ int BM = 128, BN = 128, splitk = 1; };
GemmPlan plan_gemm(const GemmArgs& a, bool ws_available);
// bytes of fp32 workspace the auto split-K plan wants for this problem (0 = no split)
size_t gemm_ws_bytes(const GemmArgs& a);
bool gemm3_eligible(const GemmArgs& a);
int gemm3_pick_bn(const GemmArgs& a);
int launch_gemm3(const GemmArgs& a, int BN, hipStream_t stream);
bool gemm4_eligible(const GemmArgs& a, int BN = 320);   // 256 x {320, 192, 384} tile (gemm4.hip)
bool gemm5_eligible(const GemmArgs& a);                 // 128 x 320 row tile, two blocks per CU (gemm5.hip)
int launch_gemm5(const GemmArgs& a, hipStream_t stream);
int launch_gemm4(const GemmArgs& a, hipStream_t stream, int BN = 320);
// true when launch_gemm can run this problem with a two-segment A operand (GemmArgs::A2): a kernel that implements it is
// eligible (row GEMM, M >= 256, N >= 128, K and K1 multiples of 64)
bool gemm_a2_ok(const GemmArgs& a);
// true when launch_gemm will run this problem on a kernel whose epilogue accumulates GemmArgs::gn_stats (a 256-row tile
// without split-K, full tiles inside one sample, 8-column-aligned operands); a.gn_* and a.ws-availability as at launch
bool gemm_gn_ok(const GemmArgs& a, bool ws_available);
// true when this problem's epilogue can carry the fp32 residual stream (GemmArgs::residual32 / C32)
bool gemm_r32_ok(const GemmArgs& a);
// algorithmic flops of one launch (2*M*N*K)
static inline double gemm_flops(const GemmArgs& a) { return 2.0 * a.M * (double)a.N * a.K; }
// algorithmic HBM bytes of one launch: every operand touched ONCE (A + W + C, + the residual; a GEGLU output is N / 2 wide; a
// convolution's A operand is its input tensor, taken as M * Cin elements -- exact for the stride-1 convolutions that carry the
// time; split-K slabs and re-reads are traffic, not algorithm) -- the same rule as bench.py::algorithmic_bytes
static inline double gemm_bytes(const GemmArgs& a) {
  const double nout = a.act == ACT_GEGLU ? a.N / 2 : a.N;
  const double a_el = (double)a.M * (a.mode == GEMM_CONV ? a.Cin : a.K);
  return 2.0 * (a_el + (double)a.N * a.K + (double)a.M * nout * (a.residual ? 2.0 : 1.0)) + (a.preact ? 2.0 * a.M * (double)a.N : 0.0);
}
static inline void gemm_prof_shape(const GemmArgs& a) {   // (fdmi_prof_shape: the launch's row of the per-shape table)
  fdmi_prof_shape(a.mode == GEMM_CONV ? 1 : 0, a.M, a.N, a.K,
                  (a.residual ? 1 : 0) | (a.act == ACT_GEGLU ? 2 : 0) | (a.dgrad ? 4 : 0) | (a.gn_stats ? 8 : 0) | ((a.splitk > 1 ? a.splitk : 0) << 8));
}
static inline bool gemm_hbm_side(const GemmArgs& a) { return gemm_flops(a) < FDMI_RIDGE_FLOP_PER_BYTE * gemm_bytes(a); }
