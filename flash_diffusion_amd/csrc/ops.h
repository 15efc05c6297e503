// Internal launcher declarations (one per hand-written kernel family).
#pragma once
#include "common.h"
#include "gemm.h"

// ---- wgrad.hip ----  C[n1][n2] += sum_m X[m][n1] * Y[m][n2] (fp32 atomics; both operands row-major, contraction over rows)
int launch_wgrad_tn(const bf16_t* X, int64_t ldx, const bf16_t* Y, int64_t ldy, int64_t M, int N1, int N2, float* C,
                    int64_t ldc, hipStream_t st);

// ---- norm.hip ----
int launch_groupnorm_fwd(const bf16_t* x, const float* gamma, const float* beta, float* stats,
                         bf16_t* y, int B, int HW, int C, int G, float eps, int silu, hipStream_t st,
                         bool stats_zeroed = false,   // stats_zeroed: caller already cleared the accumulators
                         bool stats_ready = false,    // stats_ready: they already hold the sums (producer's GEMM epilogue)
                         const bf16_t* x2 = nullptr, int C1 = 0);   // x = [x[.][C1] | x2[.][C - C1]], never materialised
int launch_groupnorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* beta,
                         const float* stats, float* bstats, bf16_t* dx, int B, int HW, int C, int G,
                         float eps, int silu, int accumulate, hipStream_t st, bool stats_zeroed = false,
                         const bf16_t* x2 = nullptr, int C1 = 0);
int launch_layernorm_fwd(const bf16_t* x, const float* gamma, const float* beta, const bf16_t* shift,
                         const bf16_t* scale, int64_t mod_ld, int rows_per_batch, bf16_t* y,
                         int64_t rows, int C, float eps, hipStream_t st, float* stats = nullptr, const float* x32 = nullptr);
int launch_layernorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const bf16_t* scale,
                         int64_t mod_ld, int rows_per_batch, bf16_t* dx, int64_t rows, int C,
                         float eps, int accumulate, hipStream_t st, const float* x32 = nullptr);   // x32: the fp32 master of x, read instead of it

// ---- attn.hip ----
__host__ __device__ static inline int attn_spad(int S) { return (S + 63) & ~63; }      // padded sequence length
__host__ __device__ static inline int attn_dvpad(int d) { return (d + 15) & ~15; }     // padded head dim (rows of X^T)
struct AttnArgs {
  // row-major operands [B, S, ld] with head h at column offset h*d
  const bf16_t* Q; int64_t ldq;
  const bf16_t* K; int64_t ldk;
  const bf16_t* V; int64_t ldv;
  const bf16_t* O; int64_t ldo;     // forward output (bwd: saved output, only for delta)
  const bf16_t* dO; int64_t lddo;
  // head-transposed copies [B, H, dvpad, spad] (zero padded), produced by launch_transpose_heads
  const bf16_t* QT; const bf16_t* KT; const bf16_t* VT; const bf16_t* dOT;
  float* lse;        // [B, H, Sq] log2-domain log-sum-exp of scale*log2e*QK^T
  const float* delta;  // [B, H, Sq] rowsum(dO * O)
  bf16_t* out;  int64_t ldout;     // fwd: O; dq kernel: dQ
  bf16_t* dK; int64_t lddk; bf16_t* dV; int64_t lddv;
  int B, H, Sq, Skv, d; float scale;
  int vt_ones;       // VT carries a row of ones at dd = d (written by launch_transpose_heads(..., ones_row=1))
};
int launch_attn_fwd(const AttnArgs& a, hipStream_t st);
int launch_attn_bwd_dq(const AttnArgs& a, hipStream_t st);
int launch_attn_bwd_dkv(const AttnArgs& a, hipStream_t st);
// X[B,S,ld] (head h at h*d) -> XT[B,H,dvpad,spad], zero padded
int launch_transpose_heads(const bf16_t* X, int64_t ld, bf16_t* XT, int B, int H, int S, int d,
                           hipStream_t st, int ones_row = 0);
// delta[b,h,s] = sum_d dO*O
int launch_attn_delta(const bf16_t* O, int64_t ldo, const bf16_t* dO, int64_t lddo, float* delta,
                      int B, int H, int S, int d, hipStream_t st);

// ---- elem.hip ----
int launch_nchw_to_nhwc(const float* x, bf16_t* y, int B, int C, int HW, int Cpad, hipStream_t st);
int launch_nhwc_to_nchw(const bf16_t* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate,
                        hipStream_t st);
int launch_nchw_grad_to_nhwc(const float* g, bf16_t* y, int64_t ldy, int B, int C, int HW, hipStream_t st);
// y[NHWC bf16, width C] += scale * r[NCHW f32]
int launch_add_nchw_to_nhwc(const float* r, float scale, bf16_t* y, int B, int C, int HW, hipStream_t st);
int launch_timestep_embed(const float* t, bf16_t* out, int B, int dim, int flip, float shift, hipStream_t st);
int launch_silu(const bf16_t* x, bf16_t* y, int64_t n, hipStream_t st);
// dst[m][dc0 + c] (=|+=) src[m][sc0 + c], c < cols
int launch_copy2d(const bf16_t* src, int64_t lds, int sc0, bf16_t* dst, int64_t ldd, int dc0,
                  int64_t rows, int cols, int accumulate, hipStream_t st);
int launch_f32_to_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t st);
// dX[B,H,W,C] (=|+=) sum of the 2x2 block of dY[B,2H,2W,C]
int launch_pool2x2_sum(const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C, int accumulate, hipStream_t st);
// GEGLU backward: pre[M][2F] in 16-wide (value|gate) interleave, dout[M][F] -> dpre[M][2F]
int launch_geglu_bwd(const bf16_t* pre, const bf16_t* dout, bf16_t* dpre, int64_t M, int F, hipStream_t st);
// out[c][r] = in[r][c]  (bf16 matrix transpose; in ld = ldi, out ld = ldo)
int launch_transpose2d(const bf16_t* in, int64_t ldi, bf16_t* out, int64_t ldo, int64_t rows, int cols,
                       hipStream_t st);
int launch_transpose2d_pad(const bf16_t* in, int64_t ldi, bf16_t* out, int64_t ldo, int64_t rows, int cols,
                           int64_t rows_pad, hipStream_t st);
// dst[rows][cols_pad] = src[rows][cols] zero padded
int launch_pad_cols(const bf16_t* src, int cols, bf16_t* dst, int cols_pad, int64_t rows, hipStream_t st);
// LoRA refresh: f32 master W[rows][cols] -> bf16 copy and bf16 transpose
int launch_cast_transpose(const float* w, bf16_t* wb, bf16_t* wtb, int rows, int cols, hipStream_t st);
// one 64x64 tile of an f32 [rows][cols] -> bf16 copy + bf16 transposed copy (LoRA refresh of a whole plan in one launch)
struct CastJob {
  const float* src; bf16_t* dst; bf16_t* dstT; int rows, cols, r0, c0; int ldT; /* row stride of dstT (0: rows) */
  // optional second pair of destinations with their own row strides (the LoRA halves of the fused [W | B] / [W^T | A^T] operands)
  bf16_t* dst2; int ld2; bf16_t* dstT2; int ldT2;
};
int launch_cast_transpose_jobs(const CastJob* jobs, int njobs, hipStream_t st);
// fused AdamW on a flat f32 buffer
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                 float eps, float wd, int step, float grad_scale, hipStream_t st);

// ---- dit.hip: adaLN-single DiT element-wise kernels ----
// y = res + gate[m / rows_per_batch] * x  (res may be null)
int launch_gate_residual(const bf16_t* x, const bf16_t* gate, int64_t gate_ld, const bf16_t* res, bf16_t* y,
                         int64_t rows, int C, int rows_per_batch, hipStream_t st, const float* res32 = nullptr, float* y32 = nullptr);
int launch_gelu_tanh(const bf16_t* x, bf16_t* y, int64_t n, hipStream_t st);
int launch_gelu_tanh_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n, hipStream_t st);
// out1[b][c] = sum_r dy ; out0[b][c] = sum_r dy * f(x)  (f = LayerNorm normalisation from stats[rows][2], identity if null)
int launch_batch_colsum(const bf16_t* dy, const bf16_t* x, const float* stats, float* out0, float* out1, int B,
                        int rows_per_batch, int C, hipStream_t st);

// ---- discriminator support + fused loss kernels (elem.hip) ----
int launch_im2col(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride,
                  int pad, hipStream_t st);
int launch_silu_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n, hipStream_t st);
int launch_colsum(const bf16_t* dy, const bf16_t* x, const float* stats, float* out0, float* out1, int64_t rows, int C,
                  int HW, int G, float eps, hipStream_t st);
int launch_distill_loss(const float* s, const float* t, int64_t n, int l1, float* out, hipStream_t st);
int launch_distill_grad(const float* s, const float* t, int64_t n, int l1, float gscale, float* ds, hipStream_t st);
int launch_dmd_loss(const float* s, const float* noisy, const float* real, const float* fake, const float* ia,
                    const float* ma, const float* kb, float* w, float* grad, float* loss, int B, int64_t per,
                    hipStream_t st);
// ---- fused scheduler element-wise math ----
struct StepCoef { float a, b, c, d; };
int launch_add_noise(const float* z, const float* noise, const float* sa, const float* sb, float* out,
                     int B, int64_t per, hipStream_t st);
int launch_axpby4(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                  const float* x3, float c3, float* out, int64_t n, hipStream_t st);

// ---- netops.hip: the frozen convolutional nets beside the denoiser (VGG16 / LPIPS, VAE, T2I adapter); `32` = fp32 plans ----
int launch_relu_mask(const bf16_t* y, bf16_t* dy, int64_t n, hipStream_t st);                 // dy *= (y > 0)
int launch_relu_mask32(const float* y, float* dy, int64_t n, hipStream_t st);
int launch_maxpool2_fwd(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, hipStream_t st);
int launch_maxpool2_fwd32(const float* x, float* y, int B, int H, int W, int C, hipStream_t st);
int launch_maxpool2_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C, int accumulate, hipStream_t st);
int launch_maxpool2_bwd32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int accumulate, hipStream_t st);
int launch_avgpool2_fwd(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, hipStream_t st);
int launch_avgpool2_fwd32(const float* x, float* y, int B, int H, int W, int C, hipStream_t st);
int launch_pixel_unshuffle(const float* x, bf16_t* y, int B, int C, int H, int W, int f, int Cpad, hipStream_t st);
int launch_pixel_unshuffle32(const float* x, float* y, int B, int C, int H, int W, int f, int Cpad, hipStream_t st);
int launch_nhwc_to_nchw_any(const bf16_t* x, float* y, int B, int C, int HW, hipStream_t st);
int launch_nhwc_to_nchw_any32(const float* x, float* y, int B, int C, int HW, hipStream_t st);
int launch_bf16_to_f32(const bf16_t* x, float* y, int64_t n, hipStream_t st);
// T5 text encoder pieces: T5LayerNorm (RMS norm, fp32 weight), the gated feed-forward's product
int launch_rmsnorm(const bf16_t* x, const float* w, bf16_t* y, int64_t rows, int C, float eps, hipStream_t st);
int launch_rmsnorm32(const float* x, const float* w, float* y, int64_t rows, int C, float eps, hipStream_t st);
int launch_ewise_mul(const bf16_t* a, const bf16_t* b, bf16_t* y, int64_t n, hipStream_t st);
int launch_ewise_mul32(const float* a, const float* b, float* y, int64_t n, hipStream_t st);
// out[b] += mean over the sample's pixels of sum_c w[c] (f0 / (|f0| + 1e-10) - f1 / (|f1| + 1e-10))^2 ; rows = B * HW pixel rows
int launch_lpips_level_fwd(const bf16_t* f0, const bf16_t* f1, const float* w, float* out, int64_t rows, int HW, int C, hipStream_t st);
int launch_lpips_level_fwd32(const float* f0, const float* f1, const float* w, float* out, int64_t rows, int HW, int C, hipStream_t st);
int launch_lpips_level_bwd(const bf16_t* f0, const bf16_t* f1, const float* w, const float* gout, bf16_t* df0, int64_t rows, int HW,
                           int C, int accumulate, hipStream_t st);
int launch_lpips_level_bwd32(const float* f0, const float* f1, const float* w, const float* gout, float* df0, int64_t rows, int HW,
                             int C, int accumulate, hipStream_t st);
// LPIPS ScalingLayer + layout: NCHW f32 [B,3,HW] -> NHWC [B*HW][Cpad]; shift / scale: 3 HOST floats each
int launch_lpips_input(const float* x, bf16_t* y, int B, int HW, int Cpad, const float* shift, const float* scale, hipStream_t st);
int launch_lpips_input32(const float* x, float* y, int B, int HW, int Cpad, const float* shift, const float* scale, hipStream_t st);
int launch_lpips_input_bwd(const bf16_t* dy, float* dx, int B, int HW, int Cpad, const float* scale, hipStream_t st);
int launch_lpips_input_bwd32(const float* dy, float* dx, int B, int HW, int Cpad, const float* scale, hipStream_t st);

// ---- ref32.hip: fp32 validation mode (exact-f32 MFMA contractions, fp32 storage, fp64 statistics) ----
int launch_wgrad_tn32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t M, int N1, int N2, float* C, int64_t ldc,
                      hipStream_t st);
int64_t attn32_scratch_elems(int B, int H, int Sq, int Skv, int bwd);   // floats of scratch the two calls below want
int launch_attn32_fwd(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                      int B, int H, int Sq, int Skv, int d, float scale, float* scratch, int64_t scratch_elems, hipStream_t st,
                      int causal = 0,    // causal: key j attends query i only for j <= i (the CLIP text encoder; Sq == Skv)
                      const float* bias = nullptr,     // [H][Sq][Skv] added to the scaled scores (T5's relative-position bias)
                      const float* kbias = nullptr);   // [B][Skv] additive key mask
int launch_attn32_bwd(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, const float* dO,
                      int64_t lddo, float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV, int64_t lddv, int B, int H, int Sq,
                      int Skv, int d, float scale, float* scratch, int64_t scratch_elems, hipStream_t st);
// stats[b][g] = (mean, rstd)
int launch_groupnorm32_fwd(const float* x, const float* gamma, const float* beta, float* stats, float* y, int B, int HW, int C, int G,
                           float eps, int silu, hipStream_t st);
int launch_groupnorm32_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* stats, float* dx,
                           int B, int HW, int C, int G, int silu, int accumulate, hipStream_t st);
int launch_layernorm32_fwd(const float* x, const float* gamma, const float* beta, const float* shift, const float* scale,
                           int64_t mod_ld, int rows_per_batch, float* y, int64_t rows, int C, float eps, hipStream_t st,
                           float* stats = nullptr);
int launch_layernorm32_bwd(const float* x, const float* dy, const float* gamma, const float* scale, int64_t mod_ld, int rows_per_batch,
                           float* dx, int64_t rows, int C, float eps, int accumulate, hipStream_t st);
int launch_nchw_to_nhwc32(const float* x, float* y, int B, int C, int HW, int Cpad, hipStream_t st);
int launch_nhwc_to_nchw32(const float* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate, hipStream_t st);
int launch_add_nchw_to_nhwc32(const float* r, float scale, float* y, int B, int C, int HW, hipStream_t st);
int launch_timestep_embed32(const float* t, float* out, int B, int dim, int flip, float shift, hipStream_t st);
int launch_silu32(const float* x, float* y, int64_t n, hipStream_t st);
int launch_silu32_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st);
int launch_gelu_tanh32(const float* x, float* y, int64_t n, hipStream_t st);
int launch_gelu_tanh32_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st);
int launch_copy2d32(const float* src, int64_t lds, int sc0, float* dst, int64_t ldd, int dc0, int64_t rows, int cols, int accumulate,
                    hipStream_t st);
int launch_pool2x2_sum32(const float* dy, float* dx, int B, int H, int W, int C, int accumulate, hipStream_t st);
int launch_geglu32_bwd(const float* pre, const float* dout, float* dpre, int64_t M, int F, hipStream_t st);
int launch_pad_cols32(const float* src, int cols, float* dst, int cols_pad, int64_t rows, hipStream_t st);
int launch_im2col32(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                    hipStream_t st);
int launch_colsum32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int64_t rows, int C, int HW, int G,
                    hipStream_t st);
int launch_gate_residual32(const float* x, const float* gate, int64_t gate_ld, const float* res, float* y, int64_t rows, int C,
                           int rows_per_batch, hipStream_t st);
int launch_batch_colsum32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int B, int rows_per_batch,
                          int C, hipStream_t st);
