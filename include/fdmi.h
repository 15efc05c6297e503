/* fdmi.h -- C ABI of libfdmi.so: the MI355X (gfx950) native Flash-Diffusion distillation hot path.
 *
 * The reference (gojasper/flash-diffusion) is 100% Python and has NO FFI: every FLOP of the hot path
 * runs inside third-party PyTorch/diffusers kernels reached through
 *     DiffusersUNet2DCondWrapper.forward  -> UNet2DConditionModel.forward(...).sample
 *         (/root/reference/src/flash/models/unets/unet.py:66-119, call at 108-119)
 * driven by FlashDiffusion.forward (/root/reference/src/flash/models/flash/flash_diffusion_model.py:
 * 179-366; student call 260-265, teacher loop 288-324, DMD 401-499, GAN 501-667) and
 * TrainingPipeline.training_step (/root/reference/src/flash/trainer/trainer.py:169-218).
 * This library is what a binding for that path would load (INTEGRATION.md shows the ctypes stub):
 *   - fdmi_unet_*      replaces UNet2DConditionModel.forward / autograd backward  (unet.py:108-119)
 *   - fdmi_gemm/attn/groupnorm/layernorm/... the individual kernels (unit-testable building blocks)
 *   - fdmi_adamw       replaces torch.optim.AdamW.step on the LoRA/discriminator params (trainer.py:100-107)
 *   - fused element-wise helpers replace the scheduler / CFG / loss arithmetic of
 *     flash_diffusion_model.py:250-252, 316-324, 328, 368-382, 466-499.
 *
 * Conventions: all functions return 0 on success, negative on error (message: fdmi_last_error(),
 * thread-local).  All pointers are DEVICE pointers owned by the caller unless stated otherwise;
 * `stream` is a hipStream_t.  bf16 tensors are raw uint16 bit patterns.  No exceptions cross the ABI,
 * no torch types appear in any signature.  The library allocates device memory only inside
 * fdmi_unet_create / fdmi_unet_set_param (packed weights) -- never on the forward/backward path.
 */
#ifndef FDMI_H_
#define FDMI_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* fdmi_last_error(void);
int fdmi_version(void);
/* Optional per-launch HIP-event timing of the MFMA kernels on the stream they are launched on
 * (bench.py roofline leg).  Buckets 0-7: gemm_kernel<BM,BN,mode> = mode*4 + (BM==64)*2 + (BN==64);
 * 8: attention fwd, 9: attention dQ, 10: attention dK/dV.  collect() synchronises, sums and resets. */
int fdmi_tune_set(int key, int value);   /* developer knobs for kernel-variant A/B runs (keys 0..63, default 0) */
/* key 50 = 1: DETERMINISTIC MODE (round 6).  Every accumulation whose order the production kernels leave to the hardware -- the fp32
 * atomics of the GroupNorm-sum GEMM epilogues (the plans then run the statistics pass instead), of that pass's blocks, of the TN
 * weight-gradient row splits (one split per tile), of atomic-accumulating GEMMs (split-K 1), of the column sums and of the scalar
 * loss kernels -- runs in a fixed order: two runs of the same step on the same inputs are bit-identical.  A test / debugging mode
 * (slower, same kernels otherwise); read at every launch, so it can be flipped between calls. */
int fdmi_tune_value(int key);            /* current value of a knob (0 for an unknown key) */
int fdmi_prof_enable(int on);
int fdmi_prof_collect(int nbuckets, double* ms, double* flops, int64_t* launches);
/* round 6: the same, plus the algorithmic HBM bytes of every bucket's launches (each operand of a launch counted once: GEMM / conv
 * A + W + C (+ residual), attention q + k + v + o) -- bench.py prices every kernel family against the bound it sits under
 * (arithmetic intensity vs the 2.5 PFLOP/s : 8 TB/s ridge).  Buckets 20 / 21: the row GEMMs of the 256 x 320 / 256 x {128,160}
 * ring kernels on the HBM side of that ridge; 22 / 23: the 128 x 320 two-blocks-per-CU row kernel and its HBM-side subset.
 * nbuckets >= 24. */
int fdmi_prof_collect2(int nbuckets, double* ms, double* flops, int64_t* launches, double* bytes);
/* round 6: one CSV line per launch of the leg the last fdmi_prof_collect* call gathered -- bucket, kind (0 row GEMM, 1 conv: s = M, N, K,
 * flags; 2 / 3 / 4 attention forward / dQ / dK-dV: s = B * H, Sq, Skv, d; -1 untagged), milliseconds, algorithmic flops and bytes --
 * the per-shape table of scripts/shape_table.py (which GEMM SHAPES of a step sit furthest from their bound).  Returns the line count. */
int fdmi_prof_dump(const char* path);

/* ---------------- GEMM / implicit-GEMM convolution (bf16 MFMA, fp32 accumulate) ----------------
 * out[M,N] = A[M,K] * W[N,K]^T ; epilogue v = alpha*acc + bias[n] + rowvec[m/rows_per_batch][n]
 *            + residual[m][n]; act (0 none, 1 SiLU, 2 GEGLU on 16-wide (value|gate) interleave, 3 ReLU, 4 GELU (erf),
 *            5 GELU (tanh approximation: diffusers FeedForward "gelu-approximate", PixArt / SD3 blocks)).
 * mode 0: A is row-major [M][lda].  mode 1: A is an NHWC activation gathered as im2col
 * (Hin,Win,Cin -> Hout,Wout; KHxKW, stride 1|2, zero pad, ups: fused nearest-2x upsample of the
 * input; dgrad: gather form of the transposed convolution).                                     */
typedef struct fdmi_gemm_desc {
  int32_t M, N, K;
  const void* A; int64_t lda;
  const void* W; int64_t ldw;
  int32_t mode;
  int32_t Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, ups, dgrad;
  const float* bias;
  const void* rowvec; int64_t rowvec_ld; int32_t rows_per_batch;
  const void* residual; int64_t ldr;
  int32_t act;
  void* preact; int64_t ldp;
  void* C; int64_t ldc; int32_t out_f32;
  float alpha;
  int32_t splitk; float* ws;
  int32_t accum_atomic;
  int32_t force_tile;
  int32_t use_glds;
  /* mode 0 only, optional (A2 == NULL: absent): a second segment of the reduction index -- columns k >= K1 of the [M][K]
   * operand are A2[m][k - K1] (row stride lda2), columns k < K1 are A[m][k]; K1 % 64 == 0, M >= 256, N >= 128.  Feeds a channel
   * concatenation (the UNet's up-path [h | skip], diffusers UNet2DConditionModel up blocks) or a LoRA up-projection
   * (y = [x | t] [W | B]^T, peft's y = W x + B A x, examples/train_flash_sd.py:191-200) through ONE GEMM without materialising
   * the concatenated operand.  fdmi_gemm_a2_ok (host only) says whether the problem qualifies.                          */
  const void* A2; int64_t lda2; int32_t K1;
  /* 1: the row vector multiplies instead of adding: v = (alpha*acc + bias[n]) * rowvec[m/rows_per_batch][n] + residual[m][n]
   * -- the adaLN gate + residual of a transformer denoiser block (diffusers BasicTransformerBlock ada_norm_single:
   * hidden_states = gate_msa * attn_output + hidden_states; JointTransformerBlock likewise) in the projection's epilogue.  */
  int32_t rowvec_mul;
} fdmi_gemm_desc;
int fdmi_gemm_a2_ok(const fdmi_gemm_desc* d);
int fdmi_gemm(const fdmi_gemm_desc* d, void* stream);
/* host-only planner query: kernel 0 = 128/64-row tiles (gemm.hip), 1 = 256 x {128,160} LDS-DMA ring (gemm3.hip),
 * 2 = 256 x 320 / 256 x 192 (gemm4.hip); the tile and the split-K factor that fdmi_gemm would use (splitk <= 0 in the descriptor
 * = let the planner split).  No device work, no GPU needed.                                       */
int fdmi_gemm_plan(const fdmi_gemm_desc* d, int32_t* kernel, int32_t* BM, int32_t* BN, int32_t* splitk);
/* The GEMM / conv above whose epilogue ALSO accumulates the GroupNorm statistics of its output for the consumer:
 * gn_stats[m / gn_rows][gn_G][2] += (sum, sum of squares) of the stored bf16 values over each group of N / gn_G channels
 * (caller pre-zeroes gn_stats; gn_rows = rows per sample = H*W).  Only the 256-row kernels without split-K own this epilogue:
 * fdmi_gemm_gn_ok (host only, no GPU needed) returns 1 when fdmi_gemm_gn will accept the problem, 0 otherwise (then run
 * fdmi_gemm and fdmi_groupnorm_fwd).  fdmi_groupnorm_apply is fdmi_groupnorm_fwd without its reduction pass.        */
int fdmi_gemm_gn_ok(const fdmi_gemm_desc* d, int gn_rows, int gn_G);
int fdmi_gemm_gn(const fdmi_gemm_desc* d, float* gn_stats, int gn_rows, int gn_G, void* stream);
int fdmi_groupnorm_apply(const void* x, const float* gamma, const float* beta, const float* stats /*[B][G][2] sums*/,
                         void* y, int B, int HW, int C, int G, float eps, int silu, void* stream);

/* TN product for weight gradients that contract over the ROWS of two row-major bf16 activations (the LoRA gradients
 * dB = dY^T t, dA = dt^T x of peft's y = W x + B A x, examples/train_flash_sd.py:191-200):
 * C[n1][n2] += sum_m X[m][n1] * Y[m][n2], C fp32 [N1][ldc] accumulated with atomics (caller zeroes it once per step).  */
int fdmi_wgrad_tn(const void* X, int64_t ldx, const void* Y, int64_t ldy, int64_t M, int N1, int N2, float* C, int64_t ldc,
                  void* stream);
/* Up to 6 such products in ONE launch (the (dB, dA) pair of a LoRA-carrying linear, the three pairs of a fused q/k/v
 * projection): the group fills the GPU together, so each product runs with fewer row splits and fewer fp32 atomics than alone.
 * Same operand rules per problem as fdmi_wgrad_tn; results as if the products ran one after the other.                     */
typedef struct fdmi_wgrad_problem {
  const void* X; int64_t ldx;
  const void* Y; int64_t ldy;
  int64_t M; int32_t N1, N2;
  float* C; int64_t ldc;
} fdmi_wgrad_problem;
int fdmi_wgrad_tn_group(const fdmi_wgrad_problem* problems, int n, void* stream);

/* ---------------- normalisation (NHWC / token-major bf16, fp32 statistics) -------------------- */
int fdmi_groupnorm_fwd(const void* x, const float* gamma, const float* beta, float* stats /*[B][G][2]*/,
                       void* y, int B, int HW, int C, int G, float eps, int silu, void* stream);
int fdmi_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                       const float* stats, float* bstats, void* dx, int B, int HW, int C, int G, float eps,
                       int silu, int accumulate, void* stream);
/* GroupNorm of a channel concatenation that is never materialised: x = [x1[B*HW][C1] | x2[B*HW][C - C1]] (the up path of
 * diffusers' UNet2DConditionModel concatenates the hidden state with a skip tensor in front of every ResNet block's norm1);
 * y / dy / dx span all C channels.  C1 % 8 == 0.                                                                       */
int fdmi_groupnorm_cat_fwd(const void* x1, const void* x2, int C1, const float* gamma, const float* beta, float* stats, void* y,
                           int B, int HW, int C, int G, float eps, int silu, void* stream);
int fdmi_groupnorm_cat_bwd(const void* x1, const void* x2, int C1, const void* dy, const float* gamma, const float* beta,
                           const float* stats, float* bstats, void* dx, int B, int HW, int C, int G, float eps, int silu,
                           int accumulate, void* stream);
int fdmi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int C,
                       float eps, void* stream);
int fdmi_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, int64_t rows, int C,
                       float eps, int accumulate, void* stream);
/* adaLN modulate (DiT, reference transformers/utils.py:8-102 feeding diffusers' ada_norm_single blocks):
 * y = LayerNorm(x) * (1 + scale[b]) + shift[b], b = row / rows_per_batch; shift / scale bf16 [B][mod_ld];
 * stats (optional) receives (mean, rstd) per row for fdmi_batch_colsum.                          */
int fdmi_layernorm_mod_fwd(const void* x, const void* shift, const void* scale, int64_t mod_ld, int rows_per_batch,
                           void* y, float* stats /*[rows][2] or NULL*/, int64_t rows, int C, float eps, void* stream);
int fdmi_layernorm_mod_bwd(const void* x, const void* dy, const void* scale, int64_t mod_ld, int rows_per_batch, void* dx,
                           int64_t rows, int C, float eps, int accumulate, void* stream);
/* y = res + gate[b] * x (res may be NULL); tanh-GELU and its backward; per-sample column sums
 * out1[b][c] = sum_r dy, out0[b][c] = sum_r dy * f(x) (f: LayerNorm normalisation from stats, identity if NULL). */
int fdmi_gate_residual(const void* x, const void* gate, int64_t gate_ld, const void* res, void* y, int64_t rows, int C,
                       int rows_per_batch, void* stream);
int fdmi_gelu_tanh(const void* x, void* y, int64_t n, void* stream);
int fdmi_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
int fdmi_batch_colsum(const void* dy, const void* x, const float* stats, float* out0, float* out1, int B,
                      int rows_per_batch, int C, void* stream);

/* ---------------- fused attention (token-major [B,S,H*d] bf16) --------------------------------
 * fwd:  O = softmax(scale Q K^T) V.  `VT` is scratch of fdmi_attn_tr_elems(B,H,Skv,d) bf16 elements
 * (the head-transposed copy of V the PV product reads); lse [B,H,Sq] f32 or NULL.
 * bwd:  given dO (and the forward's O, lse) writes dQ, dK, dV; `ws` is scratch of
 * fdmi_attn_bwd_ws_bytes(...) bytes.                                                             */
int64_t fdmi_attn_tr_elems(int B, int H, int S, int d);
int64_t fdmi_attn_bwd_ws_bytes(int B, int H, int Sq, int Skv, int d);
int fdmi_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  void* O, int64_t ldo, void* VT, float* lse, int B, int H, int Sq, int Skv, int d,
                  float scale, void* stream);
int fdmi_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  const void* O, int64_t ldo, const void* dO, int64_t lddo, const float* lse,
                  void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv, void* ws,
                  int B, int H, int Sq, int Skv, int d, float scale, void* stream);

/* ---------------- element-wise / layout helpers ----------------------------------------------- */
int fdmi_nchw_to_nhwc(const float* x, void* y, int B, int C, int HW, int Cpad, void* stream);
int fdmi_nhwc_to_nchw(const void* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate, void* stream);
int fdmi_timestep_embed(const float* t, void* out, int B, int dim, int flip, float shift, void* stream);
int fdmi_geglu_bwd(const void* pre, const void* dout, void* dpre, int64_t M, int F, void* stream);
int fdmi_pool2x2_sum(const void* dy, void* dx, int B, int H, int W, int C, int accumulate, void* stream);
int fdmi_cast_transpose(const float* w, void* wb, void* wtb, int rows, int cols, void* stream);
int fdmi_transpose2d(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, void* stream);
int fdmi_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float grad_scale, void* stream);
int fdmi_add_noise(const float* z, const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                   float* out, int B, int64_t per_sample, void* stream);
/* out = c0*x0 + c1*x1 + c2*x2 + c3*x3 (x1..x3 may be NULL): CFG combine, DPM-Solver++ update,
 * boundary-condition combine, x0 prediction ... on fp32 latents.                                */
int fdmi_axpby4(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                const float* x3, float c3, float* out, int64_t n, void* stream);

/* ---------------- discriminator building blocks (examples/train_flash_sd.py:225-240) and fused losses ----
 * The PatchGAN head is trainable, so besides fdmi_gemm (forward / dgrad) it needs a weight gradient:
 * dW = dY^T * im2col(X) as a plain GEMM over the explicit patch matrix (tiny: <= 2B x 8 x 8 pixels).
 * fdmi_colsum: out0[c] += sum_rows dy ; out1[c] += sum_rows dy * xhat (bias / GroupNorm affine grads).
 * fdmi_distill_loss/grad: FD:368-382 (l2 / l1).  fdmi_dmd_loss: FD:459-499 (weight, loss and dL/ds). */
int fdmi_im2col(const void* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                void* stream);
int fdmi_silu(const void* x, void* y, int64_t n, void* stream);
int fdmi_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
int fdmi_colsum(const void* dy, const void* x, const float* stats, float* out0, float* out1, int64_t rows, int C, int HW,
                int G, float eps, void* stream);
int fdmi_transpose2d_pad(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, int64_t rows_pad,
                         void* stream);
int fdmi_pad_cols(const void* src, int cols, void* dst, int cols_pad, int64_t rows, void* stream);
int fdmi_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int fdmi_distill_loss(const float* s, const float* t, int64_t n, int l1, float* out, void* stream);
int fdmi_distill_grad(const float* s, const float* t, int64_t n, int l1, float gscale, float* ds, void* stream);
int fdmi_dmd_loss(const float* s, const float* noisy, const float* real, const float* fake, const float* inv_alpha,
                  const float* msig_alpha, const float* kb, float* w, float* grad, float* loss, int B, int64_t per,
                  void* stream);

/* ---------------- fp32 VALIDATION MODE, op level (csrc/ref32.hip) --------------------------------------------------
 * north_star asks for loss parity with the reference's CPU path to 1e-3 relative; bf16 storage cannot give that on a
 * loss that sits behind nine denoiser evaluations (and neither can the reference's own bf16-mixed run).
 * These are the float twins of the building blocks above: fp32 storage, every contraction on the exact-f32 matrix
 * instruction v_mfma_f32_32x32x2_f32, fp64 normalisation statistics.  Same operand conventions with float data
 * (fdmi_gemm_desc: A / W / rowvec / residual / preact / C are float*, out_f32 implied, no split-K workspace);
 * GroupNorm `stats` hold (mean, rstd) per (sample, group); attention materialises the scores in caller scratch of
 * fdmi_attn_scratch_elems_f32 floats (forward: bwd = 0).  fdmi_unet_config.precision = 1 builds a whole plan on them.   */
int fdmi_gemm_f32(const fdmi_gemm_desc* d, void* stream);
int fdmi_wgrad_tn_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t M, int N1, int N2, float* C, int64_t ldc,
                      void* stream);
int64_t fdmi_attn_scratch_elems_f32(int B, int H, int Sq, int Skv, int bwd);
int fdmi_attn_fwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                      int B, int H, int Sq, int Skv, int d, float scale, float* scratch, int64_t scratch_elems, void* stream);
/* causal self-attention (key j attends query i only for j <= i), forward only: the text encoders of the conditioners
 * (transformers' CLIPTextModel behind embedders/clip/clip_embedder_model.py:10-104) run frozen under no_grad            */
int fdmi_attn_causal_fwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O,
                             int64_t ldo, int B, int H, int S, int d, float scale, float* scratch, int64_t scratch_elems, void* stream);
/* T5 text encoder (transformers' T5EncoderModel behind embedders/t5/t5_embedder_model.py:11-104; frozen, forward only):
 * attention whose scaled scores receive bias[H][Sq][Skv] (the relative-position bias, shared by the samples; may be NULL) and
 * kbias[B][Skv] (additive key mask; may be NULL) before the softmax; T5LayerNorm y = x * rsqrt(mean(x^2) + eps) * w (bf16 / fp32
 * storage, fp32 weight); the element-wise product of the gated feed-forward.                                               */
int fdmi_attn_bias_fwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                           int B, int H, int Sq, int Skv, int d, float scale, const float* bias, const float* kbias, float* scratch,
                           int64_t scratch_elems, void* stream);
int fdmi_rmsnorm(const void* x, const float* w, void* y, int64_t rows, int C, float eps, void* stream);
int fdmi_rmsnorm_f32(const float* x, const float* w, float* y, int64_t rows, int C, float eps, void* stream);
int fdmi_mul(const void* a, const void* b, void* y, int64_t n, void* stream);
int fdmi_mul_f32(const float* a, const float* b, float* y, int64_t n, void* stream);
int fdmi_attn_bwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, const float* dO,
                      int64_t lddo, float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV, int64_t lddv, int B, int H, int Sq,
                      int Skv, int d, float scale, float* scratch, int64_t scratch_elems, void* stream);
int fdmi_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* stats /*[B][G][2] = mean, rstd*/, float* y,
                           int B, int HW, int C, int G, float eps, int silu, void* stream);
int fdmi_groupnorm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* beta, const float* stats, float* dx,
                           int B, int HW, int C, int G, int silu, int accumulate, void* stream);
/* gamma / beta and shift / scale are each optional (NULL): plain, affine, or adaLN-modulated LayerNorm */
int fdmi_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* shift, const float* scale,
                           int64_t mod_ld, int rows_per_batch, float* y, float* stats /*[rows][2] or NULL*/, int64_t rows, int C,
                           float eps, void* stream);
int fdmi_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* scale, int64_t mod_ld,
                           int rows_per_batch, float* dx, int64_t rows, int C, float eps, int accumulate, void* stream);
int fdmi_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int HW, int Cpad, void* stream);
int fdmi_nhwc_to_nchw_f32(const float* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate, void* stream);
int fdmi_silu_f32(const float* x, float* y, int64_t n, void* stream);
int fdmi_silu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int fdmi_gelu_tanh_f32(const float* x, float* y, int64_t n, void* stream);
int fdmi_gelu_tanh_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int fdmi_gate_residual_f32(const float* x, const float* gate, int64_t gate_ld, const float* res, float* y, int64_t rows, int C,
                           int rows_per_batch, void* stream);
int fdmi_batch_colsum_f32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int B, int rows_per_batch,
                          int C, void* stream);
int fdmi_im2col_f32(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                    void* stream);
int fdmi_colsum_f32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int64_t rows, int C, int HW, int G,
                    void* stream);
int fdmi_timestep_embed_f32(const float* t, float* out, int B, int dim, int flip, float shift, void* stream);
int fdmi_pad_cols_f32(const float* src, int cols, float* dst, int cols_pad, int64_t rows, void* stream);

/* ---------------- UNet2DCondition plan: forward + input/LoRA-gradient backward -------------------
 * Replaces DiffusersUNet2DCondWrapper.forward -> UNet2DConditionModel.forward(...).sample
 * (/root/reference/src/flash/models/unets/unet.py:66-119) and its autograd backward.  Architecture
 * hyper-parameters as pinned in examples/train_flash_sd.py:56-114 / train_flash_sdxl.py:66-118.
 * Parameters are registered by their diffusers state_dict names as fp32 device tensors in the
 * diffusers layout (conv OIHW, linear [out][in]); the plan packs them once into its own bf16 operand
 * storage (forward + dgrad layouts).  LoRA A/B (peft semantics, scale alpha/r = 1,
 * examples/train_flash_sd.py:191-200) stay fp32 master tensors owned by the caller; their bf16
 * operand copies are refreshed at every forward and their gradients are ACCUMULATED (+=) into the
 * caller's fp32 grad buffers by backward.                                                          */
typedef struct fdmi_unet fdmi_unet;
typedef struct fdmi_unet_config {
  int32_t in_channels, out_channels;
  int32_t n_levels;             /* 1..4 */
  int32_t block_out[4];
  int32_t down_attn[4];         /* CrossAttnDownBlock2D (1) or DownBlock2D (0) per level */
  int32_t up_attn[4];           /* per up block, in up_blocks order */
  int32_t layers_per_block;
  int32_t tlayers[4];           /* transformer_layers_per_block, per level */
  int32_t heads[4];             /* number of attention heads per level (diffusers' attention_head_dim) */
  int32_t cross_dim;
  int32_t groups; float eps;    /* norm_num_groups, norm_eps */
  int32_t class_embed_dim;      /* projection_class_embeddings_input_dim, 0 = no class embedding */
  int32_t flip_sin_to_cos; float freq_shift;
  int32_t precision;            /* 0: bf16 MFMA with fp32 accumulation (the measured path, the reference's bf16-mixed);
                                   1: fp32 VALIDATION plan -- fp32 storage, every contraction on v_mfma_f32_32x32x2_f32,
                                   fp64 norm statistics (csrc/ref32.hip): the parity gate against the fp32 CPU oracle */
} fdmi_unet_config;
enum { FDMI_UNET_SAVE = 1,          /* record what backward needs (student / GAN backbone) */
       FDMI_UNET_INTERMEDIATE = 2,  /* return_intermediate=True: output the mid-block features */
       FDMI_UNET_INPUT_GRAD = 4,    /* (workspace query only) backward will also produce d/d sample */
       /* Frozen-teacher loops call the plan several times with the SAME context (text embeddings): the
          cross-attention K/V projections (and their head-transposed copies) depend on nothing else.
          CTX_FILL computes them into plan-owned buffers, CTX_REUSE (same B, L, ctx contents; no SAVE, no
          LoRA on those projections) reads them back instead of recomputing. */
       FDMI_UNET_CTX_FILL = 8, FDMI_UNET_CTX_REUSE = 16,
       /* The caller promises that sample / timestep rows B/2.. repeat rows 0..B/2-1 (a classifier-free-guidance batch
          [x | x], only the context halves differ): everything up to the first cross-attention is computed once and
          duplicated.  Ignored with SAVE / INTERMEDIATE / adapter residuals or when the first down block has no attention. */
       FDMI_UNET_CFG_HALVES = 32 };

fdmi_unet* fdmi_unet_create(const fdmi_unet_config* cfg);   /* NULL on error */
void fdmi_unet_destroy(fdmi_unet* u);
int64_t fdmi_unet_num_params(fdmi_unet* u);
int fdmi_unet_param_name(fdmi_unet* u, int64_t i, char* buf, int64_t buflen, int64_t* numel);
int fdmi_unet_set_param(fdmi_unet* u, const char* name, const float* data, int64_t numel, void* stream);
int fdmi_unet_set_lora(fdmi_unet* u, const char* target /* e.g. "...attn1.to_q" */, const float* A /*[r][in]*/,
                       const float* B /*[out][r]*/, float* A_grad, float* B_grad, int rank);
/* Declares that a LoRA pair of this rank WILL be bound to `target` (fdmi_unet_set_lora, same rank): workspace queries then account
 * for its GEMMs and gradient buffers.  Touches no device memory -- a host can size its workspace before the adapters exist.   */
int fdmi_unet_declare_lora(fdmi_unet* u, const char* target, int rank);
int fdmi_unet_ready(fdmi_unet* u);   /* 0 when every parameter has been set */
/* bytes of caller workspace one forward (+ backward when flags has SAVE) needs at this shape */
int64_t fdmi_unet_workspace_bytes(fdmi_unet* u, int B, int H, int W, int L, int flags);
/* sample [B,C,H,W] f32 NCHW, timestep [B] f32, ctx [B,L,cross_dim] f32, class_labels [B,class_embed_dim]
 * f32 or NULL -> out f32 NCHW ([B,out_channels,H,W], or the mid-block features with INTERMEDIATE).
 * `slot` (0..7) names the run state: the workspace handed to a SAVE forward must stay untouched until
 * the matching fdmi_unet_backward(slot).                                                           */
int fdmi_unet_forward(fdmi_unet* u, int slot, const float* sample, const float* timestep, const float* ctx,
                      const float* class_labels, float* out, int B, int H, int W, int L, void* workspace,
                      int64_t workspace_bytes, int flags, void* stream);
/* grad_out: f32 NCHW gradient of the forward's output; grad_sample: f32 NCHW or NULL */
int fdmi_unet_backward(fdmi_unet* u, int slot, const float* grad_out, float* grad_sample, void* stream);
double fdmi_unet_last_flops(fdmi_unet* u);  /* algorithmic MFMA flops of the last forward/backward */
/* GroupNorms of the last forward (or workspace query) whose statistics were accumulated by the producing GEMM's epilogue
 * (developer knob 14, see fdmi_gemm_gn); *total = all GroupNorms of that forward.                                      */
int fdmi_unet_last_gn_epilogue(fdmi_unet* u, int* total);
/* Algorithmic HBM bytes (every operand touched once) that the HBM-bound kernel families of the last forward -- plus its backward
 * once that ran -- move: family 0 GroupNorm reduce, 1 GroupNorm apply, 2 LayerNorm, 3 2-D transposes (LoRA wgrad operands),
 * 4 head transposes (attention V^T / Q^T / K^T / dO^T), 5 strided copies (skip concatenation, gradient scatter), 6 GEGLU backward,
 * 7 2x2 pooling (upsample backward), 8 the split-K finalize pass (fp32 slabs read + bf16 result written); -1.0 for an unknown family.  Works in workspace-query mode (no GPU).               */
double fdmi_unet_last_hbm_bytes(fdmi_unet* u, int family);
/* T2I-adapter residuals (`down_intrablock_additional_residuals`, unet.py:100-106 / flash_diffusion_model.py:208-218) for the
 * NEXT fdmi_unet_forward on this plan, consumed by it: residuals[i] is an f32 NCHW tensor with the shape of down block i's
 * output (NULL entries are skipped), added times `scale` where diffusers adds it: after the last (resnet, attention) pair of
 * a cross-attention block (before its skip connection and downsampler), to the output of an attention-free block.  The
 * adapter is frozen in the reference's recipes, so no gradient flows to the residuals.  n = 0 clears.              */
int fdmi_unet_set_down_residuals(fdmi_unet* u, const float* const* residuals, int n, float scale);

/* ---------------- the frozen convolutional networks beside the denoiser, on the same plan executor ----------------
 * The reference's shipped recipes train with distill_loss_type="lpips" (examples/configs/flash_sd.yaml:20): both outputs' centre
 * crops go through the VAE decoder and the LPIPS-VGG distance (flash_diffusion_model.py:383-397, vae/autoencoderKL.py:63-128),
 * and the canny recipe feeds a T2I-adapter's feature maps to every denoiser call (adapters/t2i_adapter.py:7-26,
 * flash_diffusion_model.py:207-218).  These are plans of the SAME executor as the UNet (conv = implicit-GEMM MFMA kernels,
 * GroupNorm, taped input gradients): fdmi_unet_set_param / _num_params / _param_name / _ready / _destroy work on the handle;
 * parameters carry the upstream state_dict names (diffusers AutoencoderKL: "post_quant_conv.weight", "decoder.conv_in.weight",
 * "decoder.mid_block.attentions.0.to_q.weight", ...; lpips.LPIPS(net="vgg"): "net.slice1.0.weight", "lin0.model.1.weight", ...;
 * diffusers T2IAdapter: "adapter.conv_in.weight", "adapter.body.0.resnets.0.block1.weight", ...).  Weights are frozen: the
 * backward returns the gradient with respect to the input only.                                                        */
enum { FDMI_NET_VAE_DECODER = 1, FDMI_NET_VGG_LPIPS = 3, FDMI_NET_T2I_ADAPTER = 4 };
typedef struct fdmi_net_config {
  int32_t kind;                /* FDMI_NET_* */
  int32_t in_channels;         /* VAE decoder: latent channels (4); LPIPS: 3; adapter: control-image channels */
  int32_t out_channels;        /* VAE decoder: image channels (3) */
  int32_t n_levels;            /* VAE decoder: len(block_out_channels) (<= 4); adapter: len(channels) (<= 4) */
  int32_t block_out[4];        /* VAE: block_out_channels of the AutoencoderKL config (decoder order is the reverse); adapter: channels */
  int32_t layers_per_block;    /* VAE: 2 (the decoder runs layers_per_block + 1 ResNet blocks per level); adapter: num_res_blocks */
  int32_t groups;              /* GroupNorm groups (32) */
  float eps;                   /* GroupNorm eps (1e-6 in AutoencoderKL) */
  int32_t precision;           /* 0 = bf16 MFMA, 1 = fp32 validation kernels */
  float lpips_shift[3], lpips_scale[3];   /* lpips ScalingLayer constants */
  int32_t adapter_downscale;   /* adapter: PixelUnshuffle factor (8: full_adapter, 16: full_adapter_xl) */
  int32_t adapter_xl;          /* adapter: 1 = FullAdapterXL block layout (downsample only in front of block 2) */
} fdmi_net_config;
fdmi_unet* fdmi_net_create(const fdmi_net_config* cfg);
/* workspace bytes of a forward (+ taped backward when flags has FDMI_UNET_SAVE) on [B, in_channels, H, W] */
int64_t fdmi_net_workspace_bytes(fdmi_unet* n, int B, int H, int W, int flags);
/* VAE decoder: x = latents [B, C, H, W] f32 (already divided by the scaling factor) -> out [B, 3, 8H, 8W] f32.
 * LPIPS: x, x2 = images [B, 3, H, W] f32 in [-1, 1] (x2 = the reference side, no gradient) -> out [B] f32 distances.
 * flags: FDMI_UNET_SAVE keeps the tape for fdmi_net_backward on the same slot.                                      */
int fdmi_net_forward(fdmi_unet* n, int slot, const float* x, const float* x2, float* out, int B, int H, int W, void* workspace,
                     int64_t workspace_bytes, int flags, void* stream);
/* grad_out: d loss / d out (VAE: [B, 3, 8H, 8W]; LPIPS: [B]) -> grad_x: d loss / d x, same shape as x */
int fdmi_net_backward(fdmi_unet* n, int slot, const float* grad_out, float* grad_x, void* stream);
/* T2I adapter (frozen, forward only): control image [B, in_channels, H, W] f32 -> n_levels feature maps, outs[i] =
 * [B, channels[i], h_i, w_i] f32 NCHW (caller-allocated; shapes from fdmi_adapter_out_shape)                           */
int fdmi_adapter_out_shape(fdmi_unet* n, int level, int H, int W, int* C, int* h, int* w);
int fdmi_adapter_forward(fdmi_unet* n, int slot, const float* x, float* const* outs, int n_outs, int B, int H, int W, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* ---------------- the transformer denoisers: PixArt-alpha's adaLN-single Transformer2D and SD3's MMDiT ----------------
 * Replace DiffusersTransformer2DWrapper.forward (/root/reference/src/flash/models/transformers/tranformers.py:49-92, with the
 * reference's AdaLayerNormSingle, transformers/utils.py:8-102) and DiffusersSD3Transformer2DWrapper.forward (tranformers.py:113-155)
 * and their autograd backward.  Same handle type and executor as the UNet plan: fdmi_unet_set_param / _num_params / _param_name
 * / _ready / _destroy / _last_flops work on it, and fdmi_unet_set_lora binds a LoRA pair to ANY linear by its module name (peft
 * semantics; the examples' target lists -- examples/train_flash_pixart.py:237-256, train_flash_sd3.py:100-121 -- cover attention,
 * feed-forward, the embedders, the adaLN projections and the patch embedding).  Parameters carry the wrappers' state_dict
 * names ("pos_embed.proj.weight" [D][C][p][p], "adaln_single.linear.weight", "transformer_blocks.0.attn1.to_q.weight",
 * "transformer_blocks.0.scale_shift_table", "scale_shift_table", "proj_out.weight"; MMDiT: "time_text_embed.text_embedder.linear_1.
 * weight", "context_embedder.weight", "transformer_blocks.0.norm1.linear.weight", "transformer_blocks.0.attn.add_q_proj.weight",
 * "norm_out.linear.weight", ...).  The positional table is an INPUT (pos [T][heads * head_dim] f32: the host crops / builds the
 * sin-cos table once per resolution), not plan state.                                                                     */
enum { FDMI_DIT_PIXART = 1, FDMI_DIT_MMDIT = 2 };
typedef struct fdmi_dit_config {
  int32_t kind;              /* FDMI_DIT_* */
  int32_t in_channels, out_channels, patch_size;
  int32_t num_layers, heads, head_dim;   /* inner width D = heads * head_dim */
  int32_t cross_dim;         /* PixArt: cross_attention_dim (= D behind the caption projection) */
  int32_t caption_channels;  /* PixArt: caption_projection input width (0: none, ctx is [B][L][cross_dim]); MMDiT: joint_attention_dim */
  int32_t tdim;              /* width of the sinusoidal timestep embedding (256) */
  int32_t vec_dim;           /* PixArt: projection_class_embeddings_input_dim (0: no vector conditioning); MMDiT: pooled_projection_dim */
  int32_t n_vec;             /* PixArt: num_vector_conditionings with use_concat_vector_conditioning (vector is [B][n_vec * vec_dim]); 0: one add_embedding */
  int32_t attention_bias;    /* PixArt: bias on to_q / to_k / to_v */
  float norm_eps;            /* PixArt: norm_eps of the blocks (the final norm uses 1e-6; the MMDiT uses 1e-6 throughout) */
  int32_t precision;         /* 0 = bf16 MFMA, 1 = fp32 validation kernels */
} fdmi_dit_config;
fdmi_unet* fdmi_dit_create(const fdmi_dit_config* cfg);   /* NULL on error */
/* bytes of caller workspace one forward (+ backward with FDMI_UNET_SAVE, + d sample with FDMI_UNET_INPUT_GRAD) needs on a
 * [B, in_channels, H, W] latent with L context tokens; masked != 0: the forward will carry key lengths */
int64_t fdmi_dit_workspace_bytes(fdmi_unet* d, int B, int H, int W, int L, int masked, int flags);
/* sample [B][in_channels][H][W] f32, timestep [B] f32, ctx [B][L][caption width] f32, vector [B][vector width] f32 (or NULL
 * without vector conditioning), pos [(H/p)(W/p)][D] f32, key_lens: HOST int32 [B] = number of valid (leading) context tokens
 * per sample, or NULL (PixArt's T5 padding mask, tranformers.py:75-77; the MMDiT takes none) -> out [B][out_keep][H][W] f32 =
 * the first out_keep of the out_channels (the wrappers drop the learned-variance half: tranformers.py:91 / :154).
 * flags: FDMI_UNET_SAVE keeps the tape for fdmi_dit_backward on the same slot (workspace untouched until then).            */
int fdmi_dit_forward(fdmi_unet* d, int slot, const float* sample, const float* timestep, const float* ctx, const float* vector,
                     const float* pos, const int32_t* key_lens, float* out, int B, int H, int W, int L, int out_keep,
                     void* workspace, int64_t workspace_bytes, int flags, void* stream);
/* grad_out [B][out_keep][H][W] f32 -> LoRA gradients ACCUMULATED (+=) into the bound buffers; grad_sample [B][in_channels][H][W]
 * f32 or NULL */
int fdmi_dit_backward(fdmi_unet* d, int slot, const float* grad_out, float* grad_sample, void* stream);

/* The frozen teacher's guidance loop over a transformer denoiser as ONE call -- fdmi_teacher_loop (below) for the fdmi_dit_* plans:
 * per step  eps = dit([x | x], t_i, [ctx_cond | ctx_uncond], [vector_cond | vector_uncond])  on the 2B batch, then
 *           x0 = a0 x + a1 eps_cond + a2 eps_uncond ;  x = a3 x + a4 x0 + a5 x0_prev          with a = coeffs[i][0..5] (HOST).
 * The PixArt recipe's DPM-Solver++ loop (flash_diffusion_model.py:288-324) has this form as the UNet's does; the SD3 recipe's
 * flow-matching Euler loop (flash_diffusion_sd3_model.py:282-314: x += dt (g eps_c + (1 - g) eps_u)) is a0 = 0, a1 = g dt, a2 =
 * (1 - g) dt, a3 = a4 = 1, a5 = 0.  x [B][in_channels][H][W] f32 is advanced in place (eps = the first in_channels output
 * channels); key_lens2: HOST int32 [2B] or NULL; workspace as fdmi_dit_workspace_bytes(d, 2B, H, W, L, masked, 0).          */
int64_t fdmi_dit_teacher_loop_scratch_bytes(fdmi_unet* d, int B, int H, int W);
int fdmi_dit_teacher_loop(fdmi_unet* d, int slot, float* x, const float* timesteps, int n, const float* ctx2, const float* vector2,
                          const float* pos, const int32_t* key_lens2, const float* coeffs, int B, int H, int W, int L,
                          void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream);

/* The frozen teacher's classifier-free-guidance loop (flash_diffusion_model.py:288-324) as ONE call: for each of the n
 * steps  eps = unet([x | x], t_i, [ctx_cond | ctx_uncond])  (one forward on the 2B batch; the cross-attention K/V of the
 * constant context are computed at step 0 and reused),
 *        x0 = a0 x + a1 eps_cond + a2 eps_uncond   (guidance folded into the x0 prediction, FD:316-319)
 *        x  = a3 x + a4 x0 + a5 x0_prev            (the scheduler's update, FD:322-324; x0_prev = previous step's x0),
 * with a = coeffs[i][0..5], HOST arrays `timesteps` [n] and `coeffs` [n][6] computed by the caller in fp32 exactly as the
 * scheduler does (DPM-Solver++ 2M, DDPM, Euler all have this form).  x [B,C,H,W] f32 is updated in place; ctx2 [2B,L,cross_dim]
 * and class_labels2 [2B,class_embed_dim] (or NULL) hold the cond rows first, then the uncond rows.  `workspace` as for a
 * forward at batch 2B (fdmi_unet_workspace_bytes(u, 2B, H, W, L, FDMI_UNET_CTX_FILL)), `scratch` of
 * fdmi_teacher_loop_scratch_bytes bytes; nothing is retained after the call.                                              */
int64_t fdmi_teacher_loop_scratch_bytes(fdmi_unet* u, int B, int H, int W);
int fdmi_teacher_loop(fdmi_unet* u, int slot, float* x, const float* timesteps, int n, const float* ctx2,
                      const float* class_labels2, const float* coeffs, int B, int H, int W, int L, void* workspace,
                      int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream);

/* ---------------- data-parallel gradient exchange (RCCL over xGMI) --------------------------------
 * One process per GPU.  Rank 0 calls fdmi_comm_unique_id and hands the 128 bytes to the other ranks out of band (env,
 * file, socket); every rank then calls fdmi_allreduce_init once.  fdmi_allreduce is the ONE collective of a training
 * step: the in-place SUM of the flat LoRA gradient (fold 1/world into fdmi_adamw's grad_scale), enqueued on `stream`.
 * librccl.so is bound at run time, on first use.  (A PyTorch host uses torch.distributed's "nccl" backend -- the same
 * RCCL -- instead: flash_diffusion_amd/trainer.py.)  Replaces Lightning DDP's gradient all-reduce,
 * examples/train_flash_sd.py:383-386.                                                             */
enum { FDMI_F32 = 0, FDMI_BF16 = 1 };
int fdmi_comm_unique_id(void* out128);
int fdmi_allreduce_init(int rank, int world, const void* uid128);
int fdmi_allreduce(void* buf, int64_t count, int dtype, void* stream);
int fdmi_allreduce_world(void);      /* 0 before init */
int fdmi_allreduce_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* FDMI_H_ */
