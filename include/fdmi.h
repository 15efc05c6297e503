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

/* ---------------- GEMM / implicit-GEMM convolution (bf16 MFMA, fp32 accumulate) ----------------
 * out[M,N] = A[M,K] * W[N,K]^T ; epilogue v = alpha*acc + bias[n] + rowvec[m/rows_per_batch][n]
 *            + residual[m][n]; act (0 none, 1 SiLU, 2 GEGLU on 16-wide (value|gate) interleave).
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
} fdmi_gemm_desc;
int fdmi_gemm(const fdmi_gemm_desc* d, void* stream);

/* ---------------- normalisation (NHWC / token-major bf16, fp32 statistics) -------------------- */
int fdmi_groupnorm_fwd(const void* x, const float* gamma, const float* beta, float* stats /*[B][G][2]*/,
                       void* y, int B, int HW, int C, int G, float eps, int silu, void* stream);
int fdmi_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                       const float* stats, float* bstats, void* dx, int B, int HW, int C, int G, float eps,
                       int silu, int accumulate, void* stream);
int fdmi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int C,
                       float eps, void* stream);
int fdmi_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, int64_t rows, int C,
                       float eps, int accumulate, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* FDMI_H_ */
