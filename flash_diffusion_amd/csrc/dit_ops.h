// Launchers of the layout / per-sample-vector kernels the transformer-denoiser plans add (dit.hip); `f32` selects the fp32
// validation plans' element type (float instead of bf16) behind the void pointers.
#pragma once
#include "ops.h"

// x [B][C][H][W] f32 -> patches [B * (H/p) * (W/p)][C * p * p] (column = (c, py, px)) and the gradient's way back
int launch_patchify(const float* x, void* patches, int B, int C, int H, int W, int p, int f32, hipStream_t st);
int launch_patchify_bwd(const void* dpatches, float* gx, int B, int C, int H, int W, int p, int f32, hipStream_t st);
// tokens [B * (H/p) * (W/p)][p * p * oc] (column = (py, px, c)) -> img [B][keep][H][W] f32, keep <= oc; and d tokens from d img
int launch_unpatchify(const void* tokens, float* img, int B, int oc, int keep, int H, int W, int p, int f32, hipStream_t st);
int launch_unpatchify_bwd(const float* gimg, void* dtokens, int B, int oc, int keep, int H, int W, int p, int f32, hipStream_t st);
// dst[b][i] = table[i] + src[b][i % src_cols], i < n
int launch_add_table(const void* src, int src_cols, const float* table, void* dst, int B, int n, int f32, hipStream_t st);
// dst[b][col0 + c] (+)= src[b][c], c < C (src fp32 [B][C])
int launch_vec_grad_add(const float* src, void* dst, int64_t ldd, int col0, int B, int C, int accumulate, int f32, hipStream_t st);
// KV [B * (T + L)][2 D] = per sample [k | v of the T latent tokens ; k | v of the L text tokens] out of the fused projections
// yx [B * T][3 D], yc [B * L][3 D] (bf16 plans)
int launch_kv_join(const bf16_t* yx, const bf16_t* yc, bf16_t* kv, int B, int T, int L, int D, hipStream_t st);
