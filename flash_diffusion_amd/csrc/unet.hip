// UNet2DCondition plan executor for libfdmi.so: the native replacement of
//   DiffusersUNet2DCondWrapper.forward -> UNet2DConditionModel.forward(...).sample
//   (/root/reference/src/flash/models/unets/unet.py:66-119) and of its autograd backward.
//
// One plan = one denoiser (architecture hyper-parameters as pinned in
// /root/reference/examples/train_flash_sd.py:56-114 / train_flash_sdxl.py:66-118, weights registered by
// their diffusers state_dict names and packed once into bf16 MFMA operand layouts: forward [N][K] and
// the dgrad layout).  forward() walks the network launching the hand-written kernels on NHWC bf16
// activations bump-allocated from a caller-owned workspace; when `save` is set it records a tape of
// backward closures (dgrad everywhere -- base weights are frozen -- and wgrad for the LoRA A/B
// matrices only), which backward() replays in reverse.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <vector>

#include "../../include/fdmi.h"
#include "wgrad.h"
#include "ops.h"
#include "dit_ops.h"

namespace {

enum { HBM_GN_REDUCE = 0, HBM_GN_APPLY, HBM_LN, HBM_TRANSPOSE2D, HBM_TRANSPOSE_HEADS, HBM_COPY2D, HBM_GEGLU_BWD, HBM_POOL,
       HBM_SPLITK, FDMI_HBM_FAMILIES };

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = false;
  void* alloc(size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    if (!dry && a + bytes > cap) return nullptr;
    off = a + bytes;
    if (off > peak) peak = off;
    return dry ? (void*)(uintptr_t)256 : (void*)(base + a);
  }
};

struct T {  // NHWC / token-major bf16 activation [rows][cols] (+ lazily allocated gradient)
  bf16_t* p = nullptr;
  bf16_t* g = nullptr;
  int64_t rows = 0;
  int cols = 0;
  int B = 0, H = 0, W = 0;
  bool ginit = false;
  float* gn = nullptr;   // GroupNorm sums [B][G][2] of this tensor, left by the producing GEMM's epilogue
  // column-slice view of another tensor (the q / k / v parts of a fused projection): row stride and gradient are the parent's
  T* parent = nullptr;
  int col0 = 0;
  int64_t ldv = 0;
  bool own = false;   // p was bump-allocated for this tensor (Run::mk): its storage may be recycled once its producer's backward ran
  int64_t goff = 0;   // element offset of this view inside the parent's gradient buffer (column views: col0; row views: row0 * cols)
  int64_t ld() const { return ldv ? ldv : cols; }
  // a channel concatenation that is never materialised (Exec::cat): columns [0, c1) live at p (row stride c1), columns
  // [c1, cols) at p2 (row stride cols - c1).  Only groupnorm() and linear_w() (the ResNet block's norm1 and 1x1 shortcut, the two
  // readers of the up path's [h | skip]) accept such a tensor; its gradient buffer g is an ordinary [rows][cols] one.
  bf16_t* p2 = nullptr;
  int c1 = 0;
  // fp32 master copy of the values (bf16 plans, round 5): the transformer denoisers' residual stream.  `p` stays the bf16 shadow
  // every GEMM operand / backward kernel reads; LayerNorm and the residual adds read and write p32 (Run::mk32, Exec::linear_w)
  float* p32 = nullptr;
};

struct Weight {
  bf16_t* w = nullptr;   // forward operand [N][K]
  bf16_t* wt = nullptr;  // dgrad operand: linear [K][N]; conv [Cin][KH][KW][Cout_pad]
  float* bias = nullptr;
  int N = 0, K = 0;      // forward GEMM dims (conv: K = KH*KW*Cin_pad)
  int Cin = 0, Cout = 0, KH = 1, KW = 1, Cin_pad = 0, Cout_pad = 0;
  bool geglu = false;
  int set = 0;           // bit0 weight, bit1 bias
  bool has_bias = true;
};
struct Norm {
  float* gamma = nullptr;
  float* beta = nullptr;
  int C = 0, set = 0;
};
struct Lora {
  const float *A_master = nullptr, *B_master = nullptr;
  float *A_grad = nullptr, *B_grad = nullptr;
  bf16_t *A = nullptr, *AT = nullptr, *B = nullptr, *BT = nullptr;
  // where the refresh writes the bf16 A / A^T copies: the own buffers above, or slices of a block's fused [3r][in] / [in][3r]
  bf16_t *A_dst = nullptr, *AT_dst = nullptr;
  int AT_ld = 0;
  int r = 0, in = 0, out = 0;
  // rank of the bf16 operand copies A [rp][in] and B^T [rp][out]: ranks below 128 are padded with zero rows so that the rank-r
  // products t = x A^T and dt = dy B run as N = 128 problems on the 256-row LDS-DMA kernels instead of the small-tile fallback
  // (C4: 574 launches x 84 us per step on `gemm_kernel<128,64>`); t / dt are then [M][rp] buffers whose first r columns everything
  // else reads (leading dimension rp).  fp32 plans: rp = r.
  int rp = 0;
  bool on = false;
  // LoRA up-projection as extra K tiles of the base GEMM (round 3): Wc [out][in + r] = [W | B] so that y = [x | t] Wc^T is ONE
  // launch (no read-modify-write pass over y), and Wtc [in][out + r] = [W^T | A^T] so that dx = [dy | dt] Wtc^T is one launch as
  // well.  The base halves are copied from the packed weight when it changes, the LoRA halves by the per-forward refresh.
  bf16_t *Wc = nullptr, *Wtc = nullptr;
};
struct LinearW {
  Weight w;
  Lora lora;
};
struct ResnetW {
  Norm n1, n2;
  Weight c1, c2, sc;
  bool has_sc = false;
  int cin = 0, cout = 0, temb_off = 0;
};
struct AttnW { LinearW q, k, v, o; };
struct TBlockW {
  Norm ln1, ln2, ln3;
  AttnW a1, a2;
  Weight ff1, ff2;
  // attn1's q / k / v projections as ONE GEMM (x is read once): plan-owned concatenated operands qkv.w [3C][C], qkv.wt [C][3C]
  // copied from the three packed weights, and -- LoRA on all three -- the concatenated A operands A3 [3r][C], AT3 [C][3r]
  Weight qkv;
  bf16_t *A3 = nullptr, *AT3 = nullptr;
  int fused_r = 0;
  // ... and, with the LoRA up-projections folded in: Wc3 [3C][C + 3r] = [Wqkv | blockdiag(B_q, B_k, B_v)], Wtc3 [C][3C + 3r] =
  // [Wqkv^T | A_q^T A_k^T A_v^T]
  bf16_t *Wc3 = nullptr, *Wtc3 = nullptr;
  // ranks below 128: blockdiag(B_q^T, B_k^T, B_v^T) [3r][3C], so that dt3 = dy3 . blockdiag is ONE N = 3r GEMM on the LDS-DMA kernels
  // instead of three N = r launches of the small-tile fallback (the zero blocks cost 3x the flops of a product that is a few
  // per cent of the layer's; refreshed with the other LoRA copies)
  bf16_t* BT3 = nullptr;
  // cross-attention K / V / V^T of a fixed context (FDMI_UNET_CTX_FILL / _REUSE), plan-owned
  bf16_t *ck = nullptr, *cv = nullptr, *cvt = nullptr;
  int64_t c_rows = 0, c_vt = 0;
  bool c_valid = false;
};
struct TransformerW {
  Norm gn;
  Weight pin, pout;
  std::vector<std::unique_ptr<TBlockW>> blocks;
  int heads = 0, C = 0;
};
struct StageW {
  std::vector<std::unique_ptr<ResnetW>> res;
  std::vector<std::unique_ptr<TransformerW>> attn;
  bool has_attn = false, has_resample = false;
  Weight resample;  // downsample (stride 2) or upsample conv
};

enum SlotKind { S_CONV_W, S_LIN_W, S_BIAS, S_GAMMA, S_BETA, S_TEMB_W, S_TEMB_B, S_VEC };
struct Slot {
  SlotKind kind;
  Weight* w = nullptr;
  Norm* n = nullptr;
  int64_t numel = 0;
  int row_off = 0, rows = 0;  // S_TEMB_*
  float** vec = nullptr;      // S_VEC: a plain fp32 device vector (the LPIPS 1x1 "lin" weights)
};

// ---- the frozen convolutional networks that sit beside the denoiser in the step, on the same executor ----
enum { NET_UNET = 0, NET_VAE_DECODER = 1, NET_VGG_LPIPS = 3, NET_T2I_ADAPTER = 4, NET_DIT_PIXART = 5, NET_DIT_MMDIT = 6 };
struct NetVae {   // diffusers AutoencoderKL: post_quant_conv + Decoder (vae/autoencoderKL.py:63-128 calls .decode)
  Weight post_quant, conv_in, conv_out;
  Norm norm_out, mid_gn;
  std::unique_ptr<ResnetW> mid_r0, mid_r1;
  Weight q, k, v, o;
  std::vector<std::unique_ptr<StageW>> up;
};
struct NetVgg {   // lpips.LPIPS(net="vgg"): torchvision VGG16 features up to relu5_3 + five non-negative 1x1 "lin" layers
  Weight conv[13];
  float* lin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};
struct AdapterBlockW {   // diffusers AdapterBlock: [AvgPool2d(2)] [1x1 in_conv] + num_res x (conv3x3, ReLU, conv1x1, + x)
  bool down = false, has_in = false;
  Weight in_conv;
  std::vector<std::unique_ptr<std::pair<Weight, Weight>>> res;
};
struct NetAdapter {
  Weight conv_in;
  std::vector<std::unique_ptr<AdapterBlockW>> body;
};

// ---- the transformer denoisers (dit_plan.h): PixArt-alpha's adaLN-single Transformer2D and SD3's MMDiT ----
struct DitEmbW { LinearW l1, l2; };   // linear_2(act(linear_1(x))): timestep / vector / text embedders, the caption projection
struct DitBlockW {                     // BasicTransformerBlock with norm_type = "ada_norm_single"
  TBlockW at;                          // a1 = attn1 (q / k / v fused into one GEMM), a2 = attn2 (cross-attention over the caption)
  LinearW ff1, ff2;
  float* table = nullptr;              // scale_shift_table [6][D]
};
struct MmBlockW {                      // JointTransformerBlock
  bool pre_only = false;               // the last block: the context stream only feeds the attention
  LinearW n1, n1c;                     // norm1.linear / norm1_context.linear (AdaLayerNormZero / ...Continuous projections)
  TBlockW x, c;                        // a1 = (to_q, to_k, to_v, to_out.0) / (add_q_proj, add_k_proj, add_v_proj, to_add_out)
  LinearW ff1, ff2, ffc1, ffc2;
};
struct NetDit {
  TransformerW tw;                     // (C, heads) for the shared fused-operand builders
  LinearW patch;                       // pos_embed.proj: a k = stride convolution IS a linear map on the folded patches
  DitEmbW temb;
  std::vector<std::unique_ptr<DitEmbW>> addemb;   // PixArt: adaln_single.add_embedding (one, or one per concatenated vector)
  LinearW ada;                         // PixArt: adaln_single.linear -> the six modulation vectors shared by every block
  DitEmbW cap;                         // PixArt: caption_projection
  DitEmbW text;                        // MMDiT: time_text_embed.text_embedder
  LinearW ctx_emb, norm_out;           // MMDiT: context_embedder, norm_out.linear
  float* table = nullptr;              // PixArt: the final scale_shift_table [2][D]
  LinearW proj_out;
  std::vector<std::unique_ptr<DitBlockW>> blocks;
  std::vector<std::unique_ptr<MmBlockW>> mblocks;
};

struct Exec;
struct Run {
  Arena arena;
  std::deque<T> tensors;
  std::vector<std::function<int(Exec&)>> tape;
  bool save = false;
  int es = 2;            // bytes per activation element: 2 = bf16 (the measured path), 4 = fp32 validation plan
  float* sc32 = nullptr; // fp32 plans: scratch of the materialised attention scores (grown on demand, reused serially)
  int64_t sc32_elems = 0;
  hipStream_t st = nullptr;
  // pool of fp32 accumulators (GroupNorm statistics, forward and backward) cleared by ONE memset per forward
  char* zpool = nullptr;
  size_t zoff = 0, zcap = 0;
  float* zalloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (!zpool || zoff + bytes > zcap) return nullptr;
    float* p = (float*)(zpool + zoff);
    zoff += bytes;
    return p;
  }
  // Backward-time recycling (the transformer plans, dit_plan.h): once the tape entry that PRODUCED a tensor has run, nobody reads
  // its value or its gradient again (every consumer's entry was recorded later, so it ran earlier) -- both buffers go to a free
  // list keyed by size and come back as the gradient / scratch buffers of the layers below.  The identical layers of a
  // transformer make the reuse near-perfect: a saved run needs about its forward footprint, not forward + backward.
  bool recycle = false;
  std::multimap<size_t, void*> freelist;
  std::vector<std::pair<size_t, T*>> tape_outs;        // (tape index, tensor it produced)
  std::vector<std::pair<void*, size_t>> scoped;        // scratch of the running tape entry
  static size_t rnd(size_t b) { return (b + 255) & ~(size_t)255; }
  void* ralloc(size_t bytes) {
    bytes = rnd(bytes);
    if (recycle) {
      auto it = freelist.find(bytes);
      if (it != freelist.end()) {
        void* p = it->second;
        freelist.erase(it);
        return p;
      }
    }
    return arena.alloc(bytes);
  }
  void rfree(void* p, size_t bytes) {
    if (recycle && p) freelist.emplace(rnd(bytes), p);
  }
  void* tmp(size_t bytes) {   // scratch that dies with the running tape entry (plain bump allocation outside a recycling backward)
    if (!recycle) return arena.alloc(bytes);
    void* p = ralloc(bytes);
    if (p) scoped.push_back({p, bytes});
    return p;
  }
  void mark_out(std::initializer_list<T*> ts) {   // call right after tape.push_back: the tensors that entry produced
    for (T* t : ts)
      if (t) tape_outs.push_back({tape.size() - 1, t});
  }
  T* x0 = nullptr;   // NHWC input (grad wrt the sample)
  T* out = nullptr;  // output tensor
  int outC = 0;
  T* mk(int64_t rows, int cols, int B = 0, int H = 0, int W = 0) {
    tensors.emplace_back();
    T* t = &tensors.back();
    t->rows = rows; t->cols = cols; t->B = B; t->H = H; t->W = W;
    t->p = (bf16_t*)arena.alloc(rnd((size_t)rows * cols * es));
    t->own = true;
    return t->p ? t : nullptr;
  }
  float* mk32(T* t) {   // the fp32 master of a bf16 tensor (T::p32)
    t->p32 = (float*)arena.alloc(rnd((size_t)t->rows * t->cols * 4));
    return t->p32;
  }
  // tensor header over caller-provided storage (no arena allocation)
  T* wrap(bf16_t* p, int64_t rows, int cols) {
    tensors.emplace_back();
    T* t = &tensors.back();
    t->rows = rows; t->cols = cols; t->p = p;
    return t;
  }
  // columns col0 .. col0 + cols of `parent` (bf16 plans)
  T* view(T* parent, int col0, int cols) {
    tensors.emplace_back();
    T* t = &tensors.back();
    t->rows = parent->rows; t->cols = cols; t->B = parent->B; t->H = parent->H; t->W = parent->W;
    t->p = parent->p + col0; t->parent = parent; t->col0 = col0; t->ldv = parent->ld(); t->goff = col0;
    return t;
  }
  // rows row0 .. row0 + rows of a (compact) `parent`, all columns: one sample's tokens (either precision)
  T* rowview(T* parent, int64_t row0, int64_t rows) {
    tensors.emplace_back();
    T* t = &tensors.back();
    t->rows = rows; t->cols = parent->cols; t->B = 1;
    t->p = (bf16_t*)((char*)parent->p + (size_t)row0 * parent->cols * es); t->parent = parent; t->goff = row0 * parent->cols;
    return t;
  }
  bool dry() const { return arena.dry; }
  const float* gvec = nullptr;   // NET_VGG_LPIPS backward: d loss / d distance [B] (device, fp32)
  std::vector<T*> outs;          // NET_T2I_ADAPTER: the feature maps of the last forward
  int dit_geom[5] = {0, 0, 0, 0, 0};   // transformer denoisers: B, H, W (latent), channels kept of the output, sample channels
};

}  // namespace

struct fdmi_unet {
  int kind = NET_UNET;
  fdmi_net_config ncfg{};
  std::unique_ptr<NetVae> vae;
  std::unique_ptr<NetVgg> vgg;
  std::unique_ptr<NetAdapter> adp;
  std::unique_ptr<NetDit> dit;
  fdmi_dit_config dcfg{};
  // T2I-adapter residuals for the NEXT forward (one f32 NCHW tensor per down block, consumed once): UW:100-106
  std::vector<const float*> down_res;
  float down_res_scale = 1.f;
  fdmi_unet_config cfg;
  bool f32 = false;      // cfg.precision == 1: fp32 validation plan (ref32.hip kernels, fp32 storage; T::p / Weight::w then hold floats)
  int nl = 0, temb_ch = 0, temb_total = 0;
  Weight conv_in, conv_out, te1, te2, ce1, ce2, temb_proj;
  Norm norm_out;
  std::vector<std::unique_ptr<StageW>> down, up;
  std::unique_ptr<ResnetW> mid_r0, mid_r1;
  std::unique_ptr<TransformerW> mid_attn;
  std::map<std::string, Slot> slots;
  std::map<std::string, LinearW*> lora_targets;
  std::vector<void*> owned;  // hipMalloc'ed packed weights
  std::vector<Lora*> loras;
  CastJob* cast_jobs = nullptr;   // device table of the LoRA refresh (rebuilt when a master pointer changes)
  int n_cast_jobs = 0;
  bool cast_dirty = true;
  bool fused_dirty = true;   // a q / k / v weight changed: the concatenated operands are rebuilt at the next forward
  int cast_mode = -1;        // value of A/B switch 17 the LoRA refresh table was built for
  Run runs[8];
  double last_flops = 0;     // algorithmic MFMA flops of the last forward/backward call
  int last_gn = 0, last_gn_epi = 0;  // GroupNorms of the last forward / how many took their sums from a GEMM epilogue
  // algorithmic HBM bytes of the HBM-bound kernel families of the last forward (+ backward): every operand touched once
  // (fdmi_unet_last_hbm_bytes; DESIGN.md section 5 prices the measured kernel times against these)
  double hbm[FDMI_HBM_FAMILIES] = {0};

  ~fdmi_unet() {
    for (void* p : owned) (void)hipFree(p);
    if (cast_jobs) (void)hipFree(cast_jobs);
  }
};

namespace {

#define RET_IF(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
#define NULL_IF(x) do { if ((x)) return nullptr; } while (0)
#define FDMI_CHECK_NULL(cond, msg) do { if (!(cond)) { fdmi_set_error(msg); return nullptr; } } while (0)

// ---------------------------------------------------------------------------------------------
// weight packing kernels (init time)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_w(bf16_t* o, float v) { *o = f2bf(v); }
__device__ __forceinline__ void put_w(float* o, float v) { *o = v; }
// conv OIHW f32 -> out[rows][KH][KW][cpad] (bf16, or f32 for a validation plan); transpose=0: rows=O, ch=I ; transpose=1: rows=I, ch=O
template <typename TO>
__global__ void pack_conv_kernel(const float* w, TO* out, int O, int I, int KH, int KW, int cpad,
                                 int transpose) {
  const int rows = transpose ? I : O, ch = transpose ? O : I;
  const int64_t total = (int64_t)rows * KH * KW * cpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cpad);
    int64_t t = i / cpad;
    const int kx = (int)(t % KW); t /= KW;
    const int ky = (int)(t % KH); t /= KH;
    const int r = (int)t;
    float v = 0.f;
    if (c < ch) {
      const int o = transpose ? c : r, ii = transpose ? r : c;
      v = w[(((int64_t)o * I + ii) * KH + ky) * KW + kx];
    }
    put_w(out + i, v);
  }
}
// linear [N][K] f32 -> w[N][K] (rows optionally GEGLU-permuted) and, when asked for, wt[K][N]
template <typename TO>
__global__ void pack_linear_kernel(const float* w, TO* out, TO* outT, int N, int K, int geglu) {
  const int64_t total = (int64_t)N * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const int n = (int)(i / K);  // packed row
    int src = n;
    if (geglu) {
      const int blk = n >> 5, off = n & 31;
      src = off < 16 ? blk * 16 + off : N / 2 + blk * 16 + off - 16;
    }
    const float v = w[(int64_t)src * K + k];
    put_w(out + i, v);
    if (outT) put_w(outT + (int64_t)k * N + n, v);
  }
}
__global__ void pack_vec_kernel(const float* b, float* out, int N, int geglu) {
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
    int src = n;
    if (geglu) {
      const int blk = n >> 5, off = n & 31;
      src = off < 16 ? blk * 16 + off : N / 2 + blk * 16 + off - 16;
    }
    out[n] = b[src];
  }
}
template <typename TO>
__global__ void cvt_rows_kernel(const float* x, TO* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) put_w(y + i, x[i]);
}
static inline int gridfor(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------------------------
struct Builder {
  fdmi_unet* U;
  // pad_out: the GEMM produces Cout_pad columns (zero weights / bias in the pad) so that the result can feed another conv
  void conv(const std::string& name, Weight& w, int cout, int cin, int k, bool bias = true, bool pad_out = false) {
    w.Cin = cin; w.Cout = cout; w.KH = w.KW = k;
    w.Cin_pad = (cin + 7) & ~7; w.Cout_pad = (cout + 7) & ~7;
    w.N = pad_out ? w.Cout_pad : cout; w.K = k * k * w.Cin_pad; w.has_bias = bias;
    U->slots[name + ".weight"] = Slot{S_CONV_W, &w, nullptr, (int64_t)cout * cin * k * k};
    if (bias) U->slots[name + ".bias"] = Slot{S_BIAS, &w, nullptr, cout};
  }
  void linear(const std::string& name, Weight& w, int out, int in, bool bias, bool geglu = false) {
    w.Cin = in; w.Cout = out; w.N = out; w.K = in; w.has_bias = bias; w.geglu = geglu;
    U->slots[name + ".weight"] = Slot{S_LIN_W, &w, nullptr, (int64_t)out * in};
    if (bias) U->slots[name + ".bias"] = Slot{S_BIAS, &w, nullptr, out};
  }
  void norm(const std::string& name, Norm& n, int C) {
    n.C = C;
    U->slots[name + ".weight"] = Slot{S_GAMMA, nullptr, &n, C};
    U->slots[name + ".bias"] = Slot{S_BETA, nullptr, &n, C};
  }
  void lin_lora(const std::string& name, LinearW& l, int out, int in, bool bias) {
    linear(name, l.w, out, in, bias);
    U->lora_targets[name] = &l;
  }
  std::unique_ptr<ResnetW> resnet(const std::string& name, int cin, int cout) {
    auto r = std::make_unique<ResnetW>();
    r->cin = cin; r->cout = cout;
    norm(name + ".norm1", r->n1, cin);
    conv(name + ".conv1", r->c1, cout, cin, 3);
    r->temb_off = U->temb_total;
    Slot sw{S_TEMB_W, &U->temb_proj, nullptr, (int64_t)cout * U->temb_ch};
    sw.row_off = r->temb_off; sw.rows = cout;
    U->slots[name + ".time_emb_proj.weight"] = sw;
    Slot sb{S_TEMB_B, &U->temb_proj, nullptr, cout};
    sb.row_off = r->temb_off; sb.rows = cout;
    U->slots[name + ".time_emb_proj.bias"] = sb;
    U->temb_total += cout;
    norm(name + ".norm2", r->n2, cout);
    conv(name + ".conv2", r->c2, cout, cout, 3);
    r->has_sc = cin != cout;
    if (r->has_sc) conv(name + ".conv_shortcut", r->sc, cout, cin, 1);
    return r;
  }
  // ResnetBlock2D without a time embedding (AutoencoderKL: temb_channels=None)
  std::unique_ptr<ResnetW> resnet_plain(const std::string& name, int cin, int cout) {
    auto r = std::make_unique<ResnetW>();
    r->cin = cin; r->cout = cout; r->temb_off = -1;
    norm(name + ".norm1", r->n1, cin);
    conv(name + ".conv1", r->c1, cout, cin, 3);
    norm(name + ".norm2", r->n2, cout);
    conv(name + ".conv2", r->c2, cout, cout, 3);
    r->has_sc = cin != cout;
    if (r->has_sc) conv(name + ".conv_shortcut", r->sc, cout, cin, 1);
    return r;
  }
  void vec(const std::string& name, float** dst, int n) {
    Slot sl{S_VEC, nullptr, nullptr, n};
    sl.vec = dst;
    U->slots[name] = sl;
  }
  std::unique_ptr<TransformerW> transformer(const std::string& name, int C, int heads, int layers) {
    auto t = std::make_unique<TransformerW>();
    t->C = C; t->heads = heads;
    norm(name + ".norm", t->gn, C);
    linear(name + ".proj_in", t->pin, C, C, true);
    linear(name + ".proj_out", t->pout, C, C, true);
    const int cd = U->cfg.cross_dim;
    for (int k = 0; k < layers; ++k) {
      auto b = std::make_unique<TBlockW>();
      const std::string bn = name + ".transformer_blocks." + std::to_string(k);
      norm(bn + ".norm1", b->ln1, C);
      norm(bn + ".norm2", b->ln2, C);
      norm(bn + ".norm3", b->ln3, C);
      lin_lora(bn + ".attn1.to_q", b->a1.q, C, C, false);
      lin_lora(bn + ".attn1.to_k", b->a1.k, C, C, false);
      lin_lora(bn + ".attn1.to_v", b->a1.v, C, C, false);
      lin_lora(bn + ".attn1.to_out.0", b->a1.o, C, C, true);
      lin_lora(bn + ".attn2.to_q", b->a2.q, C, C, false);
      lin_lora(bn + ".attn2.to_k", b->a2.k, C, cd, false);
      lin_lora(bn + ".attn2.to_v", b->a2.v, C, cd, false);
      lin_lora(bn + ".attn2.to_out.0", b->a2.o, C, C, true);
      linear(bn + ".ff.net.0.proj", b->ff1, 8 * C, C, true, true);
      linear(bn + ".ff.net.2", b->ff2, C, 4 * C, true);
      t->blocks.push_back(std::move(b));
    }
    return t;
  }
};

int build_plan(fdmi_unet* U) {
  const fdmi_unet_config& c = U->cfg;
  FDMI_CHECK(c.n_levels >= 1 && c.n_levels <= 4, "unet: n_levels must be 1..4");
  U->nl = c.n_levels;
  U->temb_ch = c.block_out[0] * 4;
  Builder b{U};
  b.conv("conv_in", U->conv_in, c.block_out[0], c.in_channels, 3);
  b.linear("time_embedding.linear_1", U->te1, U->temb_ch, c.block_out[0], true);
  b.linear("time_embedding.linear_2", U->te2, U->temb_ch, U->temb_ch, true);
  if (c.class_embed_dim > 0) {
    b.linear("class_embedding.linear_1", U->ce1, U->temb_ch, c.class_embed_dim, true);
    b.linear("class_embedding.linear_2", U->ce2, U->temb_ch, U->temb_ch, true);
  }
  int out_ch = c.block_out[0];
  for (int i = 0; i < U->nl; ++i) {
    auto s = std::make_unique<StageW>();
    const int in_ch = out_ch;
    out_ch = c.block_out[i];
    s->has_attn = c.down_attn[i] != 0;
    const std::string bn = "down_blocks." + std::to_string(i);
    for (int j = 0; j < c.layers_per_block; ++j) {
      s->res.push_back(b.resnet(bn + ".resnets." + std::to_string(j), j == 0 ? in_ch : out_ch, out_ch));
      if (s->has_attn)
        s->attn.push_back(b.transformer(bn + ".attentions." + std::to_string(j), out_ch, c.heads[i], c.tlayers[i]));
    }
    s->has_resample = i != U->nl - 1;
    if (s->has_resample) b.conv(bn + ".downsamplers.0.conv", s->resample, out_ch, out_ch, 3);
    U->down.push_back(std::move(s));
  }
  const int cm = c.block_out[U->nl - 1];
  U->mid_r0 = b.resnet("mid_block.resnets.0", cm, cm);
  U->mid_attn = b.transformer("mid_block.attentions.0", cm, c.heads[U->nl - 1], c.tlayers[U->nl - 1]);
  U->mid_r1 = b.resnet("mid_block.resnets.1", cm, cm);
  out_ch = cm;
  for (int i = 0; i < U->nl; ++i) {
    auto s = std::make_unique<StageW>();
    const int ri = U->nl - 1 - i;
    const int prev = out_ch;
    out_ch = c.block_out[ri];
    const int in_ch = c.block_out[ri - 1 >= 0 ? ri - 1 : 0];
    s->has_attn = c.up_attn[i] != 0;
    const std::string bn = "up_blocks." + std::to_string(i);
    const int n = c.layers_per_block + 1;
    for (int j = 0; j < n; ++j) {
      const int skip = j == n - 1 ? in_ch : out_ch;
      const int rin = j == 0 ? prev : out_ch;
      s->res.push_back(b.resnet(bn + ".resnets." + std::to_string(j), rin + skip, out_ch));
      if (s->has_attn)
        s->attn.push_back(b.transformer(bn + ".attentions." + std::to_string(j), out_ch, c.heads[ri], c.tlayers[ri]));
    }
    s->has_resample = i != U->nl - 1;
    if (s->has_resample) b.conv(bn + ".upsamplers.0.conv", s->resample, out_ch, out_ch, 3);
    U->up.push_back(std::move(s));
  }
  b.norm("conv_norm_out", U->norm_out, c.block_out[0]);
  b.conv("conv_out", U->conv_out, c.out_channels, c.block_out[0], 3);
  // batched time_emb_proj: one [sum Cout][temb_ch] operand
  U->temb_proj.N = U->temb_total;
  U->temb_proj.K = U->temb_ch;
  return 0;
}

template <typename Tp>
int dmalloc(fdmi_unet* U, Tp** p, size_t n) {
  void* q = nullptr;
  FDMI_HIP(hipMalloc(&q, n * sizeof(Tp)));
  // The clear runs on the NULL stream, which does not order itself against the caller's (non-blocking) streams: without the
  // wait below it can land AFTER the first kernels that fill the buffer on the caller's stream -- seen once the teacher loop
  // and the student's forward ran on two streams (a cross-attention K/V cache zeroed behind the GEMM that had just written it).
  FDMI_HIP(hipMemset(q, 0, n * sizeof(Tp)));
  FDMI_HIP(hipStreamSynchronize(nullptr));
  U->owned.push_back(q);
  *p = (Tp*)q;
  return 0;
}

int set_param(fdmi_unet* U, const std::string& name, const float* src, int64_t numel, hipStream_t st) {
  auto it = U->slots.find(name);
  FDMI_CHECK(it != U->slots.end(), "unet: unknown parameter '" + name + "'");
  Slot& s = it->second;
  FDMI_CHECK(s.numel == numel, "unet: parameter '" + name + "' has " + std::to_string(numel) +
                                   " elements, expected " + std::to_string(s.numel));
  switch (s.kind) {
    case S_CONV_W: {
      Weight& w = *s.w;
      const size_t k = U->f32 ? 2 : 1;   // (dmalloc counts bf16 elements: an fp32 plan takes twice as many)
      if (!w.w) {
        RET_IF(dmalloc(U, &w.w, k * (w.N > w.Cout ? w.N : w.Cout) * w.KH * w.KW * w.Cin_pad));   // (pad_out: zero rows)
        RET_IF(dmalloc(U, &w.wt, k * w.Cin * w.KH * w.KW * w.Cout_pad));
      }
      const int g1 = gridfor((int64_t)w.Cout * w.K), g2 = gridfor((int64_t)w.Cin * w.KH * w.KW * w.Cout_pad);
      if (U->f32) {
        hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(g1), dim3(256), 0, st, src, (float*)w.w, w.Cout, w.Cin, w.KH, w.KW, w.Cin_pad, 0);
        hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(g2), dim3(256), 0, st, src, (float*)w.wt, w.Cout, w.Cin, w.KH, w.KW, w.Cout_pad, 1);
      } else {
        hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(g1), dim3(256), 0, st, src, w.w, w.Cout, w.Cin, w.KH, w.KW, w.Cin_pad, 0);
        hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(g2), dim3(256), 0, st, src, w.wt, w.Cout, w.Cin, w.KH, w.KW, w.Cout_pad, 1);
      }
      w.set |= 1;
      break;
    }
    case S_LIN_W: {
      Weight& w = *s.w;
      if (!w.w) {
        RET_IF(dmalloc(U, &w.w, (size_t)w.N * w.K * (U->f32 ? 2 : 1)));
        if (!U->f32) RET_IF(dmalloc(U, &w.wt, (size_t)w.N * w.K));   // (an fp32 plan reads the one copy with strides)
      }
      U->fused_dirty = true;
      if (U->f32)
        hipLaunchKernelGGL(pack_linear_kernel<float>, dim3(gridfor((int64_t)w.N * w.K)), dim3(256), 0, st, src, (float*)w.w,
                           (float*)nullptr, w.N, w.K, w.geglu ? 1 : 0);
      else
        hipLaunchKernelGGL(pack_linear_kernel<bf16_t>, dim3(gridfor((int64_t)w.N * w.K)), dim3(256), 0, st, src, w.w, w.wt,
                           w.N, w.K, w.geglu ? 1 : 0);
      w.set |= 1;
      break;
    }
    case S_BIAS: {
      Weight& w = *s.w;
      if (!w.bias) RET_IF(dmalloc(U, &w.bias, (size_t)w.N));
      const int nb = (w.Cout > 0 && w.Cout < w.N) ? w.Cout : w.N;   // (a conv with padded output columns: the pad stays zero)
      hipLaunchKernelGGL(pack_vec_kernel, dim3(gridfor(nb)), dim3(256), 0, st, src, w.bias, nb, w.geglu ? 1 : 0);
      w.set |= 2;
      U->fused_dirty = true;   // (a fused q / k / v GEMM carries the three biases concatenated)
      break;
    }
    case S_GAMMA:
    case S_BETA: {
      Norm& n = *s.n;
      float** dst = s.kind == S_GAMMA ? &n.gamma : &n.beta;
      if (!*dst) RET_IF(dmalloc(U, dst, (size_t)n.C));
      hipLaunchKernelGGL(pack_vec_kernel, dim3(gridfor(n.C)), dim3(256), 0, st, src, *dst, n.C, 0);
      n.set |= s.kind == S_GAMMA ? 1 : 2;
      break;
    }
    case S_VEC: {
      if (!*s.vec) RET_IF(dmalloc(U, s.vec, (size_t)s.numel));
      hipLaunchKernelGGL(pack_vec_kernel, dim3(gridfor((int)s.numel)), dim3(256), 0, st, src, *s.vec, (int)s.numel, 0);
      break;
    }
    case S_TEMB_W:
    case S_TEMB_B: {
      Weight& w = *s.w;
      if (!w.w) {
        RET_IF(dmalloc(U, &w.w, (size_t)w.N * w.K * (U->f32 ? 2 : 1)));
        RET_IF(dmalloc(U, &w.bias, (size_t)w.N));
      }
      if (s.kind == S_TEMB_W) {
        if (U->f32)
          hipLaunchKernelGGL(cvt_rows_kernel<float>, dim3(gridfor((int64_t)s.rows * w.K)), dim3(256), 0, st, src,
                             (float*)w.w + (size_t)s.row_off * w.K, (int64_t)s.rows * w.K);
        else
          hipLaunchKernelGGL(cvt_rows_kernel<bf16_t>, dim3(gridfor((int64_t)s.rows * w.K)), dim3(256), 0, st, src,
                             w.w + (size_t)s.row_off * w.K, (int64_t)s.rows * w.K);
      } else {
        hipLaunchKernelGGL(pack_vec_kernel, dim3(gridfor(s.rows)), dim3(256), 0, st, src, w.bias + s.row_off,
                           s.rows, 0);
      }
      break;
    }
  }
  FDMI_HIP(hipGetLastError());
  return 0;
}


// ---------------------------------------------------------------------------------------------
// ops with tape (forward + recorded input-gradient closure)
// ---------------------------------------------------------------------------------------------
static bool plan_log() {
  static const bool on = getenv("FDMI_PLAN_LOG") != nullptr;
  return on;
}

template <typename F>
void for_each_tblock(fdmi_unet* U, F fn) {
  auto stage = [&](StageW& s) {
    for (auto& t : s.attn)
      for (auto& b : t->blocks) fn(*t, *b);
  };
  for (auto& s : U->down) stage(*s);
  if (U->mid_attn)
    for (auto& b : U->mid_attn->blocks) fn(*U->mid_attn, *b);
  for (auto& s : U->up) stage(*s);
  if (U->dit) {   // the transformer denoisers' blocks hold the same fused-projection record(s)
    for (auto& b : U->dit->blocks) fn(U->dit->tw, b->at);
    for (auto& b : U->dit->mblocks) { fn(U->dit->tw, b->x); fn(U->dit->tw, b->c); }
  }
}
// attn1's three projections can run as one GEMM when none of them or all of them carry a LoRA of one rank (bf16 plans;
// A/B switch 17 = 1 keeps the three separate launches)
static bool qkv_fusable(const fdmi_unet* U, const TBlockW& b) {
  if (U->f32 || fdmi_tune_get(17)) return false;
  const Lora &q = b.a1.q.lora, &k = b.a1.k.lora, &v = b.a1.v.lora;
  if (!q.on && !k.on && !v.on) return true;
  return q.on && k.on && v.on && q.r == k.r && q.r == v.r;
}
// LoRA up-projection folded into the base GEMM as extra K tiles (bf16 plans; A/B switch 31 = 1 keeps the separate launches):
// the seam of the two-segment operands sits at in (forward) / out (dgrad), so both and the rank must be multiples of 64
static bool lora_foldable(const fdmi_unet* U, const Lora& l) {
  return !U->f32 && !fdmi_tune_get(31) && l.on && (l.r % 64) == 0 && (l.in % 64) == 0 && (l.out % 64) == 0 && l.out >= 128 &&
         l.in >= 128;   // (the folded dgrad is a GEMM with N = in: gemm_a2_ok wants N >= 128 for the forward AND the dgrad shape)
}
static int copy2d_dev(bf16_t* dst, int64_t ldd, const bf16_t* src, int64_t lds, int rows, int cols, hipStream_t st) {
  FDMI_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * 2, src, (size_t)lds * 2, (size_t)cols * 2, rows, hipMemcpyDeviceToDevice, st));
  return 0;
}
// base halves of the folded LoRA operands: Wc [out][in + r] <- W, Wtc [in][out + r] <- W^T (the packed copies)
int build_lora_folded(fdmi_unet* U, hipStream_t st) {
  for (auto& kv : U->lora_targets) {
    LinearW& L = *kv.second;
    Lora& l = L.lora;
    if (!lora_foldable(U, l) || !L.w.w || !L.w.wt) continue;
    if (!l.Wc) {
      RET_IF(dmalloc(U, &l.Wc, (size_t)l.out * (l.in + l.r)));
      RET_IF(dmalloc(U, &l.Wtc, (size_t)l.in * (l.out + l.r)));
      U->cast_dirty = true;   // the refresh table gains the second destinations
    }
    RET_IF(copy2d_dev(l.Wc, l.in + l.r, L.w.w, l.in, l.out, l.in, st));
    RET_IF(copy2d_dev(l.Wtc, l.out + l.r, L.w.wt, l.out, l.in, l.out, st));
  }
  return 0;
}
// (re)build the concatenated operands of every fusable block from the packed per-projection weights
int build_fused_operands(fdmi_unet* U, hipStream_t st) {
  int rc = build_lora_folded(U, st);
  if (rc) return rc;
  for_each_tblock(U, [&](TransformerW& t, TBlockW& b) {
    if (rc || U->f32) return;
    const int C = t.C;
    const Weight *w[3] = {&b.a1.q.w, &b.a1.k.w, &b.a1.v.w};
    if (!w[0]->w || !w[1]->w || !w[2]->w) return;
    if (!b.qkv.w) {
      b.qkv.N = 3 * C; b.qkv.K = C; b.qkv.has_bias = false;
      if ((rc = dmalloc(U, &b.qkv.w, (size_t)3 * C * C))) return;
      if ((rc = dmalloc(U, &b.qkv.wt, (size_t)3 * C * C))) return;
    }
    for (int s = 0; s < 3 && !rc; ++s) {
      if (hipMemcpyAsync(b.qkv.w + (size_t)s * C * C, w[s]->w, (size_t)C * C * 2, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpy2DAsync(b.qkv.wt + (size_t)s * C, (size_t)3 * C * 2, w[s]->wt, (size_t)C * 2, (size_t)C * 2, C,
                           hipMemcpyDeviceToDevice, st) != hipSuccess) {
        fdmi_set_error("unet: copying the fused q/k/v operands failed");
        rc = -2;
      }
    }
    if (w[0]->bias && w[1]->bias && w[2]->bias) {   // (attention_bias = True: the transformer denoisers)
      if (!b.qkv.bias && (rc = dmalloc(U, &b.qkv.bias, (size_t)3 * C))) return;
      for (int s = 0; s < 3 && !rc; ++s)
        if (hipMemcpyAsync(b.qkv.bias + (size_t)s * C, w[s]->bias, (size_t)C * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
          fdmi_set_error("unet: copying the fused q/k/v bias failed");
          rc = -2;
        }
      if (rc) return;
    }
    // ... and with the three LoRA up-projections folded in (zero blocks off the diagonal: dmalloc clears)
    const Lora& lq = b.a1.q.lora;
    if (!rc && qkv_fusable(U, b) && lq.on && lora_foldable(U, lq) && lora_foldable(U, b.a1.k.lora) && lora_foldable(U, b.a1.v.lora)) {
      const int r = lq.r, ldc = C + 3 * r, ldt = 3 * C + 3 * r;
      if (!b.Wc3) {
        if ((rc = dmalloc(U, &b.Wc3, (size_t)3 * C * ldc)) || (rc = dmalloc(U, &b.Wtc3, (size_t)C * ldt))) return;
        U->cast_dirty = true;
      }
      if ((rc = copy2d_dev(b.Wc3, ldc, b.qkv.w, C, 3 * C, C, st))) return;
      if ((rc = copy2d_dev(b.Wtc3, ldt, b.qkv.wt, 3 * C, C, 3 * C, st))) return;
    }
  });
  if (!rc) U->fused_dirty = false;
  return rc;
}

struct Exec {
  fdmi_unet* U;
  Run& R;
  hipStream_t st;
  double flops = 0;
  int ctx_mode = 0;  // 0: none, 1: fill the cross-attention K/V cache, 2: reuse it (FDMI_UNET_CTX_*)
  bool gn_epi = false;  // GroupNorm statistics in the producing GEMM's epilogue (see want_gn)
  int splitk_max_rows = 0;  // > 0: only problems of at most this many rows may split K (the transformer denoisers, dit_plan.h)

  // ---- precision dispatch: an fp32 validation plan stores floats behind the same (opaque) bf16_t* handles and runs the
  // ref32.hip kernels; everything below picks the kernel family by U->f32 ----
  bool f32() const { return U->f32; }
  int es() const { return U->f32 ? 4 : 2; }
  bf16_t* off(const bf16_t* p, int64_t elems) const { return (bf16_t*)((char*)p + elems * es()); }
  static float* F(const bf16_t* p) { return (float*)p; }
  int l_copy2d(const bf16_t* src, int64_t lds, int sc0, bf16_t* dst, int64_t ldd, int dc0, int64_t rows, int cols, int acc) {
    return f32() ? launch_copy2d32(F(src), lds, sc0, F(dst), ldd, dc0, rows, cols, acc, st)
                 : launch_copy2d(src, lds, sc0, dst, ldd, dc0, rows, cols, acc, st);
  }
  float* attn_scratch(int Bn, int H, int Sq, int Skv, int bwd) {   // fp32 plans: room for the materialised scores
    const int64_t need = attn32_scratch_elems(Bn, H, Sq, Skv, bwd);
    if (need > R.sc32_elems) {
      R.sc32 = (float*)R.arena.alloc((size_t)need * 4);
      R.sc32_elems = R.sc32 ? need : 0;
    }
    return R.sc32;
  }
  // C = A W^T with W read through strides: element (n, k) at W[n * ldw + k * w_sk] (fp32 plans: dgrad operands without copies)
  int gemm_rows_strided(const bf16_t* A, int64_t lda, int64_t M, const void* W, int N, int K, int64_t ldw, int64_t w_sk,
                        bf16_t* C, int64_t ldc, const bf16_t* residual, int64_t ldr) {
    GemmArgs a = rows_args(A, lda, M, (const bf16_t*)W, N, K, nullptr, C, ldc, residual, ldr);
    a.ldw = ldw; a.w_sk = w_sk;
    return gemm(a);
  }

  bf16_t* grad_of(T* t) {  // lazily allocate the gradient buffer
    if (t->parent) {      // a column slice: the same columns of the parent's gradient
      bf16_t* pg = grad_of(t->parent);
      t->g = pg ? off(pg, t->goff) : nullptr;
      return t->g;
    }
    if (!t->g) t->g = (bf16_t*)R.ralloc((size_t)t->rows * t->cols * es());
    return t->g;
  }
  int gemm(GemmArgs& a) {
    flops += gemm_flops(a);
    if (f32()) {   // exact-f32 MFMA kernel, no split-K bookkeeping
      a.f32 = 1; a.out_f32 = 1;
      if (R.dry()) return 0;
      return launch_gemm(a, st);
    }
    if (R.dry() && plan_log()) {   // developer aid: FDMI_PLAN_LOG=1 lists every GEMM / conv of a workspace-query walk (no GPU needed)
      GemmArgs q = a;
      if (!q.accum_atomic && q.splitk == 1 && gemm_ws_bytes(q)) q.splitk = 0;
      const GemmPlan p = plan_gemm(q, true);
      fprintf(stderr, "PLANGEMM mode=%d M=%d N=%d K=%d act=%d res=%d dgrad=%d atomic=%d kernel=%d BM=%d BN=%d splitk=%d\n", a.mode, a.M, a.N,
              a.K, a.act, a.residual ? 1 : 0, a.dgrad, a.accum_atomic, p.big, p.big ? 256 : p.BM, p.BN, p.splitk);
    }
    if (!a.accum_atomic && a.splitk == 1 && (!splitk_max_rows || a.M <= splitk_max_rows)) {  // let the launcher split K when the tile grid under-fills the chip
      const size_t wsb = gemm_ws_bytes(a);
      if (wsb) {
        a.ws = (float*)R.tmp(wsb);
        FDMI_CHECK(a.ws, "unet: workspace exhausted (split-K)");
        a.splitk = 0;
        U->hbm[HBM_SPLITK] += (double)wsb + 2.0 * a.M * a.N;   // what the finalize pass reads (fp32 slabs) and writes (bf16)
      }
    }
    if (R.dry()) return 0;
    return launch_gemm(a, st);
  }
  // The consumer of this GEMM's output is a GroupNorm: let the epilogue accumulate its (sum, sum of squares) per (sample, group)
  // into an accumulator of the pre-zeroed pool, so that groupnorm() skips the reduction pass over the tensor.  Only problems the
  // 256-row kernels run without split-K qualify (gemm_gn_ok); everything else keeps the reduce kernel.  Call right before gemm(a).
  void want_gn(GemmArgs& a, int rows_per_sample) {
    const int G = U->cfg.groups;
    if (!gn_epi || rows_per_sample <= 0 || a.N % G || a.accum_atomic || a.splitk != 1 || gemm_ws_bytes(a)) return;
    a.gn_stats = (float*)(uintptr_t)256;  // (placeholder for the host-side query)
    a.gn_rows = rows_per_sample; a.gn_G = G; a.gn_cpg = a.N / G;
    float* s = gemm_gn_ok(a, false) ? R.zalloc((size_t)(a.M / rows_per_sample) * G * 2 * sizeof(float)) : nullptr;
    a.gn_stats = s;
  }
  // ---- C = A W^T style helper; accumulate=true adds into C (residual = C) ----
  static GemmArgs rows_args(const bf16_t* A, int64_t lda, int64_t M, const bf16_t* W, int N, int K, const float* bias,
                            bf16_t* C, int64_t ldc, const bf16_t* residual, int64_t ldr, int act = ACT_NONE,
                            bf16_t* preact = nullptr) {
    GemmArgs a;
    a.M = (int)M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias;
    a.C = C; a.ldc = ldc; a.residual = residual; a.ldr = ldr; a.act = act;
    a.preact = preact; a.ldp = N;
    return a;
  }
  int gemm_rows(const bf16_t* A, int64_t lda, int64_t M, const bf16_t* W, int N, int K, const float* bias,
                bf16_t* C, int64_t ldc, const bf16_t* residual, int64_t ldr, int act = ACT_NONE,
                bf16_t* preact = nullptr) {
    GemmArgs a = rows_args(A, lda, M, W, N, K, bias, C, ldc, residual, ldr, act, preact);
    return gemm(a);
  }

  int lora_refresh() {
    if (R.dry()) return 0;
    for (Lora* l : U->loras) FDMI_CHECK(l->A && l->A_master, "unet: a declared LoRA was never bound (fdmi_unet_set_lora)");
    if (f32()) return 0;   // (an fp32 plan reads the fp32 masters directly)
    if (U->cast_mode != fdmi_tune_get(17)) U->cast_dirty = true;
    if (U->cast_dirty) {
      U->cast_mode = fdmi_tune_get(17);  // (re)build the tile table of the one-launch refresh
      std::vector<CastJob> jobs;
      auto add = [&](const float* src, bf16_t* dst, bf16_t* dstT, int rows, int cols, int ldT, bf16_t* dst2 = nullptr, int ld2 = 0,
                     bf16_t* dstT2 = nullptr, int ldT2 = 0) {
        for (int r0 = 0; r0 < rows; r0 += 64)
          for (int c0 = 0; c0 < cols; c0 += 64)
            jobs.push_back(CastJob{src, dst, dstT, rows, cols, r0, c0, ldT, dst2, ld2, dstT2, ldT2});
      };
      for (Lora* l : U->loras) { l->A_dst = l->A; l->AT_dst = l->AT; l->AT_ld = 0; }
      // blocks whose attn1 projections run fused: the three A copies land in one [3r][in] / [in][3r] pair
      int rc = 0;
      for_each_tblock(U, [&](TransformerW& t, TBlockW& b) {
        Lora* l3[3] = {&b.a1.q.lora, &b.a1.k.lora, &b.a1.v.lora};
        if (rc || !l3[0]->on || !qkv_fusable(U, b)) return;
        const int r = l3[0]->r, in = l3[0]->in;
        if (!b.A3 || b.fused_r != r) {
          if ((rc = dmalloc(U, &b.A3, (size_t)3 * r * in)) || (rc = dmalloc(U, &b.AT3, (size_t)3 * r * in))) return;
          b.fused_r = r;
        }
        for (int s = 0; s < 3; ++s) {
          l3[s]->A_dst = b.A3 + (size_t)s * r * in;
          l3[s]->AT_dst = b.AT3 + (size_t)s * r;
          l3[s]->AT_ld = 3 * r;
        }
        if (r < 128 && !b.BT3) rc = dmalloc(U, &b.BT3, (size_t)3 * r * 3 * in);
      });
      RET_IF(rc);
      std::map<const Lora*, std::pair<TBlockW*, int>> in_bt3;
      for_each_tblock(U, [&](TransformerW& t, TBlockW& b) {
        if (!b.BT3 || !qkv_fusable(U, b) || !b.a1.q.lora.on) return;
        Lora* l3[3] = {&b.a1.q.lora, &b.a1.k.lora, &b.a1.v.lora};
        for (int sI = 0; sI < 3; ++sI) in_bt3[l3[sI]] = {&b, sI};
      });
      // the LoRA halves of the folded operands ride along as second destinations: A^T [in][r] -> Wtc[:, out:], B [out][r] ->
      // Wc[:, in:]; for a fused q / k / v block slice s lands at Wtc3[:, 3C + s r :] and Wc3[s C :, C + s r :]
      std::map<const Lora*, std::pair<TBlockW*, int>> in_qkv;
      for_each_tblock(U, [&](TransformerW& t, TBlockW& b) {
        if (!b.Wc3 || !qkv_fusable(U, b)) return;
        Lora* l3[3] = {&b.a1.q.lora, &b.a1.k.lora, &b.a1.v.lora};
        for (int sI = 0; sI < 3; ++sI) in_qkv[l3[sI]] = {&b, sI};
      });
      for (Lora* l : U->loras) {
        bf16_t *a2 = nullptr, *b2 = nullptr;
        int a2ld = 0, b2ld = 0;
        auto q = in_qkv.find(l);
        if (q != in_qkv.end()) {
          TBlockW& b = *q->second.first;
          const int sI = q->second.second, C = l->in, r = l->r;
          a2 = b.Wtc3 + 3 * C + sI * r; a2ld = 3 * C + 3 * r;
          b2 = b.Wc3 + (size_t)sI * C * (C + 3 * r) + C + sI * r; b2ld = C + 3 * r;
        } else if (l->Wc) {
          a2 = l->Wtc + l->out; a2ld = l->out + l->r;
          b2 = l->Wc + l->in; b2ld = l->in + l->r;
        }
        bf16_t* bt2 = nullptr;   // B^T [r][out] also into its diagonal block of the block's [3r][3C] operand
        int bt2ld = 0;
        auto q3 = in_bt3.find(l);
        if (q3 != in_bt3.end()) {
          const int sI = q3->second.second, C = l->out;
          bt2 = q3->second.first->BT3 + (size_t)sI * l->r * 3 * C + (size_t)sI * C;
          bt2ld = 3 * C;
        }
        add(l->A_master, l->A_dst, l->AT_dst, l->r, l->in, l->AT_ld, nullptr, 0, a2, a2ld);
        add(l->B_master, l->B, l->BT, l->out, l->r, 0, b2, b2ld, bt2, bt2ld);
      }
      if (U->cast_jobs) FDMI_HIP(hipFree(U->cast_jobs));
      U->cast_jobs = nullptr;
      FDMI_HIP(hipMalloc((void**)&U->cast_jobs, jobs.size() * sizeof(CastJob)));
      FDMI_HIP(hipMemcpy(U->cast_jobs, jobs.data(), jobs.size() * sizeof(CastJob), hipMemcpyHostToDevice));
      U->n_cast_jobs = (int)jobs.size();
      U->cast_dirty = false;
    }
    return launch_cast_transpose_jobs(U->cast_jobs, U->n_cast_jobs, st);
  }

  // y = x W^T + b (+ residual) (+ LoRA)
  // shadow = false: nobody reads the bf16 value of this fp32-stream output (its consumer is a LayerNorm reading the fp32 master): a run
  // without a tape then skips the bf16 store (2 of every 10 bytes of such a launch, which is HBM-bound)
  T* linear(T* x, LinearW& L, T* residual = nullptr, bool shadow = true) { return linear_w(x, L.w, residual, L.lora.on ? &L.lora : nullptr, true, 0, false, shadow); }
  // gn_rows > 0: the output feeds a GroupNorm over samples of gn_rows rows (want_gn)
  // out32 / a residual that carries an fp32 master (T::p32): the output gets one too -- v = ... + residual32 in fp32, stored to
  // y->p32 beside the bf16 y->p (GemmArgs::residual32 / C32: the transformer denoisers' residual stream)
  T* linear_w(T* x, Weight& w, T* residual = nullptr, Lora* lo = nullptr, bool need_dx = true, int gn_rows = 0, bool out32 = false, bool shadow = true) {
    T* y = R.mk(x->rows, w.N, x->B, x->H, x->W);
    if (!y) return nullptr;
    const bool r32 = !f32() && (out32 || (residual && residual->p32));
    if (r32 && !R.mk32(y)) return nullptr;
    // LoRA up-projection folded into the base GEMM: y = [x | t] [W | B]^T (+ bias, residual) -- needs the folded operands
    // (lora_foldable, built by build_lora_folded) and a problem the two-segment loaders take (M >= 256)
    const bool fold = lo && lora_foldable(U, *lo) && (R.dry() || lo->Wc != nullptr) && x->rows >= 256 && !x->p2;
    T* t = nullptr;
    const int rp = lo ? ((!f32() && x->rows >= 256 && !x->p2 && !fdmi_tune_get(44)) ? lo->rp : lo->r) : 0;   // (A/B switch 44 = 1: N = r)
    if (lo) {
      t = R.mk(x->rows, rp);
      if (!t) return nullptr;
      const bf16_t* LA = f32() ? (const bf16_t*)lo->A_master : lo->A;
      NULL_IF(gemm_rows(x->p, x->cols, x->rows, LA, rp, lo->in, nullptr, t->p, rp, nullptr, 0));
      flops -= 2.0 * x->rows * (double)(rp - lo->r) * lo->in;   // (the zero rows are not algorithmic work)
    }
    {
      GemmArgs a = rows_args(x->p, x->cols, x->rows, w.w, w.N, w.K, w.bias, y->p, w.N, residual ? residual->p : nullptr,
                             residual ? residual->cols : 0);
      if (x->p2) {   // x = [h | skip], never materialised (cat)
        a.lda = x->c1; a.A2 = x->p2; a.lda2 = x->cols - x->c1; a.K1 = x->c1;
      }
      if (fold) {
        a.W = lo->Wc; a.K = w.K + lo->r; a.ldw = a.K;
        a.A2 = t->p; a.lda2 = rp; a.K1 = w.K;
      }
      if (gn_rows > 0 && !lo) want_gn(a, gn_rows);   // (a LoRA delta is added to y afterwards: the sums would be stale)
      if (r32) {
        a.C32 = y->p32; a.ldc32 = w.N;
        if (residual && residual->p32) { a.residual = nullptr; a.residual32 = residual->p32; a.ldr32 = residual->cols; }
        if (!shadow && !R.save && !lo && !fdmi_tune_get(53)) a.C = nullptr;   // (A/B switch 53 = 1: always store the shadow)
      }
      NULL_IF(gemm(a));
      y->gn = a.gn_stats;
    }
    if (lo && !fold) {
      const bf16_t* LB = f32() ? (const bf16_t*)lo->B_master : lo->B;
      GemmArgs a2 = rows_args(t->p, rp, x->rows, LB, lo->out, lo->r, nullptr, y->p, w.N, y->p, w.N);
      if (r32) { a2.residual = nullptr; a2.residual32 = y->p32; a2.ldr32 = w.N; a2.C32 = y->p32; a2.ldc32 = w.N; }
      NULL_IF(gemm(a2));
    }
    if (R.save) {
      R.tape.push_back([x, y, residual, lo, t, need_dx, fold, &w, rp](Exec& E) -> int {
        if (!y->g) return 0;  // no gradient reached this output
        if (residual) RET_IF(E.add_grad(residual, y->g, y->cols, 0, y->cols));
        if (need_dx && !fold) {
          bf16_t* dx = E.grad_of(x);
          FDMI_CHECK(dx, "unet: workspace exhausted (grad)");
          if (E.f32())   // dx = dy W: the one [N][K] copy read k-strided
            RET_IF(E.gemm_rows_strided(y->g, w.N, x->rows, w.w, w.K, w.N, 1, w.K, dx, x->cols, x->ginit ? dx : nullptr, x->cols));
          else
            RET_IF(E.gemm_rows(y->g, w.N, x->rows, w.wt, w.K, w.N, nullptr, dx, x->cols, x->ginit ? dx : nullptr,
                               x->cols));
          x->ginit = true;
        }
        if (lo) {
          // dt = dy B ; dB += dy^T t ; dA += dt^T x ; dx += dt A
          T* dt = E.R.mk(x->rows, rp);
          FDMI_CHECK(dt, "unet: workspace exhausted (lora)");
          if (E.f32()) {
            RET_IF(E.gemm_rows_strided(y->g, w.N, x->rows, lo->B_master, lo->r, lo->out, 1, lo->r, dt->p, lo->r, nullptr, 0));
          } else {
            RET_IF(E.gemm_rows(y->g, w.N, x->rows, lo->BT, rp, lo->out, nullptr, dt->p, rp, nullptr, 0));
            E.flops -= 2.0 * x->rows * (double)(rp - lo->r) * lo->out;
          }
          // dB += dy^T t and dA += dt^T x straight from the row-major operands (wgrad.hip): no transposed copies
          E.flops += 2.0 * x->rows * lo->r * ((double)lo->out + lo->in);
          if (!E.R.dry()) {
            if (E.f32()) {
              RET_IF(launch_wgrad_tn32(F(y->g), w.N, F(t->p), lo->r, x->rows, lo->out, lo->r, lo->B_grad, lo->r, E.st));
              RET_IF(launch_wgrad_tn32(F(dt->p), lo->r, F(x->p), x->cols, x->rows, lo->r, lo->in, lo->A_grad, lo->in, E.st));
            } else {
              const WgradProblem wp[2] = {{y->g, w.N, t->p, rp, x->rows, lo->out, lo->r, lo->B_grad, lo->r},
                                          {dt->p, rp, x->p, x->cols, x->rows, lo->r, lo->in, lo->A_grad, lo->in}};
              RET_IF(launch_wgrad_tn_group(wp, 2, E.st));   // (dB, dA) in one launch
            }
          }
          if (need_dx && fold) {   // dx (+)= [dy | dt] [W^T | A^T]^T: the base input gradient and the LoRA one in ONE launch
            bf16_t* dx = E.grad_of(x);
            FDMI_CHECK(dx, "unet: workspace exhausted (grad)");
            GemmArgs d = rows_args(y->g, w.N, x->rows, lo->Wtc, w.K, w.N + lo->r, nullptr, dx, x->cols, x->ginit ? dx : nullptr,
                                   x->cols);
            d.ldw = w.N + lo->r; d.A2 = dt->p; d.lda2 = rp; d.K1 = w.N;
            RET_IF(E.gemm(d));
            x->ginit = true;
          } else if (need_dx) {
            bf16_t* dx = E.grad_of(x);
            if (E.f32())
              RET_IF(E.gemm_rows_strided(dt->p, lo->r, x->rows, lo->A_master, lo->in, lo->r, 1, lo->in, dx, x->cols, dx, x->cols));
            else
              RET_IF(E.gemm_rows(dt->p, rp, x->rows, lo->AT, lo->in, lo->r, nullptr, dx, x->cols, dx, x->cols));
          }
        }
        return 0;
      });
      R.mark_out({y, t});
    }
    return y;
  }

  // attn1's q, k, v projections of one input as ONE GEMM: y [M, 3C] = x Wqkv^T (+ per-slice LoRA deltas); x is read once
  // instead of three times, the backward's input gradient is one GEMM over K = 3C.  bf16 plans (qkv_fusable).
  T* linear_qkv(T* x, TBlockW& b, int C) {
    Lora* l3[3] = {&b.a1.q.lora, &b.a1.k.lora, &b.a1.v.lora};
    const bool lora = l3[0]->on;
    const int r = lora ? l3[0]->r : 0;
    T* y = R.mk(x->rows, 3 * C, x->B, x->H, x->W);
    if (!y) return nullptr;
    // the three LoRA up-projections folded in: y = [x | t_q t_k t_v] [Wqkv | blockdiag(B_q, B_k, B_v)]^T, one launch
    const bool fold = lora && x->rows >= 256 && lora_foldable(U, *l3[0]) && lora_foldable(U, *l3[1]) && lora_foldable(U, *l3[2]) &&
                      (R.dry() || b.Wc3 != nullptr);
    T* t3 = nullptr;
    if (lora) {
      t3 = R.mk(x->rows, 3 * r);
      if (!t3) return nullptr;
      NULL_IF(gemm_rows(x->p, x->cols, x->rows, b.A3, 3 * r, C, nullptr, t3->p, 3 * r, nullptr, 0));
    }
    if (fold) {
      GemmArgs a = rows_args(x->p, x->cols, x->rows, b.Wc3, 3 * C, C + 3 * r, b.qkv.bias, y->p, 3 * C, nullptr, 0);
      a.ldw = C + 3 * r; a.A2 = t3->p; a.lda2 = 3 * r; a.K1 = C;
      NULL_IF(gemm(a));
      flops -= 2.0 * x->rows * (3.0 * C) * (2.0 * r);   // (the zero blocks off the diagonal are not algorithmic work)
    } else {
      NULL_IF(gemm_rows(x->p, x->cols, x->rows, b.qkv.w, 3 * C, C, b.qkv.bias, y->p, 3 * C, nullptr, 0));
      if (lora)
        for (int s = 0; s < 3; ++s)
          NULL_IF(gemm_rows(t3->p + s * r, 3 * r, x->rows, l3[s]->B, C, r, nullptr, y->p + s * C, 3 * C, y->p + s * C, 3 * C));
    }
    if (R.save) {
      R.tape.push_back([x, y, t3, &b, C, r, lora, fold](Exec& E) -> int {
        if (!y->g) return 0;
        Lora* l3[3] = {&b.a1.q.lora, &b.a1.k.lora, &b.a1.v.lora};
        bf16_t* dx = E.grad_of(x);
        FDMI_CHECK(dx, "unet: workspace exhausted (grad)");
        if (!fold) {
          RET_IF(E.gemm_rows(y->g, 3 * C, x->rows, b.qkv.wt, C, 3 * C, nullptr, dx, x->cols, x->ginit ? dx : nullptr, x->cols));
          x->ginit = true;
        }
        if (lora) {   // per slice s: dt_s = dy_s B_s ; dB_s += dy_s^T t_s ; dA_s += dt_s^T x ; then dx += [dt_q | dt_k | dt_v] A3
          T* dt3 = E.R.mk(x->rows, 3 * r);
          FDMI_CHECK(dt3, "unet: workspace exhausted (lora)");
          if (r < 128 && (E.R.dry() || b.BT3) && x->rows >= 256 && !fdmi_tune_get(44)) {   // one N = 3r GEMM over blockdiag(B^T)
            RET_IF(E.gemm_rows(y->g, 3 * C, x->rows, b.BT3, 3 * r, 3 * C, nullptr, dt3->p, 3 * r, nullptr, 0));
            E.flops -= 2.0 * x->rows * (3.0 * r) * (2.0 * C);   // (the zero blocks are not algorithmic work)
          } else {
            for (int s = 0; s < 3; ++s)
              RET_IF(E.gemm_rows(y->g + s * C, 3 * C, x->rows, l3[s]->BT, r, C, nullptr, dt3->p + s * r, 3 * r, nullptr, 0));
          }
          E.flops += 3 * 2.0 * x->rows * r * (2.0 * C);
          if (!E.R.dry()) {
            WgradProblem wp[6];   // the six products of the three slices in one launch
            for (int s = 0; s < 3; ++s) {
              wp[2 * s] = WgradProblem{y->g + s * C, 3 * C, t3->p + s * r, 3 * r, x->rows, C, r, l3[s]->B_grad, r};
              wp[2 * s + 1] = WgradProblem{dt3->p + s * r, 3 * r, x->p, x->cols, x->rows, r, C, l3[s]->A_grad, C};
            }
            RET_IF(launch_wgrad_tn_group(wp, 6, E.st));
          }
          if (fold) {   // dx (+)= [dy_q dy_k dy_v | dt_q dt_k dt_v] [Wqkv^T | A_q^T A_k^T A_v^T]^T, one launch
            GemmArgs d = rows_args(y->g, 3 * C, x->rows, b.Wtc3, C, 3 * C + 3 * r, nullptr, dx, x->cols, x->ginit ? dx : nullptr, x->cols);
            d.ldw = 3 * C + 3 * r; d.A2 = dt3->p; d.lda2 = 3 * r; d.K1 = 3 * C;
            RET_IF(E.gemm(d));
            x->ginit = true;
          } else {
            RET_IF(E.gemm_rows(dt3->p, 3 * r, x->rows, b.AT3, C, 3 * r, nullptr, dx, x->cols, dx, x->cols));
          }
        }
        return 0;
      });
      R.mark_out({y, t3});
    }
    return y;
  }

  // dst->g[:, c0:c0+cols] (+)= src[:, sc0:sc0+cols]
  int add_grad(T* dst, const bf16_t* src, int64_t ld_src, int sc0, int cols, int dc0 = 0) {
    bf16_t* g = grad_of(dst);
    FDMI_CHECK(g, "unet: workspace exhausted (grad)");
    if (!R.dry()) {
      if (!dst->ginit && (dc0 != 0 || cols != dst->cols))
        FDMI_HIP(hipMemsetAsync(g, 0, (size_t)dst->rows * dst->cols * es(), st));
      RET_IF(l_copy2d(src, ld_src, sc0, g, dst->cols, dc0, dst->rows, cols, dst->ginit ? 1 : 0));
    }
    U->hbm[HBM_COPY2D] += (dst->ginit ? 6.0 : 4.0) * dst->rows * cols;
    dst->ginit = true;
    return 0;
  }

  // 3x3 (or kxk) convolution on NHWC; stride 1|2, optional fused nearest-2x upsample of the input
  // act: ACT_NONE | ACT_RELU (fused into the epilogue; the backward masks dY with the stored output first)
  T* conv(T* x, Weight& w, int stride, int ups, const bf16_t* rowvec, int64_t rowvec_ld, T* residual, bool gn_next = false,
          int act = ACT_NONE) {
    const int pad = w.KH / 2;
    const int Hv = x->H << ups, Wv = x->W << ups;
    const int Ho = (Hv + 2 * pad - w.KH) / stride + 1, Wo = (Wv + 2 * pad - w.KW) / stride + 1;
    T* y = R.mk((int64_t)x->B * Ho * Wo, w.N, x->B, Ho, Wo);
    if (!y) return nullptr;
    FDMI_CHECK_NULL(x->cols == w.Cin_pad, "conv: channel mismatch");
    GemmArgs a;
    a.mode = GEMM_CONV; a.M = (int)y->rows; a.N = w.N; a.K = w.K; a.A = x->p; a.W = w.w; a.ldw = w.K;
    a.Hin = x->H; a.Win = x->W; a.Cin = w.Cin_pad; a.Hout = Ho; a.Wout = Wo; a.KH = w.KH; a.KW = w.KW;
    a.stride = stride; a.pad = pad; a.ups = ups; a.bias = w.bias;
    a.rowvec = rowvec; a.rowvec_ld = rowvec_ld; a.rows_per_batch = Ho * Wo;
    a.residual = residual ? residual->p : nullptr; a.ldr = residual ? residual->cols : 0;
    a.C = y->p; a.ldc = w.N; a.act = act;
    if (gn_next) want_gn(a, Ho * Wo);
    NULL_IF(gemm(a));
    y->gn = a.gn_stats;
    if (R.save) {
      R.tape.push_back([x, y, residual, &w, stride, ups, pad, Hv, Wv, Ho, Wo, act](Exec& E) -> int {
        if (!y->g) return 0;
        if (act == ACT_RELU && !E.R.dry())   // every consumer of y has added its gradient by now (reverse tape order)
          RET_IF(E.f32() ? launch_relu_mask32(F(y->p), F(y->g), y->rows * y->cols, E.st)
                         : launch_relu_mask(y->p, y->g, y->rows * y->cols, E.st));
        if (residual) RET_IF(E.add_grad(residual, y->g, y->cols, 0, y->cols));
        // dgrad: gather form of the transposed conv over dY (channels padded to 8)
        const bf16_t* dy = y->g;
        int64_t dy_cols = y->cols;
        if (y->cols != w.Cout_pad) {  // e.g. conv_out (4 -> 8 channels): zero-padded copy
          bf16_t* pd = (bf16_t*)E.R.arena.alloc((size_t)y->rows * w.Cout_pad * E.es());
          FDMI_CHECK(pd, "unet: workspace exhausted");
          if (!E.R.dry())
            RET_IF(E.f32() ? launch_pad_cols32(F(y->g), y->cols, F(pd), w.Cout_pad, y->rows, E.st)
                           : launch_pad_cols(y->g, y->cols, pd, w.Cout_pad, y->rows, E.st));
          dy = pd;
          dy_cols = w.Cout_pad;
        }
        GemmArgs d;
        d.mode = GEMM_CONV; d.dgrad = 1; d.M = (int)((int64_t)x->B * Hv * Wv); d.N = w.Cin; d.K = w.KH * w.KW * w.Cout_pad;
        d.A = dy; d.W = w.wt; d.ldw = d.K;
        d.Hin = Ho; d.Win = Wo; d.Cin = (int)dy_cols; d.Hout = Hv; d.Wout = Wv; d.KH = w.KH; d.KW = w.KW;
        d.stride = stride; d.pad = pad;
        bf16_t* dx = E.grad_of(x);
        FDMI_CHECK(dx, "unet: workspace exhausted (grad)");
        if (ups) {
          bf16_t* tmp = (bf16_t*)E.R.arena.alloc((size_t)d.M * x->cols * E.es());
          FDMI_CHECK(tmp, "unet: workspace exhausted");
          d.C = tmp; d.ldc = x->cols;
          RET_IF(E.gemm(d));
          E.U->hbm[HBM_POOL] += 2.0 * x->rows * x->cols * (4 + 1 + (x->ginit ? 1 : 0));
          if (!E.R.dry())
            RET_IF(E.f32() ? launch_pool2x2_sum32(F(tmp), F(dx), x->B, x->H, x->W, x->cols, x->ginit ? 1 : 0, E.st)
                           : launch_pool2x2_sum(tmp, dx, x->B, x->H, x->W, x->cols, x->ginit ? 1 : 0, E.st));
        } else {
          d.C = dx; d.ldc = x->cols;
          if (x->ginit) { d.residual = dx; d.ldr = x->cols; }
          if (w.Cin != x->cols && !x->ginit && !E.R.dry())  // padded input channels (conv_in): zero the pad
            FDMI_HIP(hipMemsetAsync(dx, 0, (size_t)x->rows * x->cols * E.es(), E.st));
          RET_IF(E.gemm(d));
        }
        x->ginit = true;
        return 0;
      });
    }
    return y;
  }

  T* groupnorm(T* x, Norm& n, float eps, int silu) {
    T* y = R.mk(x->rows, x->cols, x->B, x->H, x->W);
    const size_t sbytes = (size_t)x->B * U->cfg.groups * 2 * sizeof(float);
    const bool ready = x->gn != nullptr;   // the producing GEMM's epilogue already left the sums (want_gn)
    ++U->last_gn;
    if (ready) ++U->last_gn_epi;
    const double el = (double)x->rows * x->cols;
    if (!ready) U->hbm[HBM_GN_REDUCE] += 2 * el;   // one read of the tensor
    U->hbm[HBM_GN_APPLY] += 4 * el;                // read + write
    float* stats = ready ? x->gn : R.zalloc(sbytes);
    const bool zeroed = stats != nullptr;
    if (!stats) stats = (float*)R.arena.alloc(sbytes);
    if (!y || !stats) return nullptr;
    const int HW = x->H * x->W, G = U->cfg.groups;
    if (!R.dry())
      NULL_IF(f32() ? launch_groupnorm32_fwd(F(x->p), n.gamma, n.beta, stats, F(y->p), x->B, HW, x->cols, G, eps, silu, st)
                    : launch_groupnorm_fwd(x->p, n.gamma, n.beta, stats, y->p, x->B, HW, x->cols, G, eps, silu, st, zeroed, ready,
                                           x->p2, x->c1));
    if (R.save) {
      R.tape.push_back([x, y, &n, stats, HW, G, eps, silu](Exec& E) -> int {
        if (!y->g) return 0;
        bf16_t* dx = E.grad_of(x);
        float* bst = E.R.zalloc((size_t)x->B * G * 2 * sizeof(float));
        const bool bzeroed = bst != nullptr;
        if (!bst) bst = (float*)E.R.arena.alloc((size_t)x->B * G * 2 * sizeof(float));
        FDMI_CHECK(dx && bst, "unet: workspace exhausted (grad)");
        E.U->hbm[HBM_GN_REDUCE] += 4.0 * x->rows * x->cols;                            // reads x and dy
        E.U->hbm[HBM_GN_APPLY] += (x->ginit ? 8.0 : 6.0) * x->rows * x->cols;        // reads x, dy (, dx), writes dx
        if (!E.R.dry())
          RET_IF(E.f32() ? launch_groupnorm32_bwd(F(x->p), F(y->g), n.gamma, n.beta, stats, F(dx), x->B, HW, x->cols, G, silu,
                                                  x->ginit ? 1 : 0, E.st)
                         : launch_groupnorm_bwd(x->p, y->g, n.gamma, n.beta, stats, bst, dx, x->B, HW, x->cols, G, eps, silu,
                                                x->ginit ? 1 : 0, E.st, bzeroed, x->p2, x->c1));
        x->ginit = true;
        return 0;
      });
    }
    return y;
  }

  T* layernorm(T* x, Norm& n) {
    T* y = R.mk(x->rows, x->cols, x->B, x->H, x->W);
    if (!y) return nullptr;
    U->hbm[HBM_LN] += 4.0 * x->rows * x->cols;
    if (!R.dry())
      NULL_IF(f32() ? launch_layernorm32_fwd(F(x->p), n.gamma, n.beta, nullptr, nullptr, 0, 1, F(y->p), x->rows, x->cols, 1e-5f, st)
                    : launch_layernorm_fwd(x->p, n.gamma, n.beta, nullptr, nullptr, 0, 1, y->p, x->rows, x->cols, 1e-5f, st));
    if (R.save) {
      R.tape.push_back([x, y, &n](Exec& E) -> int {
        if (!y->g) return 0;
        bf16_t* dx = E.grad_of(x);
        FDMI_CHECK(dx, "unet: workspace exhausted (grad)");
        E.U->hbm[HBM_LN] += (x->ginit ? 8.0 : 6.0) * x->rows * x->cols;
        if (!E.R.dry())
          RET_IF(E.f32() ? launch_layernorm32_bwd(F(x->p), F(y->g), n.gamma, nullptr, 0, 1, F(dx), x->rows, x->cols, 1e-5f,
                                                  x->ginit ? 1 : 0, E.st)
                         : launch_layernorm_bwd(x->p, y->g, n.gamma, nullptr, 0, 1, dx, x->rows, x->cols, 1e-5f, x->ginit ? 1 : 0,
                                                E.st));
        x->ginit = true;
        return 0;
      });
    }
    return y;
  }

  // the gradient of a tensor a consumer may have written through its parent (a row / column view): resolve it before testing
  bool has_grad(T* t) {
    if (t->parent && !t->g && t->parent->g) grad_of(t);
    return t->g != nullptr;
  }
  // vt_ext: caller-owned V^T buffer; vt_ready: it already holds the transposed V (cached context); o_ext: write the output into
  // this tensor (a row view of a larger one: per-sample launches of a masked cross-attention) instead of allocating it
  // skv_cap: size the key-side scratch (V^T, K^T) for this many keys instead of Skv -- the per-sample launches of a masked
  // cross-attention pass the unmasked length, so that a real run allocates exactly what the workspace query (which walks the plan
  // with every key valid) allocated and the backward's size-keyed free list sees the same sequence of sizes
  T* attention(T* q, T* k, T* v, int Bn, int H, int Sq, int Skv, bf16_t* vt_ext = nullptr, bool vt_ready = false, T* o_ext = nullptr,
               int skv_cap = 0) {
    const int d = q->cols / H;
    if (f32()) return attention32(q, k, v, Bn, H, Sq, Skv, d, o_ext);
    T* o = o_ext ? o_ext : R.mk(q->rows, q->cols, q->B, q->H, q->W);
    const int64_t tr_kv = (int64_t)Bn * H * attn_dvpad(d) * attn_spad(skv_cap > Skv ? skv_cap : Skv);
    const int64_t tr_q = (int64_t)Bn * H * attn_dvpad(d) * attn_spad(Sq);
    bf16_t* VT = vt_ext ? vt_ext : (bf16_t*)R.arena.alloc((size_t)tr_kv * 2);
    float* lse = R.save ? (float*)R.arena.alloc((size_t)Bn * H * Sq * 4) : nullptr;
    if (!o || !VT || (R.save && !lse)) return nullptr;
    AttnArgs a{};
    a.Q = q->p; a.ldq = q->ld(); a.K = k->p; a.ldk = k->ld(); a.V = v->p; a.ldv = v->ld(); a.VT = VT;
    a.lse = lse; a.out = o->p; a.ldout = o->cols; a.vt_ones = 1;
    a.B = Bn; a.H = H; a.Sq = Sq; a.Skv = Skv; a.d = d; a.scale = 1.f / sqrtf((float)d);
    flops += 4.0 * Bn * H * (double)Sq * Skv * d;
    if (!vt_ready) U->hbm[HBM_TRANSPOSE_HEADS] += 4.0 * Bn * Skv * H * d;
    if (!R.dry()) {
      if (!vt_ready) NULL_IF(launch_transpose_heads(v->p, v->ld(), VT, Bn, H, Skv, d, st, 1));
      NULL_IF(launch_attn_fwd(a, st));
    }
    if (R.save) {
      R.tape.push_back([o, q, k, v, a, Bn, H, Sq, Skv, d, tr_q, tr_kv, vt_ext](Exec& E) -> int {
        if (!E.has_grad(o)) return 0;
        bf16_t* QT = (bf16_t*)E.R.tmp((size_t)tr_q * 2);
        bf16_t* dOT = (bf16_t*)E.R.tmp((size_t)tr_q * 2);
        bf16_t* KT = (bf16_t*)E.R.tmp((size_t)tr_kv * 2);
        float* delta = (float*)E.R.tmp((size_t)Bn * H * Sq * 4);
        bf16_t *dq = E.grad_of(q), *dk = E.grad_of(k), *dv = E.grad_of(v);
        FDMI_CHECK(QT && dOT && KT && delta && dq && dk && dv, "unet: workspace exhausted (attn bwd)");
        E.flops += 2.0 * 4.0 * Bn * H * (double)Sq * Skv * d;  // algorithmic: 2x forward
        E.U->hbm[HBM_TRANSPOSE_HEADS] += 4.0 * Bn * H * d * (2.0 * Sq + Skv);
        if (!E.R.dry()) {
          RET_IF(launch_transpose_heads(q->p, q->ld(), QT, Bn, H, Sq, d, E.st));
          RET_IF(launch_transpose_heads(o->g, o->cols, dOT, Bn, H, Sq, d, E.st));
          RET_IF(launch_transpose_heads(k->p, k->ld(), KT, Bn, H, Skv, d, E.st));
          RET_IF(launch_attn_delta(o->p, o->cols, o->g, o->cols, delta, Bn, H, Sq, d, E.st));
          AttnArgs b = a;
          b.dO = o->g; b.lddo = o->cols; b.QT = QT; b.KT = KT; b.dOT = dOT; b.delta = delta;
          b.out = dq; b.ldout = q->ld(); b.dK = dk; b.lddk = k->ld(); b.dV = dv; b.lddv = v->ld();
          RET_IF(launch_attn_bwd_dq(b, E.st));
          RET_IF(launch_attn_bwd_dkv(b, E.st));
        }
        q->ginit = k->ginit = v->ginit = true;
        if (q->parent) q->parent->ginit = true;   // (the three slices of a fused projection are written together)
        if (!vt_ext) E.R.rfree((void*)a.VT, (size_t)tr_kv * 2);   // (the forward's V^T and log-sum-exp die with this entry)
        E.R.rfree(a.lse, (size_t)Bn * H * Sq * 4);
        return 0;
      });
      if (!o_ext) R.mark_out({o});
    }
    return o;
  }

  // fp32 validation plan: materialised scores, exact-f32 MFMA products, fp32 softmax (ref32.hip)
  T* attention32(T* q, T* k, T* v, int Bn, int H, int Sq, int Skv, int d, T* o_ext = nullptr) {
    T* o = o_ext ? o_ext : R.mk(q->rows, H * d, q->B, q->H, q->W);
    float* sc = attn_scratch(Bn, H, Sq, Skv, 0);
    if (!o || !sc) return nullptr;
    const float scale = 1.f / sqrtf((float)d);
    flops += 4.0 * Bn * H * (double)Sq * Skv * d;
    if (!R.dry())
      NULL_IF(launch_attn32_fwd(F(q->p), q->cols, F(k->p), k->cols, F(v->p), v->cols, F(o->p), o->cols, Bn, H, Sq, Skv, d, scale,
                                sc, R.sc32_elems, st));
    if (R.save) {
      R.tape.push_back([o, q, k, v, Bn, H, Sq, Skv, d, scale](Exec& E) -> int {
        if (!E.has_grad(o)) return 0;
        float* sc2 = E.attn_scratch(Bn, H, Sq, Skv, 1);
        bf16_t *dq = E.grad_of(q), *dk = E.grad_of(k), *dv = E.grad_of(v);
        FDMI_CHECK(sc2 && dq && dk && dv, "unet: workspace exhausted (attn bwd)");
        E.flops += 2.0 * 4.0 * Bn * H * (double)Sq * Skv * d;
        if (!E.R.dry())
          RET_IF(launch_attn32_bwd(F(q->p), q->cols, F(k->p), k->cols, F(v->p), v->cols, F(o->g), o->cols, F(dq), q->cols, F(dk),
                                   k->cols, F(dv), v->cols, Bn, H, Sq, Skv, d, scale, sc2, E.R.sc32_elems, E.st));
        q->ginit = k->ginit = v->ginit = true;
        return 0;
      });
      if (!o_ext) R.mark_out({o});
    }
    return o;
  }

  // GEGLU feed-forward input projection: [M,C] -> [M,4C]
  T* geglu(T* x, Weight& w) {
    const int F = w.N / 2;
    T* y = R.mk(x->rows, F, x->B, x->H, x->W);
    T* pre = R.save ? R.mk(x->rows, w.N) : nullptr;
    if (!y || (R.save && !pre)) return nullptr;
    NULL_IF(gemm_rows(x->p, x->cols, x->rows, w.w, w.N, w.K, w.bias, y->p, F, nullptr, 0, ACT_GEGLU,
                      pre ? pre->p : nullptr));
    if (R.save) {
      R.tape.push_back([x, y, &w, pre, F](Exec& E) -> int {
        if (!y->g) return 0;
        T* dpre = E.R.mk(x->rows, w.N);
        bf16_t* dx = E.grad_of(x);
        FDMI_CHECK(dpre && dx, "unet: workspace exhausted (geglu bwd)");
        E.U->hbm[HBM_GEGLU_BWD] += 2.0 * x->rows * (2.0 * F + F + 2.0 * F);   // reads pre [2F], dy [F]; writes dpre [2F]
        if (!E.R.dry())
          RET_IF(E.f32() ? launch_geglu32_bwd(Exec::F(pre->p), Exec::F(y->g), Exec::F(dpre->p), x->rows, F, E.st)
                         : launch_geglu_bwd(pre->p, y->g, dpre->p, x->rows, F, E.st));
        if (E.f32())
          RET_IF(E.gemm_rows_strided(dpre->p, w.N, x->rows, w.w, w.K, w.N, 1, w.K, dx, x->cols, x->ginit ? dx : nullptr, x->cols));
        else
          RET_IF(E.gemm_rows(dpre->p, w.N, x->rows, w.wt, w.K, w.N, nullptr, dx, x->cols, x->ginit ? dx : nullptr, x->cols));
        x->ginit = true;
        return 0;
      });
    }
    return y;
  }

  // virt: the result is only read by readers that take a two-part operand (groupnorm / linear_w: the ResNet block's norm1 and
  // its 1x1 shortcut): no copy, the tensor header points at both parts (A/B switch 30 = 1: always copy)
  T* cat(T* a, T* b, bool virt = false) {
    virt = virt && !f32() && !fdmi_tune_get(30) && (a->cols % 64) == 0 && (b->cols % 64) == 0 && a->rows >= 256 &&
           !a->parent && !b->parent && !a->p2 && !b->p2;
    T* y;
    if (virt) {
      R.tensors.emplace_back();
      y = &R.tensors.back();
      y->rows = a->rows; y->cols = a->cols + b->cols; y->B = a->B; y->H = a->H; y->W = a->W;
      y->p = a->p; y->p2 = b->p; y->c1 = a->cols;
    } else {
      y = R.mk(a->rows, a->cols + b->cols, a->B, a->H, a->W);
      if (!y) return nullptr;
      U->hbm[HBM_COPY2D] += 4.0 * y->rows * y->cols;
      if (!R.dry()) {
        NULL_IF(l_copy2d(a->p, a->cols, 0, y->p, y->cols, 0, a->rows, a->cols, 0));
        NULL_IF(l_copy2d(b->p, b->cols, 0, y->p, y->cols, a->cols, b->rows, b->cols, 0));
      }
    }
    if (R.save) {
      R.tape.push_back([y, a, b](Exec& E) -> int {
        if (!y->g) return 0;
        RET_IF(E.add_grad(a, y->g, y->cols, 0, a->cols));
        RET_IF(E.add_grad(b, y->g, y->cols, a->cols, b->cols));
        return 0;
      });
    }
    return y;
  }

  // [a ; a]: the point where the two halves of a classifier-free-guidance batch stop being identical (no-save runs only)
  T* dup(T* a) {
    T* y = R.mk(2 * a->rows, a->cols, 2 * a->B, a->H, a->W);
    if (!y) return nullptr;
    U->hbm[HBM_COPY2D] += 6.0 * a->rows * a->cols;
    if (!R.dry()) {
      NULL_IF(l_copy2d(a->p, a->cols, 0, y->p, y->cols, 0, a->rows, a->cols, 0));
      NULL_IF(l_copy2d(a->p, a->cols, 0, off(y->p, (int64_t)a->rows * a->cols), y->cols, 0, a->rows, a->cols, 0));
    }
    return y;
  }

  // ---- ops of the frozen nets beside the denoiser (VGG16 / LPIPS, VAE decoder, T2I adapter) ----
  T* maxpool2(T* x) {
    T* y = R.mk((int64_t)x->B * (x->H / 2) * (x->W / 2), x->cols, x->B, x->H / 2, x->W / 2);
    if (!y) return nullptr;
    if (!R.dry())
      NULL_IF(f32() ? launch_maxpool2_fwd32(F(x->p), F(y->p), x->B, x->H, x->W, x->cols, st)
                    : launch_maxpool2_fwd(x->p, y->p, x->B, x->H, x->W, x->cols, st));
    if (R.save) {
      R.tape.push_back([x, y](Exec& E) -> int {
        if (!y->g) return 0;
        bf16_t* dx = E.grad_of(x);
        FDMI_CHECK(dx, "net: workspace exhausted (grad)");
        if (!E.R.dry())
          RET_IF(E.f32() ? launch_maxpool2_bwd32(F(x->p), F(y->g), F(dx), x->B, x->H, x->W, x->cols, x->ginit ? 1 : 0, E.st)
                         : launch_maxpool2_bwd(x->p, y->g, dx, x->B, x->H, x->W, x->cols, x->ginit ? 1 : 0, E.st));
        x->ginit = true;
        return 0;
      });
    }
    return y;
  }
  T* avgpool2(T* x) {   // forward only (the adapter is frozen and its input carries no gradient)
    T* y = R.mk((int64_t)x->B * (x->H / 2) * (x->W / 2), x->cols, x->B, x->H / 2, x->W / 2);
    if (!y) return nullptr;
    if (!R.dry())
      NULL_IF(f32() ? launch_avgpool2_fwd32(F(x->p), F(y->p), x->B, x->H, x->W, x->cols, st)
                    : launch_avgpool2_fwd(x->p, y->p, x->B, x->H, x->W, x->cols, st));
    return y;
  }
  // y = a + b (element-wise; forward only: the adapter's ResNet skip)
  T* add(T* a, T* b) {
    T* y = R.mk(a->rows, a->cols, a->B, a->H, a->W);
    if (!y) return nullptr;
    if (!R.dry()) {
      NULL_IF(l_copy2d(a->p, a->cols, 0, y->p, y->cols, 0, a->rows, a->cols, 0));
      NULL_IF(l_copy2d(b->p, b->cols, 0, y->p, y->cols, 0, b->rows, b->cols, 1));
    }
    return y;
  }
  // single- or few-head attention whose head dim exceeds the flash kernels' (the VAE mid block: one head of 512): scores are
  // materialised by the exact-f32 kernels of the validation family (ref32.hip) in BOTH precisions -- a bf16 plan stages its
  // operands through fp32 copies.  One layer per decode; the FLOPs are 4 B S^2 d like any attention.
  T* attention_wide(T* q, T* k, T* v, int Bn, int H, int Sq, int Skv) {
    const int d = q->cols / H;
    if (f32()) return attention32(q, k, v, Bn, H, Sq, Skv, d);
    T* o = R.mk(q->rows, q->cols, q->B, q->H, q->W);
    const int64_t nq = q->rows * q->cols, nk = k->rows * k->cols;
    float* qf = (float*)R.arena.alloc((size_t)nq * 4);
    float* kf = (float*)R.arena.alloc((size_t)nk * 4);
    float* vf = (float*)R.arena.alloc((size_t)nk * 4);
    float* of = (float*)R.arena.alloc((size_t)nq * 4);
    float* sc = attn_scratch(Bn, H, Sq, Skv, 0);
    if (!o || !qf || !kf || !vf || !of || !sc) return nullptr;
    const float scale = 1.f / sqrtf((float)d);
    flops += 4.0 * Bn * H * (double)Sq * Skv * d;
    if (!R.dry()) {
      NULL_IF(launch_bf16_to_f32(q->p, qf, nq, st));
      NULL_IF(launch_bf16_to_f32(k->p, kf, nk, st));
      NULL_IF(launch_bf16_to_f32(v->p, vf, nk, st));
      NULL_IF(launch_attn32_fwd(qf, q->cols, kf, k->cols, vf, v->cols, of, q->cols, Bn, H, Sq, Skv, d, scale, sc, R.sc32_elems, st));
      NULL_IF(launch_f32_to_bf16(of, o->p, nq, st));
    }
    if (R.save) {
      R.tape.push_back([o, q, k, v, qf, kf, vf, of, Bn, H, Sq, Skv, d, scale, nq, nk](Exec& E) -> int {
        if (!o->g) return 0;
        float* sc2 = E.attn_scratch(Bn, H, Sq, Skv, 1);
        float* dqf = (float*)E.R.arena.alloc((size_t)nq * 4);
        float* dkf = (float*)E.R.arena.alloc((size_t)nk * 4);
        float* dvf = (float*)E.R.arena.alloc((size_t)nk * 4);
        bf16_t *dq = E.grad_of(q), *dk = E.grad_of(k), *dv = E.grad_of(v);
        FDMI_CHECK(sc2 && dqf && dkf && dvf && dq && dk && dv, "net: workspace exhausted (attn bwd)");
        FDMI_CHECK(!q->ginit && !k->ginit && !v->ginit, "attention_wide: q / k / v must have no other consumer");
        E.flops += 2.0 * 4.0 * Bn * H * (double)Sq * Skv * d;
        if (!E.R.dry()) {
          RET_IF(launch_bf16_to_f32(o->g, of, nq, E.st));   // (of is free: the forward's output was converted already)
          RET_IF(launch_attn32_bwd(qf, q->cols, kf, k->cols, vf, v->cols, of, q->cols, dqf, q->cols, dkf, k->cols, dvf, v->cols, Bn, H,
                                   Sq, Skv, d, scale, sc2, E.R.sc32_elems, E.st));
          RET_IF(launch_f32_to_bf16(dqf, dq, nq, E.st));
          RET_IF(launch_f32_to_bf16(dkf, dk, nk, E.st));
          RET_IF(launch_f32_to_bf16(dvf, dv, nk, E.st));
        }
        q->ginit = k->ginit = v->ginit = true;
        return 0;
      });
    }
    return o;
  }
  // one LPIPS level: out[b] += mean_pixels sum_c w[c] (unit(f0) - unit(f1))^2 ; gradient to f0 only (f1 = the no-grad side)
  int lpips_level(T* f0, T* f1, const float* w, float* out) {
    const int HW = f0->H * f0->W;
    if (!R.dry())
      RET_IF(f32() ? launch_lpips_level_fwd32(F(f0->p), F(f1->p), w, out, f0->rows, HW, f0->cols, st)
                   : launch_lpips_level_fwd(f0->p, f1->p, w, out, f0->rows, HW, f0->cols, st));
    if (R.save) {
      R.tape.push_back([f0, f1, w, HW](Exec& E) -> int {
        bf16_t* d0 = E.grad_of(f0);
        FDMI_CHECK(d0 && (E.R.dry() || E.R.gvec), "lpips: workspace exhausted / no output gradient");
        if (!E.R.dry())
          RET_IF(E.f32() ? launch_lpips_level_bwd32(F(f0->p), F(f1->p), w, E.R.gvec, F(d0), f0->rows, HW, f0->cols, f0->ginit ? 1 : 0, E.st)
                         : launch_lpips_level_bwd(f0->p, f1->p, w, E.R.gvec, d0, f0->rows, HW, f0->cols, f0->ginit ? 1 : 0, E.st));
        f0->ginit = true;
        return 0;
      });
    }
    return 0;
  }

  // ---- blocks ----------------------------------------------------------------------------------
  // gn_next: the block's output goes straight into a GroupNorm (the next ResNet block's or a transformer's)
  T* resnet(T* x, ResnetW& r, T* temb_all, bool gn_next = false) {
    const fdmi_unet_config& c = U->cfg;
    T* a = groupnorm(x, r.n1, c.eps, 1);
    if (!a) return nullptr;
    // (temb_all == nullptr: a ResnetBlock2D without time embedding -- the VAE's)
    T* h = temb_all ? conv(a, r.c1, 1, 0, off(temb_all->p, r.temb_off), temb_all->cols, nullptr, true)
                    : conv(a, r.c1, 1, 0, nullptr, 0, nullptr, true);
    if (!h) return nullptr;
    T* a2 = groupnorm(h, r.n2, c.eps, 1);
    if (!a2) return nullptr;
    T* sc = x;
    if (r.has_sc) {
      sc = linear_w(x, r.sc);
      if (!sc) return nullptr;
    }
    return conv(a2, r.c2, 1, 0, nullptr, 0, sc, gn_next);
  }

  // dup_after_attn1: x holds ONE half of a [x | x] guidance batch whose halves are identical up to here; the first
  // cross-attention (different context per half) is where they part: the state is duplicated right before it
  T* transformer(T* x, TransformerW& t, T* ctx, int L, bool dup_after_attn1 = false, bool gn_next = false) {
    int Bn = x->B;
    const int S = x->H * x->W;
    T* hn = groupnorm(x, t.gn, 1e-6f, 0);
    if (!hn) return nullptr;
    T* h = linear_w(hn, t.pin);
    if (!h) return nullptr;
    for (auto& bp : t.blocks) {
      TBlockW& b = *bp;
      T* n = layernorm(h, b.ln1);
      if (!n) return nullptr;
      T *q, *k, *v;
      if (qkv_fusable(U, b) && (R.dry() || b.qkv.w)) {
        T* qkv = linear_qkv(n, b, t.C);
        if (!qkv) return nullptr;
        q = R.view(qkv, 0, t.C); k = R.view(qkv, t.C, t.C); v = R.view(qkv, 2 * t.C, t.C);
      } else {
        q = linear(n, b.a1.q); k = linear(n, b.a1.k); v = linear(n, b.a1.v);
      }
      if (!q || !k || !v) return nullptr;
      T* o = attention(q, k, v, Bn, t.heads, S, S);
      if (!o) return nullptr;
      h = linear(o, b.a1.o, h);
      if (!h) return nullptr;
      if (dup_after_attn1) {
        dup_after_attn1 = false;
        h = dup(h);
        x = dup(x);   // the residual of proj_out
        if (!h || !x) return nullptr;
        Bn *= 2;
      }
      n = layernorm(h, b.ln2);
      if (!n) return nullptr;
      q = linear(n, b.a2.q);
      const bool cacheable = ctx_mode != 0 && !R.save && !b.a2.k.lora.on && !b.a2.v.lora.on;
      if (cacheable) {
        const int Cc = b.a2.k.w.N, dd = Cc / t.heads;
        const int64_t rows = (int64_t)Bn * L, vt = (int64_t)Bn * t.heads * attn_dvpad(dd) * attn_spad(L);
        if (!R.dry() && (b.c_rows != rows || b.c_vt != vt || !b.ck)) {  // (re)size the plan-owned buffers
          NULL_IF(dmalloc(U, &b.ck, (size_t)rows * Cc * (f32() ? 2 : 1)));
          NULL_IF(dmalloc(U, &b.cv, (size_t)rows * Cc * (f32() ? 2 : 1)));
          NULL_IF(dmalloc(U, &b.cvt, (size_t)vt));
          b.c_rows = rows; b.c_vt = vt; b.c_valid = false;
        }
        const bool reuse = ctx_mode == 2 && b.c_valid;
        k = R.wrap(b.ck, rows, Cc);
        v = R.wrap(b.cv, rows, Cc);
        if (!q || !k || !v) return nullptr;
        if (!reuse) {
          NULL_IF(gemm_rows(ctx->p, ctx->cols, rows, b.a2.k.w.w, Cc, b.a2.k.w.K, b.a2.k.w.bias, k->p, Cc, nullptr, 0));
          NULL_IF(gemm_rows(ctx->p, ctx->cols, rows, b.a2.v.w.w, Cc, b.a2.v.w.K, b.a2.v.w.bias, v->p, Cc, nullptr, 0));
        }
        o = attention(q, k, v, Bn, t.heads, S, L, R.dry() ? nullptr : b.cvt, reuse);
        if (!R.dry()) b.c_valid = true;
      } else {
        k = linear_w(ctx, b.a2.k.w, nullptr, b.a2.k.lora.on ? &b.a2.k.lora : nullptr, false);
        v = linear_w(ctx, b.a2.v.w, nullptr, b.a2.v.lora.on ? &b.a2.v.lora : nullptr, false);
        if (!q || !k || !v) return nullptr;
        o = attention(q, k, v, Bn, t.heads, S, L);
      }
      if (!o) return nullptr;
      h = linear(o, b.a2.o, h);
      if (!h) return nullptr;
      n = layernorm(h, b.ln3);
      if (!n) return nullptr;
      T* f = geglu(n, b.ff1);
      if (!f) return nullptr;
      h = linear_w(f, b.ff2, h);
      if (!h) return nullptr;
    }
    return linear_w(h, t.pout, x, nullptr, true, gn_next ? S : 0);
  }
};

#define FAIL_IF_NULL(p) FDMI_CHECK((p) != nullptr, "unet: op failed or workspace exhausted")

int run_forward(fdmi_unet* U, Run& R, const float* x, const float* t, const float* ctx, const float* cls,
                float* out, int B, int H, int W, int L, int flags) {
  const fdmi_unet_config& c = U->cfg;
  Exec E{U, R, R.st};
  E.ctx_mode = (flags & FDMI_UNET_CTX_REUSE) ? 2 : ((flags & FDMI_UNET_CTX_FILL) ? 1 : 0);
  // GroupNorm statistics from the producing GEMM's epilogue (A/B switch: FDMI_TUNE=14=1 turns it off).  Adapter residuals are
  // added in place AFTER their tensor was produced, so a forward that carries them keeps the reduce kernel everywhere.
  E.gn_epi = fdmi_tune_get(14) == 0 && !fdmi_det() && U->down_res.empty() && !U->f32;
  U->last_gn = U->last_gn_epi = 0;
  for (double& b : U->hbm) b = 0;
  R.tensors.clear();
  R.tape.clear();
  R.tape_outs.clear();
  R.recycle = false;
  R.arena.off = 0;
  R.es = U->f32 ? 4 : 2;
  R.sc32 = nullptr;
  R.sc32_elems = 0;
  if (U->f32) {   // one scratch region for the materialised attention scores, sized for the largest attention of the plan
    int64_t need = 0;
    for (int i = 0; i < U->nl; ++i) {
      if (!c.down_attn[i] && !c.up_attn[U->nl - 1 - i] && i != U->nl - 1) continue;
      const int S = (H >> i) * (W >> i), bw = (flags & FDMI_UNET_SAVE) ? 1 : 0;
      const int64_t a = attn32_scratch_elems(B, c.heads[i], S, S, bw), b = attn32_scratch_elems(B, c.heads[i], S, L, bw);
      need = a > need ? a : need;
      need = b > need ? b : need;
    }
    R.sc32 = (float*)R.arena.alloc((size_t)need * 4);
    FAIL_IF_NULL(R.sc32);
    R.sc32_elems = need;
  }
  R.save = (flags & FDMI_UNET_SAVE) != 0;
  const bool inter = (flags & FDMI_UNET_INTERMEDIATE) != 0;
  hipStream_t st = R.st;
  if (!R.dry() && U->fused_dirty) RET_IF(build_fused_operands(U, st));
  if (!U->loras.empty()) RET_IF(E.lora_refresh());
  {  // accumulator pool: every GroupNorm of the plan, forward + backward, 2 * B * groups floats each
    const size_t per = (((size_t)B * c.groups * 2 * sizeof(float)) + 255) & ~(size_t)255;
    R.zcap = per * 2 * 96;
    R.zoff = 0;
    R.zpool = (char*)R.arena.alloc(R.zcap);
    FAIL_IF_NULL(R.zpool);
    if (!R.dry()) FDMI_HIP(hipMemsetAsync(R.zpool, 0, R.zcap, st));
  }
  const int cin_pad = U->conv_in.Cin_pad;
  // FDMI_UNET_CFG_HALVES: sample and timestep rows B/2.. repeat rows 0..B/2-1 (a [x | x] guidance batch; only the context
  // differs): conv_in, the first ResNet block and the first self-attention are computed once and duplicated
  const bool halves = (flags & FDMI_UNET_CFG_HALVES) && !R.save && !inter && (B % 2) == 0 && !U->down.empty() &&
                      U->down[0]->has_attn && U->down_res.empty();
  const int Bx = halves ? B / 2 : B;
  // ---- inputs ----
  T* x0 = R.mk((int64_t)Bx * H * W, cin_pad, Bx, H, W);
  T* ctxb = R.mk((int64_t)B * L, c.cross_dim);   // (a copy: the tape may outlive the caller's tensor)
  FAIL_IF_NULL(x0); FAIL_IF_NULL(ctxb);
  R.x0 = x0;
  if (!R.dry()) {
    if (U->f32) {
      RET_IF(launch_nchw_to_nhwc32(x, Exec::F(x0->p), Bx, c.in_channels, H * W, cin_pad, st));
      FDMI_HIP(hipMemcpyAsync(ctxb->p, ctx, (size_t)B * L * c.cross_dim * 4, hipMemcpyDeviceToDevice, st));
    } else {
      RET_IF(launch_nchw_to_nhwc(x, x0->p, Bx, c.in_channels, H * W, cin_pad, st));
      RET_IF(launch_f32_to_bf16(ctx, ctxb->p, (int64_t)B * L * c.cross_dim, st));
    }
  }
  // ---- time embedding ----
  T* te = R.mk(B, c.block_out[0]);
  T* e1 = R.mk(B, U->temb_ch);
  T* emb = R.mk(B, U->temb_ch);
  FAIL_IF_NULL(te); FAIL_IF_NULL(e1); FAIL_IF_NULL(emb);
  if (!R.dry())
    RET_IF(U->f32 ? launch_timestep_embed32(t, Exec::F(te->p), B, c.block_out[0], c.flip_sin_to_cos, c.freq_shift, st)
                  : launch_timestep_embed(t, te->p, B, c.block_out[0], c.flip_sin_to_cos, c.freq_shift, st));
  RET_IF(E.gemm_rows(te->p, te->cols, B, U->te1.w, U->te1.N, U->te1.K, U->te1.bias, e1->p, e1->cols, nullptr, 0, ACT_SILU));
  RET_IF(E.gemm_rows(e1->p, e1->cols, B, U->te2.w, U->te2.N, U->te2.K, U->te2.bias, emb->p, emb->cols, nullptr, 0));
  if (c.class_embed_dim > 0) {
    FDMI_CHECK(cls != nullptr, "unet: class_labels (vector conditioning) required");
    T* cb = R.mk(B, c.class_embed_dim);
    T* c1 = R.mk(B, U->temb_ch);
    FAIL_IF_NULL(cb); FAIL_IF_NULL(c1);
    if (!R.dry()) {
      if (U->f32) FDMI_HIP(hipMemcpyAsync(cb->p, cls, (size_t)B * c.class_embed_dim * 4, hipMemcpyDeviceToDevice, st));
      else RET_IF(launch_f32_to_bf16(cls, cb->p, (int64_t)B * c.class_embed_dim, st));
    }
    RET_IF(E.gemm_rows(cb->p, cb->cols, B, U->ce1.w, U->ce1.N, U->ce1.K, U->ce1.bias, c1->p, c1->cols, nullptr, 0, ACT_SILU));
    RET_IF(E.gemm_rows(c1->p, c1->cols, B, U->ce2.w, U->ce2.N, U->ce2.K, U->ce2.bias, emb->p, emb->cols, emb->p, emb->cols));
  }
  T* semb = R.mk(B, U->temb_ch);
  T* temb_all = R.mk(B, U->temb_total);
  FAIL_IF_NULL(semb); FAIL_IF_NULL(temb_all);
  if (!R.dry())
    RET_IF(U->f32 ? launch_silu32(Exec::F(emb->p), Exec::F(semb->p), (int64_t)B * U->temb_ch, st)
                  : launch_silu(emb->p, semb->p, (int64_t)B * U->temb_ch, st));
  RET_IF(E.gemm_rows(semb->p, semb->cols, B, U->temb_proj.w, U->temb_proj.N, U->temb_proj.K, U->temb_proj.bias,
                     temb_all->p, temb_all->cols, nullptr, 0));
  // ---- down ----
  T* h = E.conv(x0, U->conv_in, 1, 0, nullptr, 0, nullptr, true);   // -> the first ResNet block's norm1
  FAIL_IF_NULL(h);
  std::vector<T*> skips{halves ? E.dup(h) : h};
  FAIL_IF_NULL(skips[0]);
  bool dup_pending = halves;
  // T2I-adapter residuals (diffusers UNet2DConditionModel.forward, `down_intrablock_additional_residuals`): a block WITH
  // cross-attention adds its residual to the hidden state after its last (resnet, attention) pair -- before that state is
  // pushed as a skip and before the downsampler; a block WITHOUT attention adds it to the block's output, in place, i.e.
  // also to the tensor already pushed as the block's last skip.  The residuals are constants of this forward (the adapter is
  // frozen, examples/train_flash_canny_adapter.py:362), so they are added in place and need no tape entry.
  std::vector<const float*> dres;
  if (!R.dry()) dres.swap(U->down_res);  // (a workspace query leaves them for the real forward)
  FDMI_CHECK(dres.empty() || dres.size() == U->down.size(), "unet: one adapter residual per down block expected");
  size_t di = 0;
  auto add_res = [&](T* t) -> int {
    if (dres.empty()) return 0;
    const float* r = dres[di++];
    if (r && !R.dry())
      RET_IF(U->f32 ? launch_add_nchw_to_nhwc32(r, U->down_res_scale, Exec::F(t->p), t->B, t->cols, t->H * t->W, st)
                    : launch_add_nchw_to_nhwc(r, U->down_res_scale, t->p, t->B, t->cols, t->H * t->W, st));
    return 0;
  };
  for (auto& sp : U->down) {
    StageW& s = *sp;
    for (size_t j = 0; j < s.res.size(); ++j) {
      // what reads the block's output next: a transformer's GroupNorm / the next ResNet block's norm1 / (last block of a
      // stage without downsampler) the mid block's norm1 -- or a downsampling conv, which has no norm in front of it
      const bool more = j + 1 < s.res.size() || !s.has_resample;
      h = E.resnet(h, *s.res[j], temb_all, s.has_attn || more);
      FAIL_IF_NULL(h);
      if (s.has_attn) {
        h = E.transformer(h, *s.attn[j], ctxb, L, dup_pending, more);
        dup_pending = false;
        FAIL_IF_NULL(h);
        if (j + 1 == s.res.size()) RET_IF(add_res(h));
      }
      skips.push_back(h);
    }
    if (s.has_resample) {
      h = E.conv(h, s.resample, 2, 0, nullptr, 0, nullptr, true);   // -> the next stage's first norm1
      FAIL_IF_NULL(h);
      skips.push_back(h);
    }
    if (!s.has_attn) RET_IF(add_res(h));
  }
  // ---- mid ----
  h = E.resnet(h, *U->mid_r0, temb_all, true);
  FAIL_IF_NULL(h);
  h = E.transformer(h, *U->mid_attn, ctxb, L, false, true);
  FAIL_IF_NULL(h);
  h = E.resnet(h, *U->mid_r1, temb_all);
  FAIL_IF_NULL(h);
  if (!inter) {
    for (auto& sp : U->up) {
      StageW& s = *sp;
      for (size_t j = 0; j < s.res.size(); ++j) {
        T* sk = skips.back();
        skips.pop_back();
        // virtual only when both readers take a two-part operand: the 1x1 shortcut GEMM exists and is wide enough for the LDS-DMA
        // kernels (N >= 128; a block without a shortcut conv would read the concat as a residual)
        T* hc = E.cat(h, sk, s.res[j]->has_sc && s.res[j]->cout >= 128);
        FAIL_IF_NULL(hc);
        // the up path's block outputs are concatenated with a skip before the next norm (no statistics to pass on), except
        // the very last one, which conv_norm_out reads
        const bool last = &sp == &U->up.back() && j + 1 == s.res.size() && !s.has_resample;
        h = E.resnet(hc, *s.res[j], temb_all, s.has_attn || last);
        FAIL_IF_NULL(h);
        if (s.has_attn) {
          h = E.transformer(h, *s.attn[j], ctxb, L, false, last);
          FAIL_IF_NULL(h);
        }
      }
      if (s.has_resample) {
        h = E.conv(h, s.resample, 1, 1, nullptr, 0, nullptr);
        FAIL_IF_NULL(h);
      }
    }
    T* a = E.groupnorm(h, U->norm_out, c.eps, 1);
    FAIL_IF_NULL(a);
    h = E.conv(a, U->conv_out, 1, 0, nullptr, 0, nullptr);
    FAIL_IF_NULL(h);
  }
  R.out = h;
  R.outC = h->cols;
  if (!R.dry())
    RET_IF(U->f32 ? launch_nhwc_to_nchw32(Exec::F(h->p), h->cols, out, B, h->cols, h->H * h->W, 0, st)
                  : launch_nhwc_to_nchw(h->p, h->cols, out, B, h->cols, h->H * h->W, 0, st));
  U->last_flops = E.flops;
  return 0;
}

int run_backward(fdmi_unet* U, Run& R, const float* grad_out, float* grad_x) {
  FDMI_CHECK(R.save && R.out, "unet: backward without a saved forward in this slot");
  Exec E{U, R, R.st};
  T* o = R.out;
  bf16_t* g = E.grad_of(o);
  FDMI_CHECK(g, "unet: workspace exhausted (grad)");
  if (!R.dry())   // (fp32: NCHW -> NHWC rows of the same width)
    RET_IF(U->f32 ? launch_nchw_to_nhwc32(grad_out, Exec::F(g), o->B, o->cols, o->H * o->W, o->cols, R.st)
                  : launch_nchw_grad_to_nhwc(grad_out, g, o->cols, o->B, o->cols, o->H * o->W, R.st));
  o->ginit = true;
  for (auto it = R.tape.rbegin(); it != R.tape.rend(); ++it) RET_IF((*it)(E));
  if (grad_x) {
    FDMI_CHECK(R.x0->g && R.x0->ginit, "unet: no gradient reached the input");
    if (!R.dry())
      RET_IF(U->f32 ? launch_nhwc_to_nchw32(Exec::F(R.x0->g), R.x0->cols, grad_x, R.x0->B, U->cfg.in_channels,
                                            R.x0->H * R.x0->W, 0, R.st)
                    : launch_nhwc_to_nchw(R.x0->g, R.x0->cols, grad_x, R.x0->B, U->cfg.in_channels, R.x0->H * R.x0->W, 0, R.st));
  }
  R.save = false;  // tape consumed
  R.tape.clear();
  U->last_flops = E.flops;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// plan construction + forward graphs of the frozen nets beside the denoiser
// ---------------------------------------------------------------------------------------------
int build_net(fdmi_unet* U) {
  const fdmi_net_config& c = U->ncfg;
  Builder b{U};
  if (U->kind == NET_VAE_DECODER) {
    FDMI_CHECK(c.n_levels >= 1 && c.n_levels <= 4, "vae: n_levels must be 1..4");
    auto v = std::make_unique<NetVae>();
    const int top = c.block_out[c.n_levels - 1];
    b.conv("post_quant_conv", v->post_quant, c.in_channels, c.in_channels, 1, true, /*pad_out=*/true);
    b.conv("decoder.conv_in", v->conv_in, top, c.in_channels, 3);
    v->mid_r0 = b.resnet_plain("decoder.mid_block.resnets.0", top, top);
    b.norm("decoder.mid_block.attentions.0.group_norm", v->mid_gn, top);
    b.linear("decoder.mid_block.attentions.0.to_q", v->q, top, top, true);
    b.linear("decoder.mid_block.attentions.0.to_k", v->k, top, top, true);
    b.linear("decoder.mid_block.attentions.0.to_v", v->v, top, top, true);
    b.linear("decoder.mid_block.attentions.0.to_out.0", v->o, top, top, true);
    v->mid_r1 = b.resnet_plain("decoder.mid_block.resnets.1", top, top);
    int prev = top;
    for (int i = 0; i < c.n_levels; ++i) {   // decoder levels run over reversed(block_out_channels)
      auto st = std::make_unique<StageW>();
      const int out = c.block_out[c.n_levels - 1 - i];
      const std::string bn = "decoder.up_blocks." + std::to_string(i);
      for (int j = 0; j < c.layers_per_block + 1; ++j)
        st->res.push_back(b.resnet_plain(bn + ".resnets." + std::to_string(j), j == 0 ? prev : out, out));
      st->has_resample = i != c.n_levels - 1;
      if (st->has_resample) b.conv(bn + ".upsamplers.0.conv", st->resample, out, out, 3);
      prev = out;
      v->up.push_back(std::move(st));
    }
    b.norm("decoder.conv_norm_out", v->norm_out, prev);
    b.conv("decoder.conv_out", v->conv_out, c.out_channels, prev, 3);
    U->vae = std::move(v);
  } else if (U->kind == NET_VGG_LPIPS) {
    auto g = std::make_unique<NetVgg>();
    // torchvision vgg16().features indices of the 13 convolutions, grouped into lpips' five slices (pretrained_networks.py)
    static const int idx[13] = {0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28};
    static const int slice[13] = {1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5};
    static const int ch[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
    int cin = 3;
    for (int i = 0; i < 13; ++i) {
      b.conv("net.slice" + std::to_string(slice[i]) + "." + std::to_string(idx[i]), g->conv[i], ch[i], cin, 3);
      cin = ch[i];
    }
    static const int lc[5] = {64, 128, 256, 512, 512};
    for (int l = 0; l < 5; ++l) b.vec("lin" + std::to_string(l) + ".model.1.weight", &g->lin[l], lc[l]);
    U->vgg = std::move(g);
  } else if (U->kind == NET_T2I_ADAPTER) {
    FDMI_CHECK(c.n_levels >= 1 && c.n_levels <= 4 && c.adapter_downscale > 0, "adapter: n_levels must be 1..4, downscale > 0");
    auto a = std::make_unique<NetAdapter>();
    const int f = c.adapter_downscale;
    b.conv("adapter.conv_in", a->conv_in, c.block_out[0], c.in_channels * f * f, 3);
    for (int i = 0; i < c.n_levels; ++i) {
      auto blk = std::make_unique<AdapterBlockW>();
      const int cin = i == 0 ? c.block_out[0] : c.block_out[i - 1], cout = c.block_out[i];
      // FullAdapter: every block but the first downsamples; FullAdapterXL: only block 2 (diffusers adapter.py)
      blk->down = c.adapter_xl ? (i == 2) : (i > 0);
      blk->has_in = cin != cout;
      const std::string bn = "adapter.body." + std::to_string(i);
      if (blk->has_in) b.conv(bn + ".in_conv", blk->in_conv, cout, cin, 1);
      for (int j = 0; j < c.layers_per_block; ++j) {
        auto pr = std::make_unique<std::pair<Weight, Weight>>();
        b.conv(bn + ".resnets." + std::to_string(j) + ".block1", pr->first, cout, cout, 3);
        b.conv(bn + ".resnets." + std::to_string(j) + ".block2", pr->second, cout, cout, 1);
        blk->res.push_back(std::move(pr));
      }
      a->body.push_back(std::move(blk));
    }
    U->adp = std::move(a);
  } else {
    FDMI_CHECK(false, "net: unknown kind");
  }
  return 0;
}

static void run_reset(fdmi_unet* U, Run& R, int flags) {
  U->last_gn = U->last_gn_epi = 0;
  for (double& b : U->hbm) b = 0;
  R.tensors.clear();
  R.tape.clear();
  R.tape_outs.clear();
  R.freelist.clear();
  R.scoped.clear();
  R.recycle = false;
  R.outs.clear();
  R.arena.off = 0;
  R.arena.peak = 0;   // (per run: the two-region block allocation of dit_plan.h rewinds to it)
  R.es = U->f32 ? 4 : 2;
  R.sc32 = nullptr;
  R.sc32_elems = 0;
  R.gvec = nullptr;
  R.save = (flags & FDMI_UNET_SAVE) != 0;
}
static int run_zpool(fdmi_unet* U, Run& R, int B, int n_norms) {   // accumulator pool of the GroupNorm statistics (fwd + bwd)
  const size_t per = (((size_t)B * U->cfg.groups * 2 * sizeof(float)) + 255) & ~(size_t)255;
  R.zcap = per * 2 * (size_t)(n_norms + 2);
  R.zoff = 0;
  R.zpool = (char*)R.arena.alloc(R.zcap);
  FAIL_IF_NULL(R.zpool);
  if (!R.dry()) FDMI_HIP(hipMemsetAsync(R.zpool, 0, R.zcap, R.st));
  return 0;
}

// AutoencoderKL.decode(z).sample (diffusers; the wrapper divides by the scaling factor first, vae/autoencoderKL.py:63-128):
// post_quant_conv 1x1 -> conv_in -> mid (ResNet, one-head attention over all H*W positions, ResNet) -> per level (layers + 1)
// ResNets [+ nearest-2x upsample fused into the following 3x3 conv] -> GroupNorm + SiLU -> conv_out
int run_vae_decoder(fdmi_unet* U, Run& R, const float* z, float* out, int B, int H, int W, int flags) {
  NetVae& V = *U->vae;
  const fdmi_net_config& c = U->ncfg;
  Exec E{U, R, R.st};
  E.gn_epi = fdmi_tune_get(14) == 0 && !fdmi_det() && !U->f32;
  run_reset(U, R, flags);
  int n_norms = 2 * 2 + 1 + 1;
  for (auto& st : V.up) n_norms += 2 * (int)st->res.size();
  RET_IF(run_zpool(U, R, B, n_norms));
  const int cin_pad = V.post_quant.Cin_pad;
  T* x0 = R.mk((int64_t)B * H * W, cin_pad, B, H, W);
  FAIL_IF_NULL(x0);
  R.x0 = x0;
  if (!R.dry())
    RET_IF(U->f32 ? launch_nchw_to_nhwc32(z, Exec::F(x0->p), B, c.in_channels, H * W, cin_pad, R.st)
                  : launch_nchw_to_nhwc(z, x0->p, B, c.in_channels, H * W, cin_pad, R.st));
  T* h = E.conv(x0, V.post_quant, 1, 0, nullptr, 0, nullptr);
  FAIL_IF_NULL(h);
  h = E.conv(h, V.conv_in, 1, 0, nullptr, 0, nullptr, true);
  FAIL_IF_NULL(h);
  h = E.resnet(h, *V.mid_r0, nullptr, true);
  FAIL_IF_NULL(h);
  {   // Attention(heads = 1, residual_connection = True, norm_num_groups): x + to_out(softmax(q k^T / sqrt(C)) v) on GroupNorm(x)
    T* hn = E.groupnorm(h, V.mid_gn, c.eps, 0);
    FAIL_IF_NULL(hn);
    T *q = E.linear_w(hn, V.q), *k = E.linear_w(hn, V.k), *v = E.linear_w(hn, V.v);
    FAIL_IF_NULL(q); FAIL_IF_NULL(k); FAIL_IF_NULL(v);
    T* o = E.attention_wide(q, k, v, B, 1, H * W, H * W);
    FAIL_IF_NULL(o);
    h = E.linear_w(o, V.o, h, nullptr, true, H * W);
    FAIL_IF_NULL(h);
  }
  h = E.resnet(h, *V.mid_r1, nullptr, true);
  FAIL_IF_NULL(h);
  for (auto& sp : V.up) {
    StageW& st = *sp;
    for (size_t j = 0; j < st.res.size(); ++j) {
      // the next reader is a GroupNorm (the next ResNet's norm1 / conv_norm_out) unless an upsampling conv comes first
      const bool gn_next = j + 1 < st.res.size() || !st.has_resample;
      h = E.resnet(h, *st.res[j], nullptr, gn_next);
      FAIL_IF_NULL(h);
    }
    if (st.has_resample) {
      h = E.conv(h, st.resample, 1, 1, nullptr, 0, nullptr, true);
      FAIL_IF_NULL(h);
    }
  }
  T* a = E.groupnorm(h, V.norm_out, c.eps, 1);
  FAIL_IF_NULL(a);
  h = E.conv(a, V.conv_out, 1, 0, nullptr, 0, nullptr);
  FAIL_IF_NULL(h);
  R.out = h;
  R.outC = h->cols;
  if (!R.dry())
    RET_IF(U->f32 ? launch_nhwc_to_nchw32(Exec::F(h->p), h->cols, out, B, h->cols, h->H * h->W, 0, R.st)
                  : launch_nhwc_to_nchw(h->p, h->cols, out, B, h->cols, h->H * h->W, 0, R.st));
  U->last_flops = E.flops;
  return 0;
}

// lpips.LPIPS(net="vgg", lpips=True, spatial=False).forward(in0, in1) (lpips 0.1.4): ScalingLayer, the VGG16 feature stack
// (conv3x3 + ReLU x 13, max pooling in front of slices 2..5), per tap: channel-unit-normalise both, squared difference, the
// non-negative 1x1 lin layer, spatial mean; the five levels are summed.  in1 is the no-grad side (the teacher's decode): its
// features are computed without a tape.
static int vgg_features(Exec& E, NetVgg& G, T* x, T* feats[5]) {
  T* h = x;
  int l = 0;
  for (int i = 0; i < 13; ++i) {
    if (i == 2 || i == 4 || i == 7 || i == 10) {
      h = E.maxpool2(h);
      FAIL_IF_NULL(h);
    }
    h = E.conv(h, G.conv[i], 1, 0, nullptr, 0, nullptr, false, ACT_RELU);
    FAIL_IF_NULL(h);
    if (i == 1 || i == 3 || i == 6 || i == 9 || i == 12) feats[l++] = h;
  }
  return 0;
}
int run_lpips(fdmi_unet* U, Run& R, const float* img0, const float* img1, float* out, int B, int H, int W, int flags) {
  NetVgg& G = *U->vgg;
  const fdmi_net_config& c = U->ncfg;
  FDMI_CHECK(H % 16 == 0 && W % 16 == 0, "lpips: H, W must be multiples of 16 (four 2x2 poolings)");
  Exec E{U, R, R.st};
  run_reset(U, R, flags);
  const bool save = R.save;
  const int cpad = G.conv[0].Cin_pad;
  T* x1 = R.mk((int64_t)B * H * W, cpad, B, H, W);
  T* x0 = R.mk((int64_t)B * H * W, cpad, B, H, W);
  FAIL_IF_NULL(x1); FAIL_IF_NULL(x0);
  R.x0 = x0;
  if (!R.dry()) {
    FDMI_HIP(hipMemsetAsync(out, 0, (size_t)B * sizeof(float), R.st));
    RET_IF(U->f32 ? launch_lpips_input32(img1, Exec::F(x1->p), B, H * W, cpad, c.lpips_shift, c.lpips_scale, R.st)
                  : launch_lpips_input(img1, x1->p, B, H * W, cpad, c.lpips_shift, c.lpips_scale, R.st));
    RET_IF(U->f32 ? launch_lpips_input32(img0, Exec::F(x0->p), B, H * W, cpad, c.lpips_shift, c.lpips_scale, R.st)
                  : launch_lpips_input(img0, x0->p, B, H * W, cpad, c.lpips_shift, c.lpips_scale, R.st));
  }
  T *f1[5], *f0[5];
  R.save = false;                       // the reference side: no tape
  RET_IF(vgg_features(E, G, x1, f1));
  R.save = save;
  RET_IF(vgg_features(E, G, x0, f0));
  for (int l = 0; l < 5; ++l) RET_IF(E.lpips_level(f0[l], f1[l], G.lin[l], out));
  R.out = nullptr;
  U->last_flops = E.flops;
  return 0;
}

// diffusers T2IAdapter (full_adapter / full_adapter_xl), frozen: PixelUnshuffle -> conv_in -> per block [AvgPool2d(2)]
// [1x1 in_conv] + resnets (x + conv1x1(relu(conv3x3(x)))); the output of every block is a feature map
int run_adapter(fdmi_unet* U, Run& R, const float* x, float* const* outs, int n_outs, int B, int H, int W) {
  NetAdapter& A = *U->adp;
  const fdmi_net_config& c = U->ncfg;
  const int f = c.adapter_downscale;
  FDMI_CHECK(n_outs == (int)A.body.size(), "adapter: one output per block expected");
  FDMI_CHECK(H % f == 0 && W % f == 0, "adapter: H, W must be multiples of the downscale factor");
  Exec E{U, R, R.st};
  run_reset(U, R, 0);
  const int cpad = A.conv_in.Cin_pad;
  T* h = R.mk((int64_t)B * (H / f) * (W / f), cpad, B, H / f, W / f);
  FAIL_IF_NULL(h);
  if (!R.dry())
    RET_IF(U->f32 ? launch_pixel_unshuffle32(x, Exec::F(h->p), B, c.in_channels, H, W, f, cpad, R.st)
                  : launch_pixel_unshuffle(x, h->p, B, c.in_channels, H, W, f, cpad, R.st));
  h = E.conv(h, A.conv_in, 1, 0, nullptr, 0, nullptr);
  FAIL_IF_NULL(h);
  for (size_t i = 0; i < A.body.size(); ++i) {
    AdapterBlockW& blk = *A.body[i];
    if (blk.down) {
      FDMI_CHECK(h->H % 2 == 0 && h->W % 2 == 0, "adapter: odd feature map in front of a downsampling block");
      h = E.avgpool2(h);
      FAIL_IF_NULL(h);
    }
    if (blk.has_in) {
      h = E.conv(h, blk.in_conv, 1, 0, nullptr, 0, nullptr);
      FAIL_IF_NULL(h);
    }
    for (auto& pr : blk.res) {
      T* t = E.conv(h, pr->first, 1, 0, nullptr, 0, nullptr, false, ACT_RELU);
      FAIL_IF_NULL(t);
      h = E.conv(t, pr->second, 1, 0, nullptr, 0, h);   // + x as the epilogue's residual
      FAIL_IF_NULL(h);
    }
    R.outs.push_back(h);
    if (!R.dry())
      RET_IF(U->f32 ? launch_nhwc_to_nchw_any32(Exec::F(h->p), outs[i], B, h->cols, h->H * h->W, R.st)
                    : launch_nhwc_to_nchw_any(h->p, outs[i], B, h->cols, h->H * h->W, R.st));
  }
  U->last_flops = E.flops;
  return 0;
}

int run_net_backward(fdmi_unet* U, Run& R, const float* grad_out, float* grad_x) {
  FDMI_CHECK(R.save, "net: backward without a saved forward in this slot");
  Exec E{U, R, R.st};
  if (U->kind == NET_VGG_LPIPS) {
    R.gvec = grad_out;
  } else {
    FDMI_CHECK(R.out, "net: backward without a saved forward in this slot");
    T* o = R.out;
    bf16_t* g = E.grad_of(o);
    FDMI_CHECK(g, "net: workspace exhausted (grad)");
    if (!R.dry())
      RET_IF(U->f32 ? launch_nchw_to_nhwc32(grad_out, Exec::F(g), o->B, o->cols, o->H * o->W, o->cols, R.st)
                    : launch_nchw_grad_to_nhwc(grad_out, g, o->cols, o->B, o->cols, o->H * o->W, R.st));
    o->ginit = true;
  }
  for (auto it = R.tape.rbegin(); it != R.tape.rend(); ++it) RET_IF((*it)(E));
  FDMI_CHECK(R.dry() || (R.x0->g && R.x0->ginit), "net: no gradient reached the input");
  if (grad_x && !R.dry()) {
    if (U->kind == NET_VGG_LPIPS)
      RET_IF(U->f32 ? launch_lpips_input_bwd32(Exec::F(R.x0->g), grad_x, R.x0->B, R.x0->H * R.x0->W, R.x0->cols, U->ncfg.lpips_scale, R.st)
                    : launch_lpips_input_bwd(R.x0->g, grad_x, R.x0->B, R.x0->H * R.x0->W, R.x0->cols, U->ncfg.lpips_scale, R.st));
    else
      RET_IF(U->f32 ? launch_nhwc_to_nchw32(Exec::F(R.x0->g), R.x0->cols, grad_x, R.x0->B, U->ncfg.in_channels, R.x0->H * R.x0->W, 0, R.st)
                    : launch_nhwc_to_nchw(R.x0->g, R.x0->cols, grad_x, R.x0->B, U->ncfg.in_channels, R.x0->H * R.x0->W, 0, R.st));
  }
  R.save = false;
  R.tape.clear();
  U->last_flops = E.flops;
  return 0;
}

#include "dit_plan.h"

int check_ready(fdmi_unet* U) {
  for (auto& kv : U->slots) {
    const Slot& s = kv.second;
    bool ok = true;
    if (s.kind == S_CONV_W || s.kind == S_LIN_W) ok = (s.w->set & 1) != 0;
    else if (s.kind == S_BIAS) ok = (s.w->set & 2) != 0;
    else if (s.kind == S_GAMMA) ok = (s.n->set & 1) != 0;
    else if (s.kind == S_BETA) ok = (s.n->set & 2) != 0;
    else if (s.kind == S_VEC) ok = *s.vec != nullptr;
    else ok = U->temb_proj.w != nullptr;
    FDMI_CHECK(ok, "unet: parameter '" + kv.first + "' was never set");
  }
  return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

fdmi_unet* fdmi_unet_create(const fdmi_unet_config* cfg) {
  if (!cfg) { fdmi_set_error("null config"); return nullptr; }
  auto* U = new fdmi_unet();
  U->cfg = *cfg;
  U->f32 = cfg->precision == 1;
  if (cfg->precision != 0 && cfg->precision != 1) {
    fdmi_set_error("unet: precision must be 0 (bf16 MFMA) or 1 (fp32 validation mode)");
    delete U;
    return nullptr;
  }
  if (build_plan(U)) { delete U; return nullptr; }
  return U;
}
void fdmi_unet_destroy(fdmi_unet* U) { delete U; }

int64_t fdmi_unet_num_params(fdmi_unet* U) { return U ? (int64_t)U->slots.size() : -1; }
int fdmi_unet_param_name(fdmi_unet* U, int64_t i, char* buf, int64_t buflen, int64_t* numel) {
  FDMI_CHECK(U && i >= 0 && i < (int64_t)U->slots.size(), "param index out of range");
  auto it = U->slots.begin();
  std::advance(it, i);
  FDMI_CHECK((int64_t)it->first.size() + 1 <= buflen, "param name buffer too small");
  memcpy(buf, it->first.c_str(), it->first.size() + 1);
  if (numel) *numel = it->second.numel;
  return 0;
}
int fdmi_unet_set_param(fdmi_unet* U, const char* name, const float* data, int64_t numel, void* stream) {
  FDMI_CHECK(U && name && data, "null argument");
  return set_param(U, name, data, numel, (hipStream_t)stream);
}
int fdmi_unet_set_lora(fdmi_unet* U, const char* target, const float* A, const float* B, float* A_grad,
                       float* B_grad, int rank) {
  FDMI_CHECK(U && target && A && B, "null argument");
  auto it = U->lora_targets.find(target);
  FDMI_CHECK(it != U->lora_targets.end(), std::string("unet: '") + target + "' is not a LoRA-capable linear");
  FDMI_CHECK(rank > 0 && rank % 8 == 0, "unet: LoRA rank must be a positive multiple of 8");
  Lora& l = it->second->lora;
  const Weight& w = it->second->w;
  if (l.A_master != A || l.B_master != B) U->cast_dirty = true;
  l.A_master = A; l.B_master = B; l.A_grad = A_grad; l.B_grad = B_grad;
  FDMI_CHECK(!l.on || l.r == rank, "unet: LoRA rank changed");
  if (!l.A) {
    const bool declared = l.on;
    l.r = rank; l.in = w.K; l.out = w.N;
    l.rp = (!U->f32 && rank < 128) ? 128 : rank;
    RET_IF(dmalloc(U, &l.A, (size_t)l.rp * l.in));    // (rows r .. rp stay zero: dmalloc clears, the refresh writes r rows)
    RET_IF(dmalloc(U, &l.AT, (size_t)rank * l.in));
    RET_IF(dmalloc(U, &l.B, (size_t)rank * l.out));
    RET_IF(dmalloc(U, &l.BT, (size_t)l.rp * l.out));
    l.on = true;
    if (!declared) U->loras.push_back(&l);
    U->fused_dirty = true;   // the folded [W | B] / [W^T | A^T] operands of this linear are built at the next forward
  }
  return 0;
}
int fdmi_unet_declare_lora(fdmi_unet* U, const char* target, int rank) {
  FDMI_CHECK(U && target, "null argument");
  auto it = U->lora_targets.find(target);
  FDMI_CHECK(it != U->lora_targets.end(), std::string("unet: '") + target + "' is not a LoRA-capable linear");
  FDMI_CHECK(rank > 0 && rank % 8 == 0, "unet: LoRA rank must be a positive multiple of 8");
  Lora& l = it->second->lora;
  FDMI_CHECK(!l.on || l.r == rank, "unet: LoRA rank changed");
  if (!l.on) {
    l.r = rank; l.in = it->second->w.K; l.out = it->second->w.N;
    l.rp = (!U->f32 && rank < 128) ? 128 : rank;
    l.on = true;
    U->loras.push_back(&l);
  }
  return 0;
}
int fdmi_unet_ready(fdmi_unet* U) {
  FDMI_CHECK(U, "null plan");
  return check_ready(U);
}

int64_t fdmi_unet_workspace_bytes(fdmi_unet* U, int B, int H, int W, int L, int flags) {
  if (!U) return -1;
  Run R;
  R.arena.dry = true;
  if (run_forward(U, R, nullptr, nullptr, nullptr, (const float*)(uintptr_t)256, nullptr, B, H, W, L, flags)) return -1;
  if (flags & FDMI_UNET_SAVE) {
    if (run_backward(U, R, nullptr, (flags & FDMI_UNET_INPUT_GRAD) ? (float*)(uintptr_t)256 : nullptr)) return -1;
  }
  return (int64_t)R.arena.peak + (1 << 20);
}

int fdmi_unet_forward(fdmi_unet* U, int slot, const float* sample, const float* timestep, const float* ctx,
                      const float* class_labels, float* out, int B, int H, int W, int L, void* workspace,
                      int64_t workspace_bytes, int flags, void* stream) {
  FDMI_CHECK(U && U->kind == NET_UNET && slot >= 0 && slot < 8, "bad plan / slot");
  FDMI_CHECK(sample && timestep && ctx && out && workspace, "null argument");
  FDMI_CHECK(H % (1 << (U->nl - 1)) == 0 && W % (1 << (U->nl - 1)) == 0, "unet: H, W must be divisible by 2^(levels-1)");
  Run& R = U->runs[slot];
  R.arena.base = (char*)workspace;
  R.arena.cap = (size_t)workspace_bytes;
  R.arena.dry = false;
  R.st = (hipStream_t)stream;
  return run_forward(U, R, sample, timestep, ctx, class_labels, out, B, H, W, L, flags);
}

int fdmi_unet_backward(fdmi_unet* U, int slot, const float* grad_out, float* grad_sample, void* stream) {
  FDMI_CHECK(U && slot >= 0 && slot < 8 && grad_out, "bad plan / slot / null grad");
  Run& R = U->runs[slot];
  R.st = (hipStream_t)stream;
  return run_backward(U, R, grad_out, grad_sample);
}

double fdmi_unet_last_flops(fdmi_unet* U) { return U ? U->last_flops : 0.0; }
double fdmi_unet_last_hbm_bytes(fdmi_unet* U, int family) {
  return (U && family >= 0 && family < FDMI_HBM_FAMILIES) ? U->hbm[family] : -1.0;
}
int fdmi_unet_last_gn_epilogue(fdmi_unet* U, int* total) {
  if (total) *total = U ? U->last_gn : 0;
  return U ? U->last_gn_epi : 0;
}

int fdmi_unet_set_down_residuals(fdmi_unet* U, const float* const* residuals, int n, float scale) {
  FDMI_CHECK(U != nullptr, "null plan");
  FDMI_CHECK(n == 0 || (residuals && n == (int)U->down.size()), "unet: one adapter residual per down block expected");
  U->down_res.assign(residuals, residuals + n);
  U->down_res_scale = scale;
  return 0;
}

// ---- the frozen nets beside the denoiser (VAE decoder, LPIPS-VGG, T2I adapter): same handle type, own graphs ----
fdmi_unet* fdmi_net_create(const fdmi_net_config* cfg) {
  if (!cfg) { fdmi_set_error("null config"); return nullptr; }
  if (cfg->kind != NET_VAE_DECODER && cfg->kind != NET_VGG_LPIPS && cfg->kind != NET_T2I_ADAPTER) {
    fdmi_set_error("net: kind must be FDMI_NET_VAE_DECODER, FDMI_NET_VGG_LPIPS or FDMI_NET_T2I_ADAPTER");
    return nullptr;
  }
  if (cfg->precision != 0 && cfg->precision != 1) {
    fdmi_set_error("net: precision must be 0 (bf16 MFMA) or 1 (fp32 validation mode)");
    return nullptr;
  }
  auto* U = new fdmi_unet();
  U->kind = cfg->kind;
  U->ncfg = *cfg;
  U->cfg = fdmi_unet_config{};
  U->cfg.groups = cfg->groups > 0 ? cfg->groups : 32;
  U->cfg.eps = cfg->eps;
  U->cfg.precision = cfg->precision;
  U->f32 = cfg->precision == 1;
  if (build_net(U)) { delete U; return nullptr; }
  return U;
}
static int net_run(fdmi_unet* U, Run& R, const float* x, const float* x2, float* out, int B, int H, int W, int flags) {
  if (U->kind == NET_VAE_DECODER) return run_vae_decoder(U, R, x, out, B, H, W, flags);
  if (U->kind == NET_VGG_LPIPS) return run_lpips(U, R, x, x2, out, B, H, W, flags);
  FDMI_CHECK(false, "net: this plan kind has its own entry point");
}
int64_t fdmi_net_workspace_bytes(fdmi_unet* U, int B, int H, int W, int flags) {
  if (!U || U->kind == NET_UNET) return -1;
  Run R;
  R.arena.dry = true;
  if (U->kind == NET_T2I_ADAPTER) {
    if (run_adapter(U, R, nullptr, nullptr, (int)U->adp->body.size(), B, H, W)) return -1;
    return (int64_t)R.arena.peak + (1 << 20);
  }
  if (net_run(U, R, nullptr, nullptr, nullptr, B, H, W, flags)) return -1;
  if (flags & FDMI_UNET_SAVE) {
    if (run_net_backward(U, R, nullptr, (float*)(uintptr_t)256)) return -1;
  }
  return (int64_t)R.arena.peak + (1 << 20);
}
int fdmi_net_forward(fdmi_unet* U, int slot, const float* x, const float* x2, float* out, int B, int H, int W, void* workspace,
                     int64_t workspace_bytes, int flags, void* stream) {
  FDMI_CHECK(U && U->kind != NET_UNET && slot >= 0 && slot < 8, "bad plan / slot");
  FDMI_CHECK(x && out && workspace && (U->kind != NET_VGG_LPIPS || x2), "null argument");
  Run& R = U->runs[slot];
  R.arena.base = (char*)workspace;
  R.arena.cap = (size_t)workspace_bytes;
  R.arena.dry = false;
  R.st = (hipStream_t)stream;
  return net_run(U, R, x, x2, out, B, H, W, flags);
}
int fdmi_net_backward(fdmi_unet* U, int slot, const float* grad_out, float* grad_x, void* stream) {
  FDMI_CHECK(U && U->kind != NET_UNET && slot >= 0 && slot < 8 && grad_out && grad_x, "bad plan / slot / null gradient");
  Run& R = U->runs[slot];
  R.st = (hipStream_t)stream;
  return run_net_backward(U, R, grad_out, grad_x);
}
int fdmi_adapter_out_shape(fdmi_unet* U, int level, int H, int W, int* C, int* h, int* w) {
  FDMI_CHECK(U && U->kind == NET_T2I_ADAPTER && level >= 0 && level < (int)U->adp->body.size() && C && h && w, "adapter: bad argument");
  const int f = U->ncfg.adapter_downscale;
  int hh = H / f, ww = W / f;
  for (int i = 0; i <= level; ++i)
    if (U->adp->body[i]->down) { hh /= 2; ww /= 2; }
  *C = U->ncfg.block_out[level]; *h = hh; *w = ww;
  return 0;
}
int fdmi_adapter_forward(fdmi_unet* U, int slot, const float* x, float* const* outs, int n_outs, int B, int H, int W, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  FDMI_CHECK(U && U->kind == NET_T2I_ADAPTER && slot >= 0 && slot < 8 && x && outs && workspace, "adapter: bad argument");
  Run& R = U->runs[slot];
  R.arena.base = (char*)workspace;
  R.arena.cap = (size_t)workspace_bytes;
  R.arena.dry = false;
  R.st = (hipStream_t)stream;
  return run_adapter(U, R, x, outs, n_outs, B, H, W);
}

// ---- the transformer denoisers (dit_plan.h): same handle type, fdmi_unet_set_param / _set_lora / _ready / _destroy apply ----
fdmi_unet* fdmi_dit_create(const fdmi_dit_config* cfg) {
  if (!cfg) { fdmi_set_error("null config"); return nullptr; }
  if (cfg->precision != 0 && cfg->precision != 1) {
    fdmi_set_error("dit: precision must be 0 (bf16 MFMA) or 1 (fp32 validation mode)");
    return nullptr;
  }
  auto* U = new fdmi_unet();
  U->kind = cfg->kind == FDMI_DIT_MMDIT ? NET_DIT_MMDIT : NET_DIT_PIXART;
  U->dcfg = *cfg;
  U->cfg = fdmi_unet_config{};
  U->cfg.precision = cfg->precision;
  U->f32 = cfg->precision == 1;
  if (build_dit(U)) { delete U; return nullptr; }
  return U;
}
static bool is_dit(const fdmi_unet* U) { return U && (U->kind == NET_DIT_PIXART || U->kind == NET_DIT_MMDIT); }
int64_t fdmi_dit_workspace_bytes(fdmi_unet* U, int B, int H, int W, int L, int masked, int flags) {
  if (!is_dit(U)) return -1;
  Run R;
  R.arena.dry = true;
  DitIn in{};
  in.B = B; in.H = H; in.W = W; in.L = L; in.keep = 1;
  in.vec = (const float*)(uintptr_t)256;
  if (masked) in.lens.assign(B, L);   // (per-sample launches: a few more small buffers than the one-launch form)
  if (run_dit(U, R, in, flags)) return -1;
  if (flags & FDMI_UNET_SAVE) {
    if (run_dit_backward(U, R, nullptr, (flags & FDMI_UNET_INPUT_GRAD) ? (float*)(uintptr_t)256 : nullptr)) return -1;
  }
  return (int64_t)R.arena.peak + (1 << 20);
}
int fdmi_dit_forward(fdmi_unet* U, int slot, const float* sample, const float* timestep, const float* ctx, const float* vector,
                     const float* pos, const int32_t* key_lens, float* out, int B, int H, int W, int L, int out_keep,
                     void* workspace, int64_t workspace_bytes, int flags, void* stream) {
  FDMI_CHECK(is_dit(U) && slot >= 0 && slot < 8, "dit: bad plan / slot");
  FDMI_CHECK(sample && timestep && ctx && pos && out && workspace, "dit: null argument");
  Run& R = U->runs[slot];
  R.arena.base = (char*)workspace;
  R.arena.cap = (size_t)workspace_bytes;
  R.arena.dry = false;
  R.st = (hipStream_t)stream;
  DitIn in{};
  in.x = sample; in.t = timestep; in.ctx = ctx; in.vec = vector; in.pos = pos; in.out = out;
  in.B = B; in.H = H; in.W = W; in.L = L; in.keep = out_keep;
  if (key_lens) in.lens.assign(key_lens, key_lens + B);
  return run_dit(U, R, in, flags);
}
int fdmi_dit_backward(fdmi_unet* U, int slot, const float* grad_out, float* grad_sample, void* stream) {
  FDMI_CHECK(is_dit(U) && slot >= 0 && slot < 8 && grad_out, "dit: bad plan / slot / null grad");
  Run& R = U->runs[slot];
  R.st = (hipStream_t)stream;
  return run_dit_backward(U, R, grad_out, grad_sample);
}

// ---- the frozen teacher's CFG loop without a host round trip between steps (FD:288-324) ----------------------------
static size_t tl_align(size_t x) { return (x + 255) & ~(size_t)255; }
// ... for a transformer denoiser (the PixArt recipe's DPM-Solver++ loop, flash_diffusion_model.py:288-324; the SD3 recipe's
// flow-matching Euler loop, flash_diffusion_sd3_model.py:282-314 -- x0 = the step's increment, a3 = a4 = 1)
int64_t fdmi_dit_teacher_loop_scratch_bytes(fdmi_unet* U, int B, int H, int W) {
  if (!is_dit(U) || B <= 0 || H <= 0 || W <= 0) return -1;
  const size_t per = (size_t)U->dcfg.in_channels * H * W;
  return (int64_t)(2 * tl_align(2 * B * per * 4) + tl_align(2 * (size_t)B * 4) + 2 * tl_align(B * per * 4));
}
int fdmi_dit_teacher_loop(fdmi_unet* U, int slot, float* x, const float* timesteps, int n, const float* ctx2, const float* vector2,
                          const float* pos, const int32_t* key_lens2, const float* coeffs, int B, int H, int W, int L,
                          void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream) {
  FDMI_CHECK(is_dit(U) && slot >= 0 && slot < 8, "dit: bad plan / slot");
  FDMI_CHECK(x && timesteps && ctx2 && pos && coeffs && workspace && scratch && n > 0, "dit teacher_loop: null argument");
  FDMI_CHECK(U->dcfg.in_channels <= U->dcfg.out_channels, "dit teacher_loop: the update needs out_channels >= in_channels");
  FDMI_CHECK(scratch_bytes >= fdmi_dit_teacher_loop_scratch_bytes(U, B, H, W), "dit teacher_loop: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t per = (size_t)U->dcfg.in_channels * H * W;
  const int64_t nel = (int64_t)B * per;
  char* sp = (char*)scratch;
  float* xx = (float*)sp;            sp += tl_align(2 * nel * 4);
  float* eps = (float*)sp;           sp += tl_align(2 * nel * 4);
  float* tt = (float*)sp;            sp += tl_align(2 * (size_t)B * 4);
  float* x0[2];
  x0[0] = (float*)sp;                sp += tl_align(nel * 4);
  x0[1] = (float*)sp;
  Run& R = U->runs[slot];
  R.arena.base = (char*)workspace;
  R.arena.cap = (size_t)workspace_bytes;
  R.arena.dry = false;
  R.st = st;
  DitIn in{};
  in.x = xx; in.t = tt; in.ctx = ctx2; in.vec = vector2; in.pos = pos; in.out = eps;
  in.B = 2 * B; in.H = H; in.W = W; in.L = L; in.keep = U->dcfg.in_channels;
  if (key_lens2) in.lens.assign(key_lens2, key_lens2 + 2 * B);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const float* a = coeffs + (size_t)i * 6;
    FDMI_HIP(hipMemcpyAsync(xx, x, nel * 4, hipMemcpyDeviceToDevice, st));          // [x | x]
    FDMI_HIP(hipMemcpyAsync(xx + nel, x, nel * 4, hipMemcpyDeviceToDevice, st));
    uint32_t tbits;
    memcpy(&tbits, &timesteps[i], 4);
    FDMI_HIP(hipMemsetD32Async((hipDeviceptr_t)tt, (int)tbits, 2 * (size_t)B, st));
    int rc = run_dit(U, R, in, 0);
    if (rc) return rc;
    flops += U->last_flops;
    float* cur = x0[i & 1];
    const float* prev = x0[(i & 1) ^ 1];
    // x0 = a0 x + a1 eps_c + a2 eps_u ;  x = a3 x + a4 x0 + a5 x0_prev
    rc = launch_axpby4(x, a[0], eps, a[1], eps + nel, a[2], nullptr, 0.f, cur, nel, st);
    if (rc) return rc;
    const bool use_prev = i > 0 && a[5] != 0.f;
    rc = launch_axpby4(x, a[3], cur, a[4], use_prev ? prev : nullptr, use_prev ? a[5] : 0.f, nullptr, 0.f, x, nel, st);
    if (rc) return rc;
  }
  U->last_flops = flops;
  return 0;
}
int64_t fdmi_teacher_loop_scratch_bytes(fdmi_unet* U, int B, int H, int W) {
  if (!U || B <= 0 || H <= 0 || W <= 0) return -1;
  const size_t per_in = (size_t)U->cfg.in_channels * H * W, per_out = (size_t)U->cfg.out_channels * H * W;
  return (int64_t)(tl_align(2 * B * per_in * 4) + tl_align(2 * B * per_out * 4) + tl_align(2 * (size_t)B * 4) +
                   2 * tl_align(B * per_out * 4));
}
int fdmi_teacher_loop(fdmi_unet* U, int slot, float* x, const float* timesteps, int n, const float* ctx2,
                      const float* cls2, const float* coeffs, int B, int H, int W, int L, void* workspace,
                      int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream) {
  FDMI_CHECK(U && slot >= 0 && slot < 8, "bad plan / slot");
  FDMI_CHECK(x && timesteps && ctx2 && coeffs && workspace && scratch && n > 0, "teacher_loop: null argument");
  FDMI_CHECK(U->cfg.in_channels == U->cfg.out_channels, "teacher_loop: the update needs out_channels == in_channels");
  FDMI_CHECK(scratch_bytes >= fdmi_teacher_loop_scratch_bytes(U, B, H, W), "teacher_loop: scratch too small");
  FDMI_CHECK(H % (1 << (U->nl - 1)) == 0 && W % (1 << (U->nl - 1)) == 0, "unet: H, W must be divisible by 2^(levels-1)");
  hipStream_t st = (hipStream_t)stream;
  const size_t per = (size_t)U->cfg.in_channels * H * W;
  const int64_t nel = (int64_t)B * per;
  char* sp = (char*)scratch;
  float* xx = (float*)sp;            sp += tl_align(2 * nel * 4);
  float* eps = (float*)sp;           sp += tl_align(2 * nel * 4);
  float* tt = (float*)sp;            sp += tl_align(2 * (size_t)B * 4);
  float* x0[2];
  x0[0] = (float*)sp;                sp += tl_align(nel * 4);
  x0[1] = (float*)sp;
  Run& R = U->runs[slot];
  R.arena.base = (char*)workspace;
  R.arena.cap = (size_t)workspace_bytes;
  R.arena.dry = false;
  R.st = st;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const float* a = coeffs + (size_t)i * 6;
    FDMI_HIP(hipMemcpyAsync(xx, x, nel * 4, hipMemcpyDeviceToDevice, st));          // [x | x]
    FDMI_HIP(hipMemcpyAsync(xx + nel, x, nel * 4, hipMemcpyDeviceToDevice, st));
    uint32_t tbits;
    memcpy(&tbits, &timesteps[i], 4);
    FDMI_HIP(hipMemsetD32Async((hipDeviceptr_t)tt, (int)tbits, 2 * (size_t)B, st));
    int rc = run_forward(U, R, xx, tt, ctx2, cls2, eps, 2 * B, H, W, L,
                         (i == 0 ? FDMI_UNET_CTX_FILL : FDMI_UNET_CTX_REUSE) | (fdmi_tune_get(13) ? 0 : FDMI_UNET_CFG_HALVES));
    if (rc) return rc;
    flops += U->last_flops;
    float* cur = x0[i & 1];
    const float* prev = x0[(i & 1) ^ 1];
    // x0 = a0 x + a1 eps_c + a2 eps_u ;  x = a3 x + a4 x0 + a5 x0_prev
    rc = launch_axpby4(x, a[0], eps, a[1], eps + nel, a[2], nullptr, 0.f, cur, nel, st);
    if (rc) return rc;
    const bool use_prev = i > 0 && a[5] != 0.f;
    rc = launch_axpby4(x, a[3], cur, a[4], use_prev ? prev : nullptr, use_prev ? a[5] : 0.f, nullptr, 0.f, x, nel, st);
    if (rc) return rc;
  }
  U->last_flops = flops;
  return 0;
}

}  // extern "C"
