// extern "C" surface of libfdmi.so (op-level entry points).  See include/fdmi.h.
#include "../../include/fdmi.h"
#include "ops.h"
#include "wgrad.h"

static thread_local std::string g_err;
void fdmi_set_error(const std::string& msg) { g_err = msg; }

// ---- per-launch event profiling -------------------------------------------------------------
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int bucket; double bytes; int subset; int kind; long long s[4]; };
struct ProfRow { int bucket, kind; long long s[4]; float ms; double flops, bytes; };
std::vector<ProfRow> g_last;   // the launches of the last collected leg (fdmi_prof_dump)
int g_kind = -1;
long long g_shape[4] = {0, 0, 0, 0};
bool g_prof = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
static int g_tune[64] = {0};
int fdmi_tune_get(int key) { return (key >= 0 && key < 64) ? g_tune[key] : 0; }
bool fdmi_prof_on() { return g_prof; }
// tune 20 = 0 (default): the launch that follows takes the record's two events as its own start / stop events
// (FDMI_KLAUNCH -> hipExtLaunchKernelGGL); 1: the events are recorded on the stream around the launch
static bool g_take = false;
void fdmi_prof_begin(hipStream_t st, int bucket, double flops, double bytes, int subset) {
  ProfRec r{prof_event(), prof_event(), flops, bucket, bytes, subset, g_kind, {g_shape[0], g_shape[1], g_shape[2], g_shape[3]}};
  g_kind = -1;
  if (fdmi_tune_get(20)) (void)hipEventRecord(r.a, st);
  else g_take = true;
  g_recs.push_back(r);
}
void fdmi_prof_shape(int kind, long long s0, long long s1, long long s2, long long s3) {
  g_kind = kind; g_shape[0] = s0; g_shape[1] = s1; g_shape[2] = s2; g_shape[3] = s3;
}
bool fdmi_prof_take(hipEvent_t* a, hipEvent_t* b) {
  if (!g_take) return false;
  g_take = false;
  *a = g_recs.back().a; *b = g_recs.back().b;
  return true;
}
void fdmi_prof_end(hipStream_t st) {
  if (fdmi_tune_get(20)) (void)hipEventRecord(g_recs.back().b, st);
  else if (g_take) { g_take = false; g_pool.push_back(g_recs.back().a); g_pool.push_back(g_recs.back().b); g_recs.pop_back(); }  // no launch took them
}

extern "C" {

int fdmi_tune_set(int key, int value) {
  FDMI_CHECK(key >= 0 && key < 64, "tune key out of range");
  g_tune[key] = value;
  return 0;
}
int fdmi_tune_value(int key) { return fdmi_tune_get(key); }
int fdmi_prof_enable(int on) { g_prof = on != 0; return 0; }
int fdmi_prof_collect2(int nbuckets, double* ms, double* flops, int64_t* launches, double* bytes);
int fdmi_prof_collect(int nbuckets, double* ms, double* flops, int64_t* launches) {
  return fdmi_prof_collect2(nbuckets, ms, flops, launches, nullptr);
}
int fdmi_prof_collect2(int nbuckets, double* ms, double* flops, int64_t* launches, double* bytes) {
  FDMI_CHECK(nbuckets >= PROF_NBUCKETS, "prof_collect: need >= PROF_NBUCKETS (24) buckets");
  for (int i = 0; i < nbuckets; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; if (bytes) bytes[i] = 0; }
  FDMI_HIP(hipDeviceSynchronize());
  int bad = 0;
  g_last.clear();
  for (auto& r : g_recs) {
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && t >= 0.f) {
      g_last.push_back(ProfRow{r.bucket, r.kind, {r.s[0], r.s[1], r.s[2], r.s[3]}, t, r.flops, r.bytes});
      ms[r.bucket] += t; flops[r.bucket] += r.flops; launches[r.bucket] += 1;
      if (bytes) bytes[r.bucket] += r.bytes;
      if (bytes && r.subset >= 0 && r.subset < nbuckets) {   // (subset buckets exist only for the caller that asks for bytes)
        ms[r.subset] += t; flops[r.subset] += r.flops; launches[r.subset] += 1; bytes[r.subset] += r.bytes;
      }
    } else {
      ++bad;
    }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  (void)hipGetLastError();
  FDMI_CHECK(bad == 0, "prof_collect: " + std::to_string(bad) + " launch records without a valid elapsed time");
  return 0;
}

int fdmi_prof_dump(const char* path) {
  FDMI_CHECK(path != nullptr, "prof_dump: null path");
  FILE* f = fopen(path, "w");
  FDMI_CHECK(f != nullptr, std::string("prof_dump: cannot open ") + path);
  fprintf(f, "bucket,kind,s0,s1,s2,s3,ms,flops,bytes\n");
  for (auto& r : g_last)
    fprintf(f, "%d,%d,%lld,%lld,%lld,%lld,%.6f,%.0f,%.0f\n", r.bucket, r.kind, r.s[0], r.s[1], r.s[2], r.s[3], r.ms, r.flops, r.bytes);
  fclose(f);
  return (int)g_last.size();
}

const char* fdmi_last_error(void) { return g_err.c_str(); }
int fdmi_version(void) { return 1; }

static GemmArgs gemm_args_from(const fdmi_gemm_desc* d) {
  GemmArgs a;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.A = (const bf16_t*)d->A; a.lda = d->lda;
  a.W = (const bf16_t*)d->W; a.ldw = d->ldw;
  a.mode = d->mode;
  a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->Cin; a.Hout = d->Hout; a.Wout = d->Wout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.ups = d->ups; a.dgrad = d->dgrad;
  a.bias = d->bias;
  a.rowvec = (const bf16_t*)d->rowvec; a.rowvec_ld = d->rowvec_ld; a.rows_per_batch = d->rows_per_batch;
  a.residual = (const bf16_t*)d->residual; a.ldr = d->ldr;
  a.act = d->act;
  a.preact = (bf16_t*)d->preact; a.ldp = d->ldp;
  a.C = d->C; a.ldc = d->ldc; a.out_f32 = d->out_f32;
  a.alpha = d->alpha;
  a.splitk = d->splitk; a.ws = d->ws;
  a.accum_atomic = d->accum_atomic;
  a.force_tile = d->force_tile;
  a.use_glds = d->use_glds;
  a.A2 = (const bf16_t*)d->A2; a.lda2 = d->lda2; a.K1 = d->K1;
  a.rowvec_mul = d->rowvec_mul;
  return a;
}
int fdmi_gemm_a2_ok(const fdmi_gemm_desc* d) { return d != nullptr && gemm_a2_ok(gemm_args_from(d)) ? 1 : 0; }
int fdmi_gemm(const fdmi_gemm_desc* d, void* stream) {
  FDMI_CHECK(d != nullptr, "null descriptor");
  return launch_gemm(gemm_args_from(d), (hipStream_t)stream);
}
// host-only: which kernel / tile / split the planner would launch for this problem (no device work)
int fdmi_gemm_plan(const fdmi_gemm_desc* d, int32_t* kernel, int32_t* BM, int32_t* BN, int32_t* splitk) {
  FDMI_CHECK(d != nullptr && kernel && BM && BN && splitk, "gemm_plan: null argument");
  const GemmArgs a = gemm_args_from(d);
  const GemmPlan p = plan_gemm(a, a.ws != nullptr || a.accum_atomic || a.splitk <= 0);
  *kernel = p.big; *BM = p.big ? 256 : p.BM; *BN = p.BN; *splitk = p.splitk;
  return 0;
}

int fdmi_wgrad_tn(const void* X, int64_t ldx, const void* Y, int64_t ldy, int64_t M, int N1, int N2, float* C, int64_t ldc,
                  void* stream) {
  return launch_wgrad_tn((const bf16_t*)X, ldx, (const bf16_t*)Y, ldy, M, N1, N2, C, ldc, (hipStream_t)stream);
}
int fdmi_wgrad_tn_group(const fdmi_wgrad_problem* problems, int n, void* stream) {
  FDMI_CHECK(problems != nullptr && n >= 1 && n <= WGRAD_GROUP_MAX, "wgrad_tn_group: 1 ... 6 problems");
  WgradProblem pr[WGRAD_GROUP_MAX];
  for (int i = 0; i < n; ++i)
    pr[i] = WgradProblem{(const bf16_t*)problems[i].X, problems[i].ldx, (const bf16_t*)problems[i].Y, problems[i].ldy, problems[i].M,
                         problems[i].N1, problems[i].N2, problems[i].C, problems[i].ldc};
  return launch_wgrad_tn_group(pr, n, (hipStream_t)stream);
}
static GemmArgs gemm_gn_args_from(const fdmi_gemm_desc* d, float* gn_stats, int gn_rows, int gn_G) {
  GemmArgs a = gemm_args_from(d);
  a.gn_stats = gn_stats; a.gn_rows = gn_rows; a.gn_G = gn_G;
  a.gn_cpg = gn_G > 0 ? a.N / gn_G : 0;
  return a;
}
int fdmi_gemm_gn_ok(const fdmi_gemm_desc* d, int gn_rows, int gn_G) {
  if (!d) return 0;
  const GemmArgs a = gemm_gn_args_from(d, (float*)(uintptr_t)256, gn_rows, gn_G);   // (host-only query: any non-null pointer)
  return gemm_gn_ok(a, a.ws != nullptr || a.accum_atomic) ? 1 : 0;
}
int fdmi_gemm_gn(const fdmi_gemm_desc* d, float* gn_stats, int gn_rows, int gn_G, void* stream) {
  FDMI_CHECK(d != nullptr && gn_stats != nullptr, "gemm_gn: null argument");
  return launch_gemm(gemm_gn_args_from(d, gn_stats, gn_rows, gn_G), (hipStream_t)stream);
}
int fdmi_groupnorm_apply(const void* x, const float* gamma, const float* beta, const float* stats, void* y, int B, int HW,
                         int C, int G, float eps, int silu, void* stream) {
  return launch_groupnorm_fwd((const bf16_t*)x, gamma, beta, (float*)stats, (bf16_t*)y, B, HW, C, G, eps, silu,
                              (hipStream_t)stream, true, true);
}
int fdmi_groupnorm_fwd(const void* x, const float* gamma, const float* beta, float* stats, void* y, int B,
                       int HW, int C, int G, float eps, int silu, void* stream) {
  return launch_groupnorm_fwd((const bf16_t*)x, gamma, beta, stats, (bf16_t*)y, B, HW, C, G, eps, silu,
                              (hipStream_t)stream);
}
int fdmi_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                       const float* stats, float* bstats, void* dx, int B, int HW, int C, int G, float eps,
                       int silu, int accumulate, void* stream) {
  return launch_groupnorm_bwd((const bf16_t*)x, (const bf16_t*)dy, gamma, beta, stats, bstats, (bf16_t*)dx, B,
                              HW, C, G, eps, silu, accumulate, (hipStream_t)stream);
}
int fdmi_groupnorm_cat_fwd(const void* x1, const void* x2, int C1, const float* gamma, const float* beta, float* stats, void* y,
                           int B, int HW, int C, int G, float eps, int silu, void* stream) {
  return launch_groupnorm_fwd((const bf16_t*)x1, gamma, beta, stats, (bf16_t*)y, B, HW, C, G, eps, silu, (hipStream_t)stream,
                              false, false, (const bf16_t*)x2, C1);
}
int fdmi_groupnorm_cat_bwd(const void* x1, const void* x2, int C1, const void* dy, const float* gamma, const float* beta,
                           const float* stats, float* bstats, void* dx, int B, int HW, int C, int G, float eps, int silu,
                           int accumulate, void* stream) {
  return launch_groupnorm_bwd((const bf16_t*)x1, (const bf16_t*)dy, gamma, beta, stats, bstats, (bf16_t*)dx, B, HW, C, G, eps,
                              silu, accumulate, (hipStream_t)stream, false, (const bf16_t*)x2, C1);
}
int fdmi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int C,
                       float eps, void* stream) {
  return launch_layernorm_fwd((const bf16_t*)x, gamma, beta, nullptr, nullptr, 0, 1, (bf16_t*)y, rows, C, eps,
                              (hipStream_t)stream);
}
int fdmi_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, int64_t rows, int C,
                       float eps, int accumulate, void* stream) {
  return launch_layernorm_bwd((const bf16_t*)x, (const bf16_t*)dy, gamma, nullptr, 0, 1, (bf16_t*)dx, rows, C,
                              eps, accumulate, (hipStream_t)stream);
}
int fdmi_layernorm_mod_fwd(const void* x, const void* shift, const void* scale, int64_t mod_ld, int rows_per_batch,
                           void* y, float* stats, int64_t rows, int C, float eps, void* stream) {
  FDMI_CHECK(shift && scale, "layernorm_mod_fwd: shift and scale are required");
  return launch_layernorm_fwd((const bf16_t*)x, nullptr, nullptr, (const bf16_t*)shift, (const bf16_t*)scale, mod_ld,
                              rows_per_batch, (bf16_t*)y, rows, C, eps, (hipStream_t)stream, stats);
}
int fdmi_layernorm_mod_bwd(const void* x, const void* dy, const void* scale, int64_t mod_ld, int rows_per_batch, void* dx,
                           int64_t rows, int C, float eps, int accumulate, void* stream) {
  FDMI_CHECK(scale != nullptr, "layernorm_mod_bwd: scale is required");
  return launch_layernorm_bwd((const bf16_t*)x, (const bf16_t*)dy, nullptr, (const bf16_t*)scale, mod_ld, rows_per_batch,
                              (bf16_t*)dx, rows, C, eps, accumulate, (hipStream_t)stream);
}
int fdmi_gate_residual(const void* x, const void* gate, int64_t gate_ld, const void* res, void* y, int64_t rows, int C,
                       int rows_per_batch, void* stream) {
  return launch_gate_residual((const bf16_t*)x, (const bf16_t*)gate, gate_ld, (const bf16_t*)res, (bf16_t*)y, rows, C,
                              rows_per_batch, (hipStream_t)stream);
}
int fdmi_gelu_tanh(const void* x, void* y, int64_t n, void* stream) {
  return launch_gelu_tanh((const bf16_t*)x, (bf16_t*)y, n, (hipStream_t)stream);
}
int fdmi_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  return launch_gelu_tanh_bwd((const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, (hipStream_t)stream);
}
int fdmi_batch_colsum(const void* dy, const void* x, const float* stats, float* out0, float* out1, int B,
                      int rows_per_batch, int C, void* stream) {
  return launch_batch_colsum((const bf16_t*)dy, (const bf16_t*)x, stats, out0, out1, B, rows_per_batch, C,
                             (hipStream_t)stream);
}

int64_t fdmi_attn_tr_elems(int B, int H, int S, int d) {
  return (int64_t)B * H * attn_dvpad(d) * attn_spad(S);
}
static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
int64_t fdmi_attn_bwd_ws_bytes(int B, int H, int Sq, int Skv, int d) {
  // QT, dOT (Sq), KT (Skv), delta
  return align256(fdmi_attn_tr_elems(B, H, Sq, d) * 2) * 2 + align256(fdmi_attn_tr_elems(B, H, Skv, d) * 2) +
         align256((int64_t)B * H * Sq * 4);
}
int fdmi_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                  int64_t ldo, void* VT, float* lse, int B, int H, int Sq, int Skv, int d, float scale,
                  void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_transpose_heads((const bf16_t*)V, ldv, (bf16_t*)VT, B, H, Skv, d, st, 1);
  if (rc) return rc;
  AttnArgs a{};
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.VT = (const bf16_t*)VT; a.lse = lse; a.out = (bf16_t*)O; a.ldout = ldo; a.vt_ones = 1;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.d = d; a.scale = scale;
  return launch_attn_fwd(a, st);
}
int fdmi_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  const void* O, int64_t ldo, const void* dO, int64_t lddo, const float* lse, void* dQ,
                  int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv, void* ws, int B, int H, int Sq,
                  int Skv, int d, float scale, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  bf16_t* QT = (bf16_t*)w;  w += align256(fdmi_attn_tr_elems(B, H, Sq, d) * 2);
  bf16_t* dOT = (bf16_t*)w; w += align256(fdmi_attn_tr_elems(B, H, Sq, d) * 2);
  bf16_t* KT = (bf16_t*)w;  w += align256(fdmi_attn_tr_elems(B, H, Skv, d) * 2);
  float* delta = (float*)w;
  int rc;
  if ((rc = launch_transpose_heads((const bf16_t*)Q, ldq, QT, B, H, Sq, d, st))) return rc;
  if ((rc = launch_transpose_heads((const bf16_t*)dO, lddo, dOT, B, H, Sq, d, st))) return rc;
  if ((rc = launch_transpose_heads((const bf16_t*)K, ldk, KT, B, H, Skv, d, st))) return rc;
  if ((rc = launch_attn_delta((const bf16_t*)O, ldo, (const bf16_t*)dO, lddo, delta, B, H, Sq, d, st))) return rc;
  AttnArgs a{};
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.dO = (const bf16_t*)dO; a.lddo = lddo; a.QT = QT; a.KT = KT; a.dOT = dOT;
  a.lse = (float*)lse; a.delta = delta;
  a.out = (bf16_t*)dQ; a.ldout = lddq; a.dK = (bf16_t*)dK; a.lddk = lddk; a.dV = (bf16_t*)dV; a.lddv = lddv;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.d = d; a.scale = scale;
  if ((rc = launch_attn_bwd_dq(a, st))) return rc;
  return launch_attn_bwd_dkv(a, st);
}

int fdmi_nchw_to_nhwc(const float* x, void* y, int B, int C, int HW, int Cpad, void* stream) {
  return launch_nchw_to_nhwc(x, (bf16_t*)y, B, C, HW, Cpad, (hipStream_t)stream);
}
int fdmi_nhwc_to_nchw(const void* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate, void* stream) {
  return launch_nhwc_to_nchw((const bf16_t*)x, ldx, y, B, C, HW, accumulate, (hipStream_t)stream);
}
int fdmi_timestep_embed(const float* t, void* out, int B, int dim, int flip, float shift, void* stream) {
  return launch_timestep_embed(t, (bf16_t*)out, B, dim, flip, shift, (hipStream_t)stream);
}
int fdmi_geglu_bwd(const void* pre, const void* dout, void* dpre, int64_t M, int F, void* stream) {
  return launch_geglu_bwd((const bf16_t*)pre, (const bf16_t*)dout, (bf16_t*)dpre, M, F, (hipStream_t)stream);
}
int fdmi_pool2x2_sum(const void* dy, void* dx, int B, int H, int W, int C, int accumulate, void* stream) {
  return launch_pool2x2_sum((const bf16_t*)dy, (bf16_t*)dx, B, H, W, C, accumulate, (hipStream_t)stream);
}
int fdmi_cast_transpose(const float* w, void* wb, void* wtb, int rows, int cols, void* stream) {
  return launch_cast_transpose(w, (bf16_t*)wb, (bf16_t*)wtb, rows, cols, (hipStream_t)stream);
}
int fdmi_transpose2d(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, void* stream) {
  return launch_transpose2d((const bf16_t*)in, ldi, (bf16_t*)out, ldo, rows, cols, (hipStream_t)stream);
}
int fdmi_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float grad_scale, void* stream) {
  return launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)stream);
}
int fdmi_add_noise(const float* z, const float* noise, const float* sa, const float* sb, float* out, int B,
                   int64_t per, void* stream) {
  return launch_add_noise(z, noise, sa, sb, out, B, per, (hipStream_t)stream);
}
int fdmi_axpby4(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                const float* x3, float c3, float* out, int64_t n, void* stream) {
  return launch_axpby4(x0, c0, x1, c1, x2, c2, x3, c3, out, n, (hipStream_t)stream);
}

int fdmi_im2col(const void* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                void* stream) {
  return launch_im2col((const bf16_t*)x, (bf16_t*)out, B, H, W, C, Ho, Wo, KH, KW, stride, pad, (hipStream_t)stream);
}
int fdmi_silu(const void* x, void* y, int64_t n, void* stream) {
  return launch_silu((const bf16_t*)x, (bf16_t*)y, n, (hipStream_t)stream);
}
int fdmi_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  return launch_silu_bwd((const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, (hipStream_t)stream);
}
int fdmi_colsum(const void* dy, const void* x, const float* stats, float* out0, float* out1, int64_t rows, int C, int HW,
                int G, float eps, void* stream) {
  return launch_colsum((const bf16_t*)dy, (const bf16_t*)x, stats, out0, out1, rows, C, HW, G, eps, (hipStream_t)stream);
}
int fdmi_transpose2d_pad(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int cols, int64_t rows_pad,
                         void* stream) {
  return launch_transpose2d_pad((const bf16_t*)in, ldi, (bf16_t*)out, ldo, rows, cols, rows_pad, (hipStream_t)stream);
}
int fdmi_pad_cols(const void* src, int cols, void* dst, int cols_pad, int64_t rows, void* stream) {
  return launch_pad_cols((const bf16_t*)src, cols, (bf16_t*)dst, cols_pad, rows, (hipStream_t)stream);
}
int fdmi_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  return launch_f32_to_bf16(x, (bf16_t*)y, n, (hipStream_t)stream);
}
int fdmi_distill_loss(const float* s, const float* t, int64_t n, int l1, float* out, void* stream) {
  return launch_distill_loss(s, t, n, l1, out, (hipStream_t)stream);
}
int fdmi_distill_grad(const float* s, const float* t, int64_t n, int l1, float gscale, float* ds, void* stream) {
  return launch_distill_grad(s, t, n, l1, gscale, ds, (hipStream_t)stream);
}
int fdmi_dmd_loss(const float* s, const float* noisy, const float* real, const float* fake, const float* inv_alpha,
                  const float* msig_alpha, const float* kb, float* w, float* grad, float* loss, int B, int64_t per,
                  void* stream) {
  return launch_dmd_loss(s, noisy, real, fake, inv_alpha, msig_alpha, kb, w, grad, loss, B, per, (hipStream_t)stream);
}

// ---- fp32 validation mode, op level (csrc/ref32.hip): float twins of the entry points above.  Operand conventions are the
// bf16 ones with float storage; GroupNorm statistics are (mean, rstd) per (sample, group). ----
int fdmi_gemm_f32(const fdmi_gemm_desc* d, void* stream) {
  FDMI_CHECK(d != nullptr, "null descriptor");
  GemmArgs a = gemm_args_from(d);
  a.f32 = 1; a.out_f32 = 1; a.splitk = 1; a.ws = nullptr;
  return launch_gemm32(a, (hipStream_t)stream);
}
int fdmi_wgrad_tn_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t M, int N1, int N2, float* C, int64_t ldc,
                      void* stream) {
  return launch_wgrad_tn32(X, ldx, Y, ldy, M, N1, N2, C, ldc, (hipStream_t)stream);
}
int64_t fdmi_attn_scratch_elems_f32(int B, int H, int Sq, int Skv, int bwd) { return attn32_scratch_elems(B, H, Sq, Skv, bwd); }
int fdmi_attn_fwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                      int B, int H, int Sq, int Skv, int d, float scale, float* scratch, int64_t scratch_elems, void* stream) {
  return launch_attn32_fwd(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Sq, Skv, d, scale, scratch, scratch_elems, (hipStream_t)stream);
}
int fdmi_attn_causal_fwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O,
                             int64_t ldo, int B, int H, int S, int d, float scale, float* scratch, int64_t scratch_elems, void* stream) {
  return launch_attn32_fwd(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, S, S, d, scale, scratch, scratch_elems, (hipStream_t)stream, 1);
}
int fdmi_attn_bias_fwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                           int B, int H, int Sq, int Skv, int d, float scale, const float* bias, const float* kbias, float* scratch,
                           int64_t scratch_elems, void* stream) {
  return launch_attn32_fwd(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Sq, Skv, d, scale, scratch, scratch_elems, (hipStream_t)stream, 0, bias,
                           kbias);
}
int fdmi_rmsnorm(const void* x, const float* w, void* y, int64_t rows, int C, float eps, void* stream) {
  return launch_rmsnorm((const bf16_t*)x, w, (bf16_t*)y, rows, C, eps, (hipStream_t)stream);
}
int fdmi_rmsnorm_f32(const float* x, const float* w, float* y, int64_t rows, int C, float eps, void* stream) {
  return launch_rmsnorm32(x, w, y, rows, C, eps, (hipStream_t)stream);
}
int fdmi_mul(const void* a, const void* b, void* y, int64_t n, void* stream) {
  return launch_ewise_mul((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n, (hipStream_t)stream);
}
int fdmi_mul_f32(const float* a, const float* b, float* y, int64_t n, void* stream) { return launch_ewise_mul32(a, b, y, n, (hipStream_t)stream); }
int fdmi_attn_bwd_f32(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, const float* dO,
                      int64_t lddo, float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV, int64_t lddv, int B, int H, int Sq,
                      int Skv, int d, float scale, float* scratch, int64_t scratch_elems, void* stream) {
  return launch_attn32_bwd(Q, ldq, K, ldk, V, ldv, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, B, H, Sq, Skv, d, scale, scratch,
                           scratch_elems, (hipStream_t)stream);
}
int fdmi_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* stats, float* y, int B, int HW, int C, int G,
                           float eps, int silu, void* stream) {
  return launch_groupnorm32_fwd(x, gamma, beta, stats, y, B, HW, C, G, eps, silu, (hipStream_t)stream);
}
int fdmi_groupnorm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* beta, const float* stats, float* dx,
                           int B, int HW, int C, int G, int silu, int accumulate, void* stream) {
  return launch_groupnorm32_bwd(x, dy, gamma, beta, stats, dx, B, HW, C, G, silu, accumulate, (hipStream_t)stream);
}
int fdmi_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* shift, const float* scale,
                           int64_t mod_ld, int rows_per_batch, float* y, float* stats, int64_t rows, int C, float eps, void* stream) {
  return launch_layernorm32_fwd(x, gamma, beta, shift, scale, mod_ld, rows_per_batch, y, rows, C, eps, (hipStream_t)stream, stats);
}
int fdmi_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* scale, int64_t mod_ld,
                           int rows_per_batch, float* dx, int64_t rows, int C, float eps, int accumulate, void* stream) {
  return launch_layernorm32_bwd(x, dy, gamma, scale, mod_ld, rows_per_batch, dx, rows, C, eps, accumulate, (hipStream_t)stream);
}
int fdmi_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int HW, int Cpad, void* stream) {
  return launch_nchw_to_nhwc32(x, y, B, C, HW, Cpad, (hipStream_t)stream);
}
int fdmi_nhwc_to_nchw_f32(const float* x, int64_t ldx, float* y, int B, int C, int HW, int accumulate, void* stream) {
  return launch_nhwc_to_nchw32(x, ldx, y, B, C, HW, accumulate, (hipStream_t)stream);
}
int fdmi_silu_f32(const float* x, float* y, int64_t n, void* stream) { return launch_silu32(x, y, n, (hipStream_t)stream); }
int fdmi_silu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  return launch_silu32_bwd(x, dy, dx, n, (hipStream_t)stream);
}
int fdmi_gelu_tanh_f32(const float* x, float* y, int64_t n, void* stream) { return launch_gelu_tanh32(x, y, n, (hipStream_t)stream); }
int fdmi_gelu_tanh_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  return launch_gelu_tanh32_bwd(x, dy, dx, n, (hipStream_t)stream);
}
int fdmi_gate_residual_f32(const float* x, const float* gate, int64_t gate_ld, const float* res, float* y, int64_t rows, int C,
                           int rows_per_batch, void* stream) {
  return launch_gate_residual32(x, gate, gate_ld, res, y, rows, C, rows_per_batch, (hipStream_t)stream);
}
int fdmi_batch_colsum_f32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int B, int rows_per_batch,
                          int C, void* stream) {
  return launch_batch_colsum32(dy, x, stats, out0, out1, B, rows_per_batch, C, (hipStream_t)stream);
}
int fdmi_im2col_f32(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad,
                    void* stream) {
  return launch_im2col32(x, out, B, H, W, C, Ho, Wo, KH, KW, stride, pad, (hipStream_t)stream);
}
int fdmi_colsum_f32(const float* dy, const float* x, const float* stats, float* out0, float* out1, int64_t rows, int C, int HW, int G,
                    void* stream) {
  return launch_colsum32(dy, x, stats, out0, out1, rows, C, HW, G, (hipStream_t)stream);
}
int fdmi_timestep_embed_f32(const float* t, float* out, int B, int dim, int flip, float shift, void* stream) {
  return launch_timestep_embed32(t, out, B, dim, flip, shift, (hipStream_t)stream);
}
int fdmi_pad_cols_f32(const float* src, int cols, float* dst, int cols_pad, int64_t rows, void* stream) {
  return launch_pad_cols32(src, cols, dst, cols_pad, rows, (hipStream_t)stream);
}

}  // extern "C"
