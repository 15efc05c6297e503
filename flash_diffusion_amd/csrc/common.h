// Shared device/host helpers for the fdmi kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16, 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define FDMI_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16 (round-to-nearest-even) through the native __bf16 type: lowers to the gfx950
// v_cvt_pk_bf16_f32 instruction (one VALU op per PAIR instead of ~6 integer ops per element).
typedef float fdmi_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 fdmi_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const fdmi_f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fdmi_b2));
}
// (v_rcp_f32, 1 ulp: the IEEE-exact `/` and __frcp_rn expand to a ~10-instruction v_div_scale / v_div_fmas / v_div_fixup
// sequence per element, which made the GEGLU epilogue as long as its main loop at K = 320)
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float x) { return x * rcp_fast(1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below bf16/fp32-epilogue resolution): one exp,
// one rcp and five fmas instead of the ~40-instruction branchy libm erff in the GEGLU epilogues.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = rcp_fast(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
// exact-GELU for the GEGLU epilogues (round 4; they are VALU-bound: half of a K = 320 GEGLU launch is its epilogue).
// erfc(z) = 2^(-Q(z)) with Q a degree-6 polynomial without constant term, fitted on [0, 4] (weighted least squares on the
// absolute error of erf; evaluated in fp32: |error of erf| <= 3.1e-7, the Abramowitz-Stegun form above has 1.5e-7 on paper and
// 4.6e-7 through gelu in fp32 -- this one 5.1e-7).  With z = |x| / sqrt(2) folded into the coefficients and the 1/2 of
// 0.5 x (1 + erf) folded into the exponent:   gelu(x) = max(x, 0) - |x| * 2^(-(1 + |x| P(|x|))),
// one v_exp_f32 (natively base 2) and six fmas per element -- no reciprocal, no second transcendental, no copysign.
// |x| is clamped to the fitted range (beyond it erfc < 2e-8 and the polynomial's negative leading term would take over).
__device__ __forceinline__ float gelu_f(float x) {
  const float a = fabsf(x);
  const float c = fminf(a, 5.65685425f);
  float t = -1.98600017e-05f;
  t = fmaf(t, c, 0.000662309048f);
  t = fmaf(t, c, -0.00775970362f);
  t = fmaf(t, c, 0.0529644412f);
  t = fmaf(t, c, 0.459066224f);
  t = fmaf(t, c, 1.1511191f);
  const float q1 = fmaf(t, c, 1.f);
  return fmaf(-a, __builtin_amdgcn_exp2f(-q1), fmaxf(x, 0.f));
}
// the same function of TWO values at once (round 6): the polynomial and the final fma as packed fp32 operations (v_pk_fma_f32: two
// per lane per issue slot; the GEGLU epilogue of a K = 320 launch is ~85 us of VALU issue out of 282 us); every operation is the
// scalar one per component, so the results are bit-identical to gelu_f
typedef float fdmi_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fdmi_f2 gelu_f2(fdmi_f2 x) {
  const fdmi_f2 a = {fabsf(x.x), fabsf(x.y)};
  const fdmi_f2 c = {fminf(a.x, 5.65685425f), fminf(a.y, 5.65685425f)};
  fdmi_f2 t = {-1.98600017e-05f, -1.98600017e-05f};
  t = __builtin_elementwise_fma(t, c, (fdmi_f2){0.000662309048f, 0.000662309048f});
  t = __builtin_elementwise_fma(t, c, (fdmi_f2){-0.00775970362f, -0.00775970362f});
  t = __builtin_elementwise_fma(t, c, (fdmi_f2){0.0529644412f, 0.0529644412f});
  t = __builtin_elementwise_fma(t, c, (fdmi_f2){0.459066224f, 0.459066224f});
  t = __builtin_elementwise_fma(t, c, (fdmi_f2){1.1511191f, 1.1511191f});
  const fdmi_f2 q1 = __builtin_elementwise_fma(t, c, (fdmi_f2){1.f, 1.f});
  const fdmi_f2 e = {__builtin_amdgcn_exp2f(-q1.x), __builtin_amdgcn_exp2f(-q1.y)};
  return __builtin_elementwise_fma(-a, e, (fdmi_f2){fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)});
}
// tanh-approximated GELU: 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3): one exp, one rcp
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float u2 = 1.5957691216057308f * fmaf(0.044715f * x * x, x, x);
  return x * rcp_fast(1.f + __expf(-u2));
}
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf = 0.5f * (1.f + erf_fast(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}


// ---- host side -------------------------------------------------------------------------
void fdmi_set_error(const std::string& msg);
#define FDMI_CHECK(cond, msg)                                                         \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      fdmi_set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)
#define FDMI_HIP(call)                                                                \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      fdmi_set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " +   \
                     hipGetErrorString(e_));                                          \
      return -2;                                                                      \
    }                                                                                 \
  } while (0)

// optional per-launch HIP-event profiling (bench.py roofline leg); see capi.hip
enum { PROF_GEMM0 = 0 /* +mode*4 + tile */, PROF_ATTN_FWD = 8, PROF_ATTN_DQ = 9, PROF_ATTN_DKV = 10,
       PROF_GEMM3 = 11 /* + mode*2 + (BN==128) */, PROF_GEMM4 = 15 /* + mode (256x320) */, PROF_GEMM4_192 = 17 /* + mode */,
       PROF_WGRAD_TN = 19,
       // round 6: SUBSET buckets -- the row GEMMs of the 256-row ring kernels whose arithmetic intensity lies on the HBM side of the
       // 2.5 PFLOP/s : 8 TB/s ridge (312.5 flop per algorithmic byte) are counted a second time here (their family bucket keeps
       // them too), so that the bench line can price them against the bound they sit under: N = K = 320 with
       // a residual has 107 flop/B, K = 1280 -> 320 has 213
       PROF_GEMM4_ROW_HBM = 20, PROF_GEMM3_ROW_HBM = 21,
       PROF_GEMM5 = 22 /* the 128 x 320 two-blocks-per-CU row kernel (gemm5.hip) */, PROF_GEMM5_HBM = 23 /* its HBM-side subset */, PROF_NBUCKETS = 24 };
constexpr double FDMI_RIDGE_FLOP_PER_BYTE = 2.5e15 / 8.0e12;
int fdmi_tune_get(int key);   // developer tuning knobs (fdmi_tune_set)
// Deterministic mode (knob 50 = 1; round 6, VERDICT r5 item 5 / ADVICE r5): every floating-point accumulation whose ORDER the
// production kernels leave to the hardware -- fp32 atomics of the GroupNorm-sum epilogues, of gn_reduce's blocks, of the TN
// weight-gradient row splits, of the column sums and of the scalar losses -- runs in a fixed order instead (one contributor per
// output element, ordered in-block reductions), so that two runs of a step are BIT-IDENTICAL.  A test / debugging mode: slower
// (single row split per weight-gradient tile, one block per sample in the statistics passes), same kernels otherwise.
static inline bool fdmi_det() { return fdmi_tune_get(50) != 0; }
bool fdmi_prof_on();
// bytes: algorithmic HBM bytes of the launch (every operand once); subset: a second bucket that also counts the launch (-1: none)
void fdmi_prof_begin(hipStream_t st, int bucket, double flops, double bytes = 0.0, int subset = -1);
void fdmi_prof_end(hipStream_t st);
// round 6: the SHAPE of the launch the next fdmi_prof_begin records (kind 0 row GEMM / 1 conv: M, N, K, flags = residual | GEGLU << 1 |
// dgrad << 2 | GroupNorm sums << 3 | split-K << 8; kind 2 / 3 / 4 attention forward / dQ / dK-dV: B * H, Sq, Skv, d); fdmi_prof_dump
// writes one line per launch of the last collected leg -- the per-shape table of scripts/shape_table.py
void fdmi_prof_shape(int kind, long long s0, long long s1, long long s2, long long s3);
bool fdmi_prof_take(hipEvent_t* a, hipEvent_t* b);
// A profiled launch carries its own start / stop events (hipExtLaunchKernelGGL: the timestamps of the dispatch itself -- what
// rocprofv3's kernel trace reports) instead of two event packets around it, whose elapsed time also holds the dispatch latency
// on either side (several us per launch; developer knob 20 = 1 goes back to the brackets).  `prof` false: the plain launch.
#define FDMI_KLAUNCH(prof, kernel, grid, block, smem, st, ...)                                              \
  do {                                                                                                      \
    hipEvent_t pa_ = nullptr, pb_ = nullptr;                                                                \
    if ((prof) && fdmi_prof_take(&pa_, &pb_))                                                               \
      hipExtLaunchKernelGGL(kernel, grid, block, smem, st, pa_, pb_, 0, __VA_ARGS__);                       \
    else                                                                                                    \
      hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                                       \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
