// Does the matrix pipe run beside the VALU on a CDNA4 SIMD?  (dev tool, round 5; the d = 40 attention forward shows MFMA-busy 42 % +
// VALU-active 50 % = 92 % of its wall: are its two instruction streams taking turns because of the kernel's structure, or is that
// what the SIMD does?)  One "tile" of work per loop trip, shaped like attn_fwd32_kernel<3,2> per 64 keys x 32 queries:
//   M: 14 v_mfma_f32_32x32x16_bf16 (two chains of 3, two chains of 4: K Q^T and V^T P)
//   V: 32 v_exp_f32 + 16 v_cvt_pk_bf16_f32 + 24 v_fma_f32 (exponentials, packing, the rest)
// modes: 0 = M only, 1 = V only, 2 = M then V in one wave (independent data, order pinned), 3 = M and V interleaved in one wave
// (one MFMA, then ~5 VALU), run at 1, 2 and 3 waves per SIMD on all 256 CUs.  Prints ns per tile per SIMD-resident wave set.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EXP8(a0, a1, a2, a3, a4, a5, a6, a7)                                                                         \
  asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"                         \
               "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"                           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define CVT4(a0, a1, a2, a3, a4, a5, a6, a7)                                                                         \
  asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define FMA6(a0, a1, a2, a3, a4, a5, c)                                                                              \
  asm volatile("v_fma_f32 %0, %0, %6, %6\n v_fma_f32 %1, %1, %6, %6\n v_fma_f32 %2, %2, %6, %6\n"                    \
               "v_fma_f32 %3, %3, %6, %6\n v_fma_f32 %4, %4, %6, %6\n v_fma_f32 %5, %5, %6, %6"                      \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(c))

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint4* src, float* out, int iters) {
  union U { uint4 u; bf16x8 v; };
  U a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i].u = src[(threadIdx.x * 4 + i) & 4095]; b[i].u = src[(threadIdx.x * 4 + i + 77) & 4095]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + .1f, x2 = x0 + .2f, x3 = x0 + .3f, x4 = x0 + .4f, x5 = x0 + .5f, x6 = x0 + .6f, x7 = x0 + .7f;
  const float c = 0.999f;
  auto M1 = [&](int i, int j) { acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3].v, b[(i + j) & 3].v, acc[i], 0, 0, 0); };
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      // K Q^T: two chains of three; V^T P: two chains of four
      M1(0, 0); M1(1, 0); M1(0, 1); M1(1, 1); M1(0, 2); M1(1, 2);
      M1(2, 0); M1(3, 0); M1(2, 1); M1(3, 1); M1(2, 2); M1(3, 2); M1(2, 3); M1(3, 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 || MODE == 2) {
      EXP8(x0, x1, x2, x3, x4, x5, x6, x7); EXP8(x0, x1, x2, x3, x4, x5, x6, x7);
      EXP8(x0, x1, x2, x3, x4, x5, x6, x7); EXP8(x0, x1, x2, x3, x4, x5, x6, x7);
      CVT4(x0, x1, x2, x3, x4, x5, x6, x7); CVT4(x0, x1, x2, x3, x4, x5, x6, x7);
      CVT4(x0, x1, x2, x3, x4, x5, x6, x7); CVT4(x0, x1, x2, x3, x4, x5, x6, x7);
      FMA6(x0, x1, x2, x3, x4, x5, c); FMA6(x2, x3, x4, x5, x6, x7, c); FMA6(x0, x1, x2, x3, x4, x5, c); FMA6(x2, x3, x4, x5, x6, x7, c);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 3) {   // 14 groups: one MFMA + (32 + 16 + 24) / 14 ~ 5 VALU each
#define GRP(i, j, BODY) M1(i, j); __builtin_amdgcn_sched_barrier(0); BODY; __builtin_amdgcn_sched_barrier(0)
      GRP(0, 0, EXP8(x0, x1, x2, x3, x4, x5, x6, x7)); GRP(1, 0, CVT4(x0, x1, x2, x3, x4, x5, x6, x7));
      GRP(0, 1, EXP8(x0, x1, x2, x3, x4, x5, x6, x7)); GRP(1, 1, FMA6(x0, x1, x2, x3, x4, x5, c));
      GRP(0, 2, EXP8(x0, x1, x2, x3, x4, x5, x6, x7)); GRP(1, 2, CVT4(x0, x1, x2, x3, x4, x5, x6, x7));
      GRP(2, 0, EXP8(x0, x1, x2, x3, x4, x5, x6, x7)); GRP(3, 0, FMA6(x2, x3, x4, x5, x6, x7, c));
      GRP(2, 1, CVT4(x0, x1, x2, x3, x4, x5, x6, x7)); GRP(3, 1, FMA6(x0, x1, x2, x3, x4, x5, c));
      GRP(2, 2, CVT4(x0, x1, x2, x3, x4, x5, x6, x7)); GRP(3, 2, FMA6(x2, x3, x4, x5, x6, x7, c));
      GRP(2, 3, (void)0); GRP(3, 3, (void)0);
    }
  }
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[(blockIdx.x * 256 + threadIdx.x) & 0xffff] = s;
}

template <int MODE, int WPS>
static double run(const uint4* d, float* o, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(256 * WPS), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best * 1e6 / iters;   // ns per loop trip (every SIMD runs WPS waves, each doing one tile per trip)
}
int main() {
  std::vector<unsigned> h(4096 * 4);
  for (auto& x : h) {
    unsigned lo = (rand() & 0x7f) | (0x3f00 - ((rand() & 3) << 7)) | ((rand() & 1) << 15);
    unsigned hi = (rand() & 0x7f) | (0x3f00 - ((rand() & 3) << 7)) | ((rand() & 1) << 15);
    x = lo | (hi << 16);
  }
  uint4* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&o, 65536 * 4);
  const int iters = 20000;
  printf("ns per loop trip (one tile per resident wave; a SIMD holds WPS waves): 14 MFMA 32x32x16 | 32 exp + 16 cvt_pk + 24 fma\n");
  printf("waves/SIMD  MFMA only   VALU only   M then V    interleaved  | per tile: M / V / M-then-V / interleaved (ns per tile per SIMD = trip / WPS)\n");
#define ROW(W)                                                                                                        \
  {                                                                                                                   \
    const double m = run<0, W>(d, o, iters), v = run<1, W>(d, o, iters), s = run<2, W>(d, o, iters), i = run<3, W>(d, o, iters); \
    printf("%5d     %9.1f   %9.1f   %9.1f   %9.1f    | %7.1f %7.1f %7.1f %7.1f   (M + V = %.1f, max = %.1f)\n", W, m, v, s, i, m / W, v / W,  \
           s / W, i / W, (m + v) / W, (m > v ? m : v) / W);                                                              \
  }
  ROW(1) ROW(2) ROW(3)
  return 0;
}
