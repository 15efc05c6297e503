// MFMA throughput under full-chip load (dev tool): 16x16x32 vs 32x32x16 bf16, 2 waves per SIMD (512 threads) and 1 wave per SIMD
// (256 threads, 16 independent 32x32 accumulators), on RANDOM and on ZERO operands.  The guide's 2495 TFLOP/s figure
// (MI355X_MICROARCH.md) is a micro-benchmark ceiling; its "DVFS give-back" note says the same binary runs 2.30 GHz on zero-filled and
// 1.90 - 1.95 GHz on random operands.  This tool prints both fills so the bench line's `sustained_mfma_pflops` can say which is which
// (run it under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` for the clock: GRBM_GUI_ACTIVE / duration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(KIND == 2 ? 256 : 512) void k(const uint4* src, float* out, int iters) {
  union U { uint4 u; bf16x8 v; };
  U a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i].u = src[(threadIdx.x * 8 + i) & 4095]; b[i].u = src[(threadIdx.x * 8 + i + 77) & 4095]; }
  float s = 0.f;
  if (KIND == 0) {
    f32x4 acc[40];
    for (int i = 0; i < 40; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 40; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7].v, b[(i >> 1) & 7].v, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 40; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i & 7].v, a[(i >> 1) & 7].v, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 40; ++i) s += acc[i][0] + acc[i][3];
  } else if (KIND == 1) {
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 7].v, b[(i >> 1) & 7].v, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][15];
  } else {   // one wave per SIMD, 16 independent 32x32 accumulators (256 registers), 80 MFMAs per iteration like KIND 1's two waves
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 7].v, b[(i >> 1) & 7].v, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  std::vector<unsigned> h(4096 * 4), z(4096 * 4, 0u);
  for (auto& x : h) {  // random bf16 pairs in [-1, 1)
    unsigned lo = (rand() & 0x7f) | (0x3f00 - ((rand() & 3) << 7)) | ((rand() & 1) << 15);
    unsigned hi = (rand() & 0x7f) | (0x3f00 - ((rand() & 3) << 7)) | ((rand() & 1) << 15);
    x = lo | (hi << 16);
  }
  uint4* d; float* o;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&o, 256 * 512 * 4);
  const int iters = 4000;
  const char* names[3] = {"16x16x32 2 waves/SIMD", "32x32x16 2 waves/SIMD", "32x32x16 1 wave/SIMD "};
  for (int fill = 0; fill < 2; ++fill) {
    hipMemcpy(d, fill == 0 ? h.data() : z.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int kind = 0; kind < 3; ++kind) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (kind == 0) hipLaunchKernelGGL((k<0>), dim3(256), dim3(512), 0, 0, d, o, iters);
        else if (kind == 1) hipLaunchKernelGGL((k<1>), dim3(256), dim3(512), 0, 0, d, o, iters);
        else hipLaunchKernelGGL((k<2>), dim3(256), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // flops per iteration per CU: kind 0: 80 MFMA x 16384 x 8 waves; kind 1: 40 x 32768 x 8; kind 2: 80 x 32768 x 4
        double fl = (double)iters * 80 * 16384.0 * 8 * 256;
        printf("%s %s rep %d: %.3f ms  %.1f TFLOP/s\n", names[kind], fill == 0 ? "random" : "zeros ", rep, ms, fl / ms / 1e9);
      }
    }
  }
  return 0;
}
