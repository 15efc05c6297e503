// MFMA throughput under full-chip load (dev tool): 16x16x32 vs 32x32x16 bf16, 2 waves per SIMD, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512) void k(const uint4* src, float* out, int iters) {
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
  } else {
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
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  std::vector<unsigned> h(4096 * 4);
  for (auto& x : h) {  // random bf16 pairs in [-1, 1)
    unsigned lo = (rand() & 0x7f) | (0x3f00 - ((rand() & 3) << 7)) | ((rand() & 1) << 15);
    unsigned hi = (rand() & 0x7f) | (0x3f00 - ((rand() & 3) << 7)) | ((rand() & 1) << 15);
    x = lo | (hi << 16);
  }
  uint4* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&o, 256 * 512 * 4);
  const int iters = 4000;
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      if (kind == 0) hipLaunchKernelGGL((k<0>), dim3(256), dim3(512), 0, 0, d, o, iters);
      else hipLaunchKernelGGL((k<1>), dim3(256), dim3(512), 0, 0, d, o, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      // flops: kind0: 80 MFMA x 16384 per iter per wave; kind1: 40 MFMA x 32768
      double fl = (double)iters * 80 * 16384.0 * 8 * 256;
      printf("%s rep %d: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "16x16x32" : "32x32x16", rep, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
