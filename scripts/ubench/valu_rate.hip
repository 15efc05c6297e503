// VALU issue-rate microbenchmark (dev tool): cycles per wave64 instruction for a few opcodes, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
template <int OP>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float c = out[0];
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) {  // v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                     "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (OP == 1) {  // v_exp_f32
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 2) {  // v_pk_fma_f16
        asm volatile("v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n"
                     "v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (OP == 3) {  // v_pk_fma_f32 (two registers per operand)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                     "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&a0));
      } else if (OP == 4) {  // v_cvt_pk_bf16_f32
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                     "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 5) {  // v_max3_f32
        asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n"
                     "v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (OP == 6) {  // v_exp_f16
        asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
                     "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == 7) {  // v_pk_add_f16 / integer pk
        asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                     "v_pk_lshlrev_b16 %4, 10, %4\n v_pk_lshlrev_b16 %5, 10, %5\n v_pk_lshlrev_b16 %6, 10, %6\n v_pk_lshlrev_b16 %7, 10, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0) / (float)(iters * REP);
  out[2 + (threadIdx.x & 1)] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP>
void run(const char* name, float* d) {
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * waves), 0, 0, d, 200);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * waves), 0, 0, d, 200);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    // wall-clock per instruction per SIMD: ms / (iters*REP*waves)
    printf("%-18s waves/SIMD=%d  s_memtime cyc/instr(one wave)=%.2f  wall ns/instr/SIMD=%.3f\n", name, waves, h[1], ms * 1e6 / (200.0 * REP * waves));
  }
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  run<0>("v_fma_f32", d); run<1>("v_exp_f32", d); run<2>("v_pk_fma_f16", d); run<3>("v_pk_fma_f32", d);
  run<4>("v_cvt_pk_bf16_f32", d); run<5>("v_max3_f32", d); run<6>("v_exp_f16", d); run<7>("v_pk_add/lshl_u16", d);
  return 0;
}
