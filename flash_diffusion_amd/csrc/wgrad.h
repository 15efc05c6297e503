// Grouped LoRA weight gradients (wgrad.hip): several independent products C_i[n1][n2] += sum_m X_i[m][n1] * Y_i[m][n2] in ONE launch.
// (Declared here and not in ops.h: ops.h is part of the attention kernels' source set, whose counter files are keyed by its hash.)
#pragma once
#include "common.h"

struct WgradProblem {
  const bf16_t* X; int64_t ldx;    // [M][ldx], N1 columns used
  const bf16_t* Y; int64_t ldy;    // [M][ldy], N2 columns used
  int64_t M; int N1, N2;
  float* C; int64_t ldc;           // [N1][ldc] fp32, accumulated with atomics
};
constexpr int WGRAD_GROUP_MAX = 6;
// n <= WGRAD_GROUP_MAX problems with the same constraints as launch_wgrad_tn.  One launch when the streaming kernel runs them with
// the same narrow-operand tile (they do for the LoRA gradients of one linear / one fused q/k/v: the narrow operand is the rank);
// otherwise -- or with developer switch 47 = 1 -- one launch_wgrad_tn per problem.
int launch_wgrad_tn_group(const WgradProblem* pr, int n, hipStream_t st);
