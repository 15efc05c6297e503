// HBM streaming rate by ACCESS PATTERN on a row-major [M][320] bf16 matrix (640-byte rows), dev tool (round 4).
// Why: every tile geometry of the K = N = 320 row GEMMs stops at ~3.3 TB/s, and with the lean epilogue (40 instead of ~250
// instructions per 16-byte store) the epilogue of a 256 x 320 tile still takes as long -- is it the pattern?  Each thread moves
// 20 x 16 bytes per 256-row tile in all patterns:
//   0  the GEMM epilogue's: a wave instruction covers 16 rows x 64 bytes (4 lanes per row), then the next row fragment, then the
//      next 64-byte column chunk (8 waves as 4 x 2 over the 256 x 320 tile)
//   1  whole 128-byte lines: a wave instruction covers 8 rows x 128 bytes, the block walks the five lines of its 256 rows
//   2  contiguous: the tile is one 160 KB run, a wave instruction covers 1 KB of it
// modes: read (xor-reduced), write, copy (read one matrix, write another with the same pattern); 1 / 2 / 4 blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROWB = 640, TROWS = 256, NACC = 20;

template <int PAT>
__device__ __forceinline__ size_t off_of(int tile, int i, int tid) {
  const size_t base = (size_t)tile * TROWS * ROWB;
  if (PAT == 0) {
    const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, g = lane >> 4, j = lane & 15;
    const int pr = i >> 2, mf = i & 3;
    const int row = wm * 64 + mf * 16 + j;
    const int colb = wn * 320 + ((2 * pr + (g & 1)) * 16 + (g >> 1) * 8) * 2;
    return base + (size_t)row * ROWB + colb;
  } else if (PAT == 1) {
    const int line = i / 4, rg = i & 3;          // 5 lines x 4 row groups of 64
    const int row = rg * 64 + (tid >> 3);
    return base + (size_t)row * ROWB + line * 128 + (tid & 7) * 16;
  } else {
    return base + ((size_t)i * 512 + tid) * 16;
  }
}

template <int PAT, int MODE>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ src, char* __restrict__ dst, int ntiles, unsigned* sink) {
  const int tid = threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    uint4 v[NACC];
    if (MODE != 1) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) v[i] = *(const uint4*)(src + off_of<PAT>(t, i, tid));
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    } else {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        uint4 w = MODE == 1 ? make_uint4(t, i, tid, 7) : v[i];
        *(uint4*)(dst + off_of<PAT>(t, i, tid)) = w;
      }
    }
  }
  if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int PAT, int MODE>
float run(std::vector<char*>& bufs, int ntiles, int occ, unsigned* sink, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int nb = (int)bufs.size();
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL((stream_kernel<PAT, MODE>), dim3(256 * occ), dim3(512), 0, 0, bufs[r % nb], bufs[(r + nb / 2) % nb], ntiles, sink);
  hipEventRecord(a, 0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((stream_kernel<PAT, MODE>), dim3(256 * occ), dim3(512), 0, 0, bufs[r % nb], bufs[(r + nb / 2) % nb], ntiles, sink);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;   // us per launch
}

int main() {
  const int M = 131072, ntiles = M / TROWS;
  const size_t bytes = (size_t)M * ROWB;
  std::vector<char*> bufs(8);
  for (auto& p : bufs) { hipMalloc(&p, bytes); hipMemset(p, 1, bytes); }
  unsigned* sink; hipMalloc(&sink, 4);
  hipDeviceSynchronize();
  const char* pn[3] = {"epilogue 16 rows x 64 B", "lines 8 rows x 128 B", "contiguous 1 KB"};
  const char* mn[3] = {"read", "write", "copy"};
  for (int occ : {1, 2, 4}) {
    for (int mode = 0; mode < 3; ++mode) {
      float us[3];
      us[0] = mode == 0 ? run<0, 0>(bufs, ntiles, occ, sink, 24) : mode == 1 ? run<0, 1>(bufs, ntiles, occ, sink, 24) : run<0, 2>(bufs, ntiles, occ, sink, 24);
      us[1] = mode == 0 ? run<1, 0>(bufs, ntiles, occ, sink, 24) : mode == 1 ? run<1, 1>(bufs, ntiles, occ, sink, 24) : run<1, 2>(bufs, ntiles, occ, sink, 24);
      us[2] = mode == 0 ? run<2, 0>(bufs, ntiles, occ, sink, 24) : mode == 1 ? run<2, 1>(bufs, ntiles, occ, sink, 24) : run<2, 2>(bufs, ntiles, occ, sink, 24);
      const double moved = bytes * (mode == 2 ? 2.0 : 1.0);
      printf("%d block(s)/CU %-5s |", occ, mn[mode]);
      for (int p = 0; p < 3; ++p) printf(" %s: %6.1f us %5.2f TB/s |", pn[p], us[p], moved / us[p] / 1e6);
      printf("\n");
    }
  }
  return 0;
}
