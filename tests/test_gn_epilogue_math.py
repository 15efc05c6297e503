"""CPU: the lane bookkeeping of the GroupNorm-statistics epilogue (csrc/gemm_tile.h, `GN` path of tile_epilogue) restated in numpy
and checked against direct sums: which (row, 8- or 4-column chunk) a lane (g, j) of a wave owns after the fragment-pair swap, how
a chunk's columns split over two channel groups, the 16-lane butterfly over j, and which accumulator the flushing lane adds to --
for the wave tiles of the three kernels that carry the epilogue (256x320: 10 fragments per wave; 256x160: 5, i.e. two pairs and
the 4-column tail; 256x128: 4) and the group widths of the SD / SDXL UNets.  It restates the index math (the kernel itself runs
in tests/test_zz_dit_gpu.py on the GPU): a change to one must be mirrored in the other."""
import numpy as np
import pytest

rng = np.random.default_rng(0)


def emulate(NF, MF, BNw, cpg, G, rows_per_sample, M=512, waves_n=2):
    # one 256-row x (2*BNw) tile region starting at m0=256 (second M tile), n0=0; wave tile 64 x BNw
    N = cpg * G
    Y = rng.standard_normal((M, N)).astype(np.float32)
    stats = np.zeros((M // rows_per_sample, G, 2), np.float64)
    for m0 in range(0, M, 256):
      for n0 in range(0, N, waves_n * BNw):
        for wm in range(4):
          for wn in range(waves_n):
            mw, nw = m0 + wm * 64, n0 + wn * BNw
            # pairs
            for pr in range(NF // 2):
                nf = 2 * pr
                lane_s = np.zeros((64, 4))
                lane_n = np.zeros(64, int)
                for lane in range(64):
                    g, j = lane >> 4, lane & 15
                    n = nw + (nf + (g & 1)) * 16 + (g >> 1) * 8
                    lane_n[lane] = n
                    split = (n // cpg + 1) * cpg - n
                    for mf in range(MF):
                        m = mw + mf * 16 + j
                        v = Y[m, n:n + 8]
                        for e in range(8):
                            if e < split: lane_s[lane, 0] += v[e]; lane_s[lane, 1] += v[e] ** 2
                            else: lane_s[lane, 2] += v[e]; lane_s[lane, 3] += v[e] ** 2
                # butterfly over j within each 16-lane group
                for o in (1, 2, 4, 8):
                    lane_s = lane_s + lane_s[np.arange(64) ^ o]
                for lane in range(64):
                    g, j = lane >> 4, lane & 15
                    if j == 0:
                        n = lane_n[lane]; gA = n // cpg; b = mw // rows_per_sample
                        stats[b, gA] += lane_s[lane, :2]
                        if (n + 7) // cpg != gA: stats[b, gA + 1] += lane_s[lane, 2:]
            if NF & 1:
                lane_s = np.zeros((64, 4)); lane_n = np.zeros(64, int)
                for lane in range(64):
                    g, j = lane >> 4, lane & 15
                    n = nw + (NF - 1) * 16 + g * 4
                    lane_n[lane] = n
                    split = (n // cpg + 1) * cpg - n
                    for mf in range(MF):
                        m = mw + mf * 16 + j
                        v = Y[m, n:n + 4]
                        for e in range(4):
                            if e < split: lane_s[lane, 0] += v[e]; lane_s[lane, 1] += v[e] ** 2
                            else: lane_s[lane, 2] += v[e]; lane_s[lane, 3] += v[e] ** 2
                for o in (1, 2, 4, 8):
                    lane_s = lane_s + lane_s[np.arange(64) ^ o]
                for lane in range(64):
                    g, j = lane >> 4, lane & 15
                    if j == 0:
                        n = lane_n[lane]; gA = n // cpg; b = mw // rows_per_sample
                        stats[b, gA] += lane_s[lane, :2]
                        if (n + 3) // cpg != gA: stats[b, gA + 1] += lane_s[lane, 2:]
    Yr = Y.reshape(M // rows_per_sample, rows_per_sample, G, cpg).astype(np.float64)
    ref = np.stack([Yr.sum((1, 3)), (Yr ** 2).sum((1, 3))], -1)
    return np.abs(stats - ref).max() / np.abs(ref).max()


@pytest.mark.parametrize("NF,BNw,cpg", [(10, 160, 10), (10, 160, 20), (10, 160, 40), (5, 80, 10), (5, 80, 15), (4, 64, 8), (4, 64, 20)])
def test_epilogue_lane_sums_equal_direct_group_sums(NF, BNw, cpg):
    assert emulate(NF, 4, BNw, cpg, 32, 256) < 1e-7
