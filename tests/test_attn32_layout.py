"""CPU restatement of the index bookkeeping the 32x32x16 attention kernels rest on (flash_diffusion_amd/csrc/attn.hip:
attn_fwd32_kernel, attn_bwd_dq32_kernel, attn_bwd_dkv32_kernel).  No GPU: numpy / integer arithmetic only.  What is pinned:

* the row permutation of the staged K (V, Q, dO) tiles: C-layout rows of `v_mfma_f32_32x32x16_bf16` map to CONSECUTIVE keys of the
  transposed operand's tile, so that one lane half's 8 contraction slots of a k-step are one 16-byte LDS read;
* the LDS images (128-byte rows with the (row >> 1) & 7 chunk swizzle; the 32-byte rows of the second K sub-tile): every fragment
  read of a wave is bank-conflict free under the ds_read_b128 lane grouping of /opt/skills/guides/MI355X_MICROARCH.md (LDS table);
* where the ones row / the running-maximum column of the forward live (O^T row d, K' column d) in register / lane terms;
* the shared-memory budget per block of every instantiation (two blocks per CU must fit 160 KiB).
"""
import itertools

import numpy as np


def keyperm(r):
    """attn32_keyperm: swap bits 2 and 3"""
    return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)


def c_row(reg, h):
    """row of accumulator register `reg` (0..15) in lane half h of a 32x32 MFMA result: (reg & 3) + 8 (reg >> 2) + 4 h"""
    return (reg & 3) + 8 * (reg >> 2) + 4 * h


def test_keyperm_is_an_involution_inside_groups_of_16():
    for r in range(64):
        assert keyperm(keyperm(r)) == r and keyperm(r) // 16 == r // 16


def test_c_layout_rows_are_consecutive_keys_of_the_transposed_tile():
    """k-step st = 2 kb + hs of V^T P (K^T dS, dO^T P, Q^T dS): lane half h packs registers 8 hs .. 8 hs + 7 of block kb as its B-operand
    slots e = 0..7 (contraction index 8 h + e); the staged row those registers belong to holds key keyperm(row): it must be key
    16 st + 8 h + e, the e-th of 8 consecutive columns of the transposed operand's tile."""
    for kb, hs, h, e in itertools.product(range(2), range(2), range(2), range(8)):
        reg = 8 * hs + e
        row = 32 * kb + c_row(reg, h)
        assert keyperm(row) == 16 * (2 * kb + hs) + 8 * h + e


# ds_read_b128: a wave's 64 lanes are served in four groups of 16 (LDS table of the guide); conflict-free = 16 distinct 16-byte slots
# of the 256-byte bank row inside a group
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS = B128_GROUPS + [[x + 32 for x in g] for g in B128_GROUPS]


def conflict_free(addr_of_lane):
    for g in B128_GROUPS:
        slots = {(addr_of_lane(l) // 16) % 16 for l in g}
        if len(slots) != 16:
            return False
    return True


def test_fragment_reads_of_the_row_tiles_are_conflict_free():
    """K_A / V_A / Q_A / dO_A (rows of 128 B, logical chunk c at physical chunk c ^ ((row >> 1) & 7)): lane (h, n) reads row 32 kb + n,
    chunk 2 ks + h -- and the transposed tiles (V^T, K^T, Q^T, dO^T) use the same image with st in place of ks."""
    for kb, ks in itertools.product(range(3), range(4)):
        def addr(l, kb=kb, ks=ks):
            h, n = l >> 5, l & 31
            row = 32 * kb + n
            return row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) * 16)
        assert conflict_free(addr), (kb, ks)


def test_second_k_subtile_reads_cost_at_most_two_cycles_per_group():
    """K_B (columns 64..79 of an 80-wide head: 32-byte rows, key rho at rho * 32 + h * 16): 2-way conflicts at worst (one k-step of five)"""
    for kb in range(2):
        worst = 0
        for g in B128_GROUPS:
            slots = [(((32 * kb + (l & 31)) * 32 + (l >> 5) * 16) // 16) % 16 for l in g]
            worst = max(worst, max(slots.count(s) for s in set(slots)))
        assert worst <= 2


def test_dma_image_matches_the_fragment_addressing():
    """LDS-DMA writes lane-linear: wave w, piece i, lane l lands at (8 (w + 4 i) + l / 8) * 128 + (l % 8) * 16 and must carry logical chunk
    (l % 8) ^ swizzle(row) of the row holding key keyperm(row)"""
    image = {}
    for w, i, l in itertools.product(range(4), range(2), range(64)):
        row, pc = 8 * (w + 4 * i) + (l >> 3), l & 7
        c = pc ^ ((row >> 1) & 7)
        image[row * 128 + pc * 16] = (keyperm(row), c)
    for kb, ks, l in itertools.product(range(2), range(4), range(64)):
        h, n = l >> 5, l & 31
        row = 32 * kb + n
        key, c = image[row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) * 16)]
        assert key == keyperm(row) and c == 2 * ks + h          # columns 16 ks + 8 h .. + 7 of that key


def test_ones_row_and_maximum_column_positions():
    for d in (8, 16, 24, 32, 40, 56, 72):
        # denominator = O^T row d: block d / 32, register 4 ((d % 32) / 8) + (d % 4), lane half ((d % 8) / 4)
        db, dl = d >> 5, d & 31
        reg, hh = 4 * (dl >> 3) + (dl & 3), (dl & 7) >> 2
        assert 32 * db + c_row(reg, hh) == d
        # running maximum = column d of K' / Q': k-step d / 16, lane half (d / 8) % 2, element 0 of that fragment
        ks, h = d >> 4, (d >> 3) & 1
        assert 16 * ks + 8 * h == d


def test_shared_memory_budgets_allow_two_blocks_per_cu():
    lds = 160 * 1024
    fwd = {(ks, db): 2 * (8192 + db * 4096 + (2048 if ks > 4 else 0)) for ks, db in ((3, 2), (4, 2), (5, 3))}
    dq = {(ks, db): 2 * (16384 + db * 4096 + (4096 if ks > 4 else 0)) for ks, db in ((3, 2), (4, 2), (5, 3))}
    dkv = {(ks, db, ns): ns * (16384 + 2 * db * 4096 + (4096 if ks > 4 else 0) + 1024) for ks, db, ns in ((3, 2, 2), (4, 2, 2), (5, 3, 1))}
    for table in (fwd, dq, dkv):
        for key, bytes_ in table.items():
            assert 2 * bytes_ <= lds, (key, bytes_)
    assert fwd[(3, 2)] * 3 <= lds        # the d <= 40 forward runs three blocks (waves per SIMD) per CU
    assert dkv[(5, 3, 1)] * 2 <= lds and 2 * (2 * dkv[(5, 3, 1)]) > lds      # why the 80-wide dK / dV kernel is single-staged
