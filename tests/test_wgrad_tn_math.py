"""CPU: the index bookkeeping of the TN weight-gradient kernel (csrc/wgrad.hip) restated in numpy on top of the MFMA operand
convention the validated NT kernels use (gemm.hip: a fragment = 8 consecutive k of index lane & 15, k-block lane >> 4; lane (g, j)
of the 16x16 result holds first-operand indices g*4 .. g*4+3 and second-operand index j): the transposing scatter of a 64-row
slab (8 x 8 blocks, one per thread, transposed in registers: round 4) into the swizzled [index][64 k] LDS image, the fragment reads at that image's addresses, the wave / fragment -> (n1, n2)
mapping of the epilogue, zero fill past M / N1 / N2 and the row splits -- against X^T Y.  It restates the index math (the kernel
itself runs in tests/test_zz_dit_gpu.py on the GPU): a change to one must be mirrored in the other."""
import numpy as np
import pytest

BN1, BN2, BK = 64, 128, 64


def img_off(row, k):
    return ((row * 8 + ((k >> 3) ^ ((row >> 1) & 7))) << 4) + ((k & 7) << 1)


def emulate(M, N1, N2, rows_per_split, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.integers(-3, 4, (M, N1)).astype(np.float64)     # small integers: every partial sum is exact
    Y = rng.integers(-3, 4, (M, N2)).astype(np.float64)
    C = np.zeros((N1, N2))
    for bz in range((M + rows_per_split - 1) // rows_per_split):
        m_beg, m_end = bz * rows_per_split, min(M, (bz + 1) * rows_per_split)
        for by in range((N1 + BN1 - 1) // BN1):
            for bx in range((N2 + BN2 - 1) // BN2):
                n1_0, n2_0 = by * BN1, bx * BN2
                acc = np.zeros((4, 2, 4, 64, 4))                       # [wave][f1][f2][lane][r]
                for kt in range((m_end - m_beg + BK - 1) // BK):
                    m0 = m_beg + kt * BK
                    sx, sy = np.zeros(BN1 * 64), np.zeros(BN2 * 64)    # images indexed by byte offset / 2
                    for tid in range(256):                               # staging: one 8 (rows) x 8 (columns) block per thread
                        sw, lane = tid >> 6, tid & 63
                        if sw == 3:
                            continue                                     # (the fourth wave only multiplies)
                        br, bc = lane & 7, (8 if sw == 2 else 0) + (lane >> 3)
                        img, src, n0, N = (sx, X, n1_0, N1) if sw == 0 else (sy, Y, n2_0, N2)
                        sn0 = n0 + bc * 8
                        rr = np.zeros((8, 8))                            # the block's 8 rows as loaded (zero past M / N)
                        for i in range(8):
                            m = m0 + br * 8 + i
                            if sn0 < N and m < m_end:
                                rr[i] = src[m, sn0:sn0 + 8]
                        for e in range(8):                               # register transpose: column e -> one 16-byte chunk
                            n = bc * 8 + e
                            base = ((n * 8 + (br ^ ((n >> 1) & 7))) << 4) // 2
                            assert base == img_off(n, br * 8) // 2
                            for q in range(4):                           # dword q = (row 2q | row 2q + 1 << 16)
                                img[base + 2 * q], img[base + 2 * q + 1] = rr[2 * q, e], rr[2 * q + 1, e]
                    for wave in range(4):
                        w1, w2 = wave >> 1, wave & 1
                        for ks in range(2):
                            xf = np.zeros((2, 64, 8))
                            yf = np.zeros((4, 64, 8))
                            for lane in range(64):
                                g, j = lane >> 4, lane & 15
                                pc = (ks * 4 + g) ^ (j >> 1)
                                for f in range(2):
                                    base = (((w1 * 32 + f * 16 + j) * 8 + pc) * 16) // 2
                                    xf[f, lane] = sx[base:base + 8]
                                for f in range(4):
                                    base = (((w2 * 64 + f * 16 + j) * 8 + pc) * 16) // 2
                                    yf[f, lane] = sy[base:base + 8]
                            for f1 in range(2):
                                for f2 in range(4):
                                    A = np.zeros((16, 32))
                                    Bm = np.zeros((32, 16))
                                    for lane in range(64):              # the MFMA operand convention
                                        A[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = xf[f1, lane]
                                        Bm[(lane >> 4) * 8:(lane >> 4) * 8 + 8, lane & 15] = yf[f2, lane]
                                    D = A @ Bm
                                    for lane in range(64):
                                        for r in range(4):
                                            acc[wave, f1, f2, lane, r] += D[(lane >> 4) * 4 + r, lane & 15]
                for wave in range(4):
                    w1, w2 = wave >> 1, wave & 1
                    for f1 in range(2):
                        for f2 in range(4):
                            for lane in range(64):
                                g, j = lane >> 4, lane & 15
                                n2 = n2_0 + w2 * 64 + f2 * 16 + j
                                for r in range(4):
                                    n1 = n1_0 + w1 * 32 + f1 * 16 + g * 4 + r
                                    if n1 < N1 and n2 < N2:
                                        C[n1, n2] += acc[wave, f1, f2, lane, r]
    return np.abs(C - X.T @ Y).max()


@pytest.mark.parametrize("M,N1,N2,rps", [(128, 64, 128, 64), (200, 64, 128, 128), (77, 128, 320, 64), (130, 72, 40, 192),
                                         (200, 192, 64, 128), (100, 136, 8, 64)])   # (the rank-64 / rank-8 dB shapes: a half-empty second tile)
def test_tn_kernel_bookkeeping_equals_xt_y(M, N1, N2, rps):
    assert emulate(M, N1, N2, rps) == 0.0


# ---- round 5: the streaming kernel (wgrad_tn2_kernel): LDS-DMA of the row-major slabs with a source-side granule swizzle, fragments
# by the transposing LDS read `ds_read_b64_tr_b16`, reduction-index assignment row = 32 ks + 16 h + 4 g + j ------------------------
def tr_read(img, addrs):
    """ds_read_b64_tr_b16 of ONE 16-lane group: lane i supplies the byte address of 4 contiguous 16-bit elements = piece i (row
    i >> 2, quarter i & 3) of a [4 rows][16 columns] block; lane c receives column c of the block, elements j = rows 0 .. 3
    (cdna_hip_programming.md: `lane l, elem j reads lds[(l & 15) + j * 16 + (l >> 4) * 64]` in the dense layout)."""
    block = np.zeros((4, 16))
    for i, a in enumerate(addrs):
        assert a % 8 == 0
        block[i >> 2, (i & 3) * 4:(i & 3) * 4 + 4] = img[a // 2:a // 2 + 4]
    return block.T.copy()          # [lane c][j]


def bank_groups(addrs):
    """the 8-bank groups (32 B of the 256-B bank row) a set of 8-byte accesses touches"""
    return [(a % 256) // 32 for a in addrs]


def emulate2(M, N1, N2, rows_per_split, BN1, ct, seed=0, check_banks=False):
    BN2, BK, NST = 128, 64, 4
    RB1, RB2 = BN1 * 2, BN2 * 2
    XB, YB = BK * RB1, BK * RB2
    W1 = BN1 // 64
    W2 = 4 // W1
    F1, F2 = 4, BN2 // W2 // 16
    rng = np.random.default_rng(seed)
    X = rng.integers(-3, 4, (M, N1)).astype(np.float64)
    Y = rng.integers(-3, 4, (M, N2)).astype(np.float64)
    C = np.zeros((N2, N1)) if ct else np.zeros((N1, N2))

    def src_chunk(r, slot, RB):
        f = (r & 7) if RB == 256 else ((r >> 1) & 3)
        return (((slot >> 1) ^ f) << 1) | (slot & 1)
    for bz in range((M + rows_per_split - 1) // rows_per_split):
        m_beg, m_end = bz * rows_per_split, min(M, (bz + 1) * rows_per_split)
        nk = (m_end - m_beg + BK - 1) // BK
        for by in range((N1 + BN1 - 1) // BN1):
            for bx in range((N2 + BN2 - 1) // BN2):
                n1_0, n2_0 = by * BN1, bx * BN2
                acc = np.zeros((4, F1, F2, 64, 4))
                for kt in range(nk):
                    img = np.full((XB + YB) // 2, np.nan)                       # one ring stage (every byte must be written)
                    m0 = m_beg + kt * BK
                    for wave in range(4):                                        # DMA: piece = 1 KB lane-linear at wave * 1024 + 4096 i
                        for (P, RB, off, src, n0, N) in ((XB // 4096, RB1, 0, X, n1_0, N1), (YB // 4096, RB2, XB, Y, n2_0, N2)):
                            S = RB // 16
                            RPP = 64 // S
                            for i in range(P):
                                for lane in range(64):
                                    r = (wave + 4 * i) * RPP + lane // S
                                    col = n0 + src_chunk(r, lane % S, RB) * 8
                                    dst = off + wave * 1024 + 4096 * i + lane * 16
                                    assert dst == off + r * RB + (lane % S) * 16   # the lane-linear piece IS row r, slot lane % S
                                    m = m0 + r
                                    img[dst // 2:dst // 2 + 8] = src[m, col:col + 8] if (m < m_end and col < N) else 0.0
                    assert not np.isnan(img).any()
                    for wave in range(4):
                        w1, w2 = wave // W2, wave % W2
                        for ks in range(2):
                            def frag(off, RB, nf):
                                out = np.zeros((64, 8))
                                for h in range(2):
                                    allad = []
                                    for g in range(4):
                                        ad = []
                                        for c in range(16):
                                            br = 4 * g + (c >> 2)
                                            fl = (br & 7) if RB == 256 else ((br >> 1) & 3)
                                            ad.append(off + br * RB + (c & 3) * 8 + 32 * ks * RB + ((nf ^ fl) << 5) + h * 16 * RB)
                                        out[g * 16:(g + 1) * 16, 4 * h:4 * h + 4] = tr_read(img, ad)
                                        allad.append(ad)
                                    if check_banks:   # serviced in two 32-lane halves: each must spread over all eight 8-bank groups twice
                                        for half in (allad[0] + allad[1], allad[2] + allad[3]):
                                            assert sorted(bank_groups(half)) == sorted(list(range(8)) * 4), bank_groups(half)
                                return out
                            xf = [frag(0, RB1, w1 * F1 + f) for f in range(F1)]
                            yf = [frag(XB, RB2, w2 * F2 + f) for f in range(F2)]
                            for f1 in range(F1):
                                for f2 in range(F2):
                                    A, B = (yf[f2], xf[f1]) if ct else (xf[f1], yf[f2])
                                    # MFMA 16x16x32: D[i][j] += sum over lane groups g and slots e of A[lane (g, i)][e] * B[lane (g, j)][e]
                                    for g in range(4):
                                        for c in range(16):
                                            for r in range(4):
                                                i = g * 4 + r
                                                acc[wave, f1, f2, g * 16 + c, r] += sum(
                                                    float(A[gg * 16 + i] @ B[gg * 16 + c]) for gg in range(4))
                for wave in range(4):
                    w1, w2 = wave // W2, wave % W2
                    for f1 in range(F1):
                        for f2 in range(F2):
                            for lane in range(64):
                                g, c = lane >> 4, lane & 15
                                for r in range(4):
                                    n1 = n1_0 + (w1 * F1 + f1) * 16 + (c if ct else g * 4 + r)
                                    n2 = n2_0 + (w2 * F2 + f2) * 16 + (g * 4 + r if ct else c)
                                    if n1 < N1 and n2 < N2:
                                        if ct:
                                            C[n2, n1] += acc[wave, f1, f2, lane, r]
                                        else:
                                            C[n1, n2] += acc[wave, f1, f2, lane, r]
    ref = X.T @ Y
    return C, (ref.T if ct else ref)


@pytest.mark.parametrize("M,N1,N2,rps,BN1,ct", [(128, 128, 128, 128, 128, False), (100, 64, 72, 64, 64, False),
                                                (192, 128, 160, 128, 128, True), (64, 40, 128, 64, 64, True)])
def test_streaming_tn_kernel_bookkeeping(M, N1, N2, rps, BN1, ct):
    C, ref = emulate2(M, N1, N2, rps, BN1, ct, check_banks=(M == 128))
    assert np.array_equal(C, ref)
