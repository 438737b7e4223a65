"""numpy index model of the 32x32x16-MFMA variant of conv_halo_kernel (csrc/conv_halo.hip, template flag M32; experimental,
OCR_HALO_MFMA32=1).  It replays, lane by lane, what the kernel does with ADDRESSES — the LDS image the LDS-DMA builds (source
swizzle), the fragment reads (row / 16-byte position per lane, zero-row redirect), v_mfma_f32_32x32x16_bf16's operand and result
layout, the epilogue's (pixel, channel) mapping — and compares the result with a direct 3x3 SAME convolution.  Arithmetic is
exact (small integers in float64), so any index mistake shows up as a mismatch.  Also checks that the row swizzle
sw(r) = (r & 7) ^ ((r >> 3) & 1) makes every ds_read_b128 of a 32-row fragment bank-conflict free for every tap shift.

MFMA layout (cdna_hip_programming.md, Fragment layout; composable_kernel WarpGemmAttributeMfmaImplBf16Bf16F32M32N32K16):
  A: lane l holds A[row = l & 31][k = 8 * (l >> 5) + 0..7];  B: lane l holds B[k = 8 * (l >> 5) + 0..7][col = l & 31];
  D: lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
Run: python tools/halo_m32_model.py   (also imported by tests/test_halo_m32_model.py)
"""
import numpy as np

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def sw(r):
    """16-byte position swizzle of LDS row r (halo rows and weight rows alike)."""
    return (r & 7) ^ ((r >> 3) & 1)


def max_bank_conflict():
    """Worst number of lanes of one ds_read_b128 lane group that hit the same 4 banks, over all fragment base rows."""
    worst = 0
    for base in range(256):
        for half in (0, 1):
            for grp in B128_GROUPS:
                seen = {}
                for l in grp:
                    row = base + l
                    bank4 = ((row & 1) << 3) | ((0 ^ half ^ sw(row)) & 7)          # (row parity, position): 4 of 64 banks
                    seen[bank4] = seen.get(bank4, 0) + 1
                worst = max(worst, max(seen.values()))
    return worst


def mfma_32x32x16(afr, bfr, acc):
    """afr, bfr: [64 lanes][8]; acc: [64 lanes][16] — D += A * B in the hardware's lane layout."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = afr[l]
        Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = bfr[l]
    D = A @ Bm
    for l in range(64):
        for r in range(16):
            acc[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]


def run_tile(P, Q, M, N, C, cW, cH, m0, n0, BN, NW):
    """One workgroup of the M32 kernel: returns {(m, n): value} for the tile at (m0, n0).
    P [M][C] activations, Q [N][9*C] packed weights (tap-major), as conv_halo_kernel's HaloArgs."""
    H = cH
    BM = 32 * NW
    WAVES_N = BN // 64
    WAVES_M = NW // WAVES_N
    WM = BM // WAVES_M
    FM32 = WM // 32
    NR = BM + 2 * H + 2
    NRpad = (NR + 8 * NW) // (8 * NW) * (8 * NW)
    PI = NRpad // (8 * NW)
    QI = BN // (8 * NW)
    mfirst = m0 - H - 1
    out = {}
    nchunks = C // 64
    acc = {(w, a, b): np.zeros((64, 16)) for w in range(NW) for a in range(2) for b in range(FM32)}
    for chunk in range(nchunks):
        # ---- LDS image of the halo stage: position p of row r holds source chunk p ^ sw(r) (the DMA writes lane-linearly, so the
        #      swizzle is applied to the SOURCE chunk: lane -> row 8*g + (lane >> 3), position lane & 7)
        halo = np.zeros((NRpad, 8, 8))
        for wave in range(NW):
            for j in range(PI):
                g = wave * PI + j
                for lane in range(64):
                    rsub, pos = lane >> 3, lane & 7
                    r = g * 8 + rsub
                    csrc = pos ^ rsub ^ (g & 1)                  # == pos ^ sw(r)
                    assert csrc == pos ^ sw(r)
                    m = mfirst + r
                    if r < NR and 0 <= m < M:
                        halo[r, pos] = P[m, chunk * 64 + csrc * 8:chunk * 64 + csrc * 8 + 8]
        for tap in range(9):
            k0 = tap * C + chunk * 64
            wt = np.zeros((BN, 8, 8))
            for wave in range(NW):
                for j in range(QI):
                    g = wave * QI + j
                    for lane in range(64):
                        rsub, pos = lane >> 3, lane & 7
                        rl = g * 8 + rsub
                        csrc = pos ^ rsub ^ (g & 1)
                        n = n0 + rl
                        if n < N:
                            wt[rl, pos] = Q[n, k0 + csrc * 8:k0 + csrc * 8 + 8]
            shift = (H + 1) + (tap // 3 - 1) * H + (tap % 3 - 1)
            for wave in range(NW):
                wm, wn = wave // WAVES_N, wave % WAVES_N
                for kq in range(4):
                    afr = {a: np.zeros((64, 8)) for a in range(2)}
                    bfr = {b: np.zeros((64, 8)) for b in range(FM32)}
                    for lane in range(64):
                        r32, half = lane & 31, lane >> 5
                        swA = (lane & 7) ^ ((lane >> 3) & 1)
                        for a in range(2):
                            row = wn * 64 + a * 32 + r32
                            pos = ((half ^ swA) ^ (kq << 1)) & 7
                            assert pos == ((kq * 2 + half) ^ sw(row)) & 7
                            afr[a][lane] = wt[row, pos]
                        swB = ((r32 + shift) & 7) ^ (((r32 + shift) >> 3) & 1)
                        for b in range(FM32):
                            m = m0 + wm * WM + b * 32 + r32
                            valid = False
                            if m < M:
                                h, w = m % H, (m // H) % cW
                                ww, hh = w + tap // 3 - 1, h + tap % 3 - 1
                                valid = 0 <= ww < cW and 0 <= hh < H
                            row = wm * WM + b * 32 + r32 + shift
                            pos = ((half ^ swB) ^ (kq << 1)) & 7
                            assert pos == ((kq * 2 + half) ^ sw(row)) & 7
                            zrow = NR + ((r32 + shift) & 1)
                            assert zrow < NRpad and (zrow & 1) == (row & 1)            # same bank half as the real row
                            bfr[b][lane] = halo[row, pos] if valid else halo[zrow, pos]
                            if not valid:
                                assert not halo[zrow].any()
                    for a in range(2):
                        for b in range(FM32):
                            mfma_32x32x16(afr[a], bfr[b], acc[(wave, a, b)])
    for wave in range(NW):
        wm, wn = wave // WAVES_N, wave % WAVES_N
        for b in range(FM32):
            for a in range(2):
                for lane in range(64):
                    m = m0 + wm * WM + b * 32 + (lane & 31)
                    if m >= M:
                        continue
                    for j in range(4):
                        n = n0 + wn * 64 + a * 32 + j * 8 + (lane >> 5) * 4
                        for e in range(4):
                            if n + e < N:
                                out[(m, n + e)] = acc[(wave, a, b)][lane, 4 * j + e]
    return out


def direct_conv(P, Q, M, N, C, cW, cH):
    ref = np.zeros((M, N))
    for m in range(M):
        h, w, img = m % cH, (m // cH) % cW, m // (cH * cW)
        for tap in range(9):
            ww, hh = w + tap // 3 - 1, h + tap % 3 - 1
            if 0 <= ww < cW and 0 <= hh < cH:
                src = (img * cW + ww) * cH + hh
                ref[m] += Q[:, tap * C:(tap + 1) * C] @ P[src]
    return ref


def check(Nb=2, cW=20, cH=4, C=64, N=128, BN=128, NW=4, seed=0, tiles=None):
    rng = np.random.RandomState(seed)
    M = Nb * cW * cH
    P = rng.randint(-3, 4, size=(M, C)).astype(np.float64)
    Q = rng.randint(-3, 4, size=(N, 9 * C)).astype(np.float64)
    ref = direct_conv(P, Q, M, N, C, cW, cH)
    BM = 32 * NW
    mt, nt = (M + BM - 1) // BM, (N + BN - 1) // BN
    todo = tiles if tiles is not None else [(i, j) for i in range(mt) for j in range(nt)]
    seen = 0
    for (i, j) in todo:
        out = run_tile(P, Q, M, N, C, cW, cH, i * BM, j * BN, BN, NW)
        for (m, n), v in out.items():
            assert v == ref[m, n], ((i, j), m, n, v, ref[m, n])
            seen += 1
    if tiles is None:
        assert seen == M * N, (seen, M * N)
    return seen


if __name__ == '__main__':
    assert max_bank_conflict() == 1
    print('swizzle: every 32-row ds_read_b128 fragment is conflict free')
    print('BN=128 NW=4 H=4 :', check(), 'outputs equal the direct convolution')
    print('BN=64  NW=4 H=8 :', check(Nb=1, cW=12, cH=8, C=128, N=64, BN=64, NW=4, seed=1), 'outputs equal the direct convolution')
