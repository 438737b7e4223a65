"""Host-side index model of csrc/wgrad9.hip (no GPU needed): replays the kernel's address arithmetic — LDS-DMA source
swizzle, ds_read_b64_tr_b16 gather, tap shifts, SAME-padding masks, k permutation, accumulator ownership, split / tile map —
in numpy on small shapes and compares with a direct evaluation of  dW[t][ci][co] = sum_m X[m + shift(t)][ci] * dY[m][co].
A mistake in any of the formulas shows up here before a GPU minute is spent.   python tools/wgrad9_model.py
"""
import itertools
import sys

import numpy as np

NDMA, STAGE = 5, 8 * 5 * 1024


def plan(M, W, H, Cin, Cout):
    T = (Cin // 64) * (Cout // 64)
    S = 1
    while S * 2 * T <= 256 and M // (S * 2) >= 512:
        S *= 2
    kps = -(-(-(-M // S)) // 128) * 128
    mp = 1 if S >= 8 else (2 if T % (8 // S) == 0 else 0)
    return S, kps, mp


def reference(X, dY, W, H):
    M, Cin = X.shape
    Cout = dY.shape[1]
    dW = np.zeros((9, Cin, Cout))
    m = np.arange(M)
    h, w = m % H, (m // H) % W
    for t in range(9):
        dw_, dh_ = t // 3 - 1, t % 3 - 1
        ok = (w + dw_ >= 0) & (w + dw_ < W) & (h + dh_ >= 0) & (h + dh_ < H)
        src = m + dw_ * H + dh_
        Xs = np.zeros_like(X)
        Xs[ok] = X[src[ok]]
        dW[t] = Xs.T @ dY
    return dW


def kernel_model(X, dY, W, H, force=None):
    M, Cin = X.shape
    Cout = dY.shape[1]
    S, kps, mp = force or plan(M, W, H, Cin, Cout)
    T_ci, T_co = Cin // 64, Cout // 64
    T = T_ci * T_co
    part = np.full((S, 9, Cin, Cout), np.nan)
    cs_part = np.full((S, Cout), np.nan)
    NR = 128 + 2 * H + 2
    nhalo = (NR + 7) >> 3
    NRp = nhalo << 3
    hs = {4: 2, 8: 3, 16: 4}[H]
    ncol = 32 >> hs
    lane = np.arange(64)
    rr, pp = lane >> 3, lane & 7
    qsrc = ((((pp >> 1) ^ ((rr >> 1) & 3)) << 1) | (pp & 1)) * 8
    g4, L = lane >> 4, lane & 15
    seen = set()
    for b in range(S * T):
        if mp == 1:
            x, q = b & 7, b >> 3
            split, tile = (q // T) * 8 + x, q % T
        elif mp == 2:
            x, q, G = b & 7, b >> 3, 8 // S
            split, tile = x // G, (x % G) * (T // G) + q
        else:
            split, tile = b // T, b % T
        assert (split, tile) not in seen and split < S and tile < T
        seen.add((split, tile))
        ti, tj = tile // T_co, tile % T_co
        ci0, co0 = ti * 64, tj * 64
        kbeg = split * kps
        kend = min(M, kbeg + kps)
        nsteps = (kend - kbeg + 127) >> 7 if kend > kbeg else 0
        acc = np.zeros((8, 9, 4, 4, 64))                 # [wave][tap][c][r][lane]
        cs = np.zeros(64)
        wc = (kbeg >> hs) % W
        for step in range(nsteps):
            lds = np.full(STAGE // 2, np.nan)
            for wave in range(8):
                for i in range(NDMA):
                    u = wave + 8 * i
                    vals = np.zeros((64, 8))
                    if u < nhalo:
                        r = 8 * u + rr
                        px = kbeg - (H + 1) + r + step * 128
                        ok = (r < NR) & (px >= 0) & (px < M)
                        for l in np.nonzero(ok)[0]:
                            vals[l] = X[px[l], ci0 + qsrc[l]: ci0 + qsrc[l] + 8]
                    elif u < nhalo + 16:
                        r = 8 * (u - nhalo) + rr
                        px = kbeg + r + step * 128
                        for l in np.nonzero(px < kend)[0]:
                            vals[l] = dY[px[l], co0 + qsrc[l]: co0 + qsrc[l] + 8]
                    lds[u * 512: (u + 1) * 512] = vals.reshape(-1)       # lane-linear, 16 B = 8 elements per lane

            def tr_read(addr_bytes):           # addr per lane -> [lane][4] elements
                out = np.zeros((64, 4))
                for l in range(64):
                    grp, Ll = l >> 4, l & 15
                    for e in range(4):
                        sup = grp * 16 + 4 * e + (Ll >> 2)          # supplying lane of row e
                        a = addr_bytes[sup]
                        assert a % 8 == 0
                        out[l, e] = lds[a // 2 + (Ll & 3)]
                return out

            for wave in range(8):
                cb, kh = wave & 3, wave >> 2
                rowl = kh * 64 + 4 * g4 + (L >> 2)
                offB = [NRp * 128 + rowl * 128 + ((c ^ ((rowl >> 1) & 3)) << 5) + (L & 3) * 8 for c in range(4)]
                offA = []
                for t in range(9):
                    a0 = rowl + (H + 1) + (t // 3 - 1) * H + (t % 3 - 1)
                    offA.append(a0 * 128 + ((cb ^ ((a0 >> 1) & 3)) << 5) + (L & 3) * 8)
                mlo = (4 * g4) % H == 0
                mhi = (4 * g4 + 4) % H == 0
                for kk in range(2):
                    cbase = (wc + ((kh * 64 + kk * 32) >> hs)) % W
                    bnd = cbase == 0 or cbase + ncol >= W
                    w0 = cbase + ((4 * g4) >> hs)
                    w1 = cbase + ((4 * g4 + 16) >> hs)
                    assert (w0 < 2 * W).all() and (w1 < 2 * W).all()
                    w0 = np.where(w0 >= W, w0 - W, w0)
                    w1 = np.where(w1 >= W, w1 - W, w1)
                    Bf = []
                    for c in range(4):
                        lo = tr_read(offB[c] + kk * 32 * 128)
                        hi = tr_read(offB[c] + kk * 32 * 128 + 16 * 128)
                        Bf.append(np.concatenate([lo, hi], 1))          # [lane][8]
                    for t in range(9):
                        lo = tr_read(offA[t] + kk * 32 * 128)
                        hi = tr_read(offA[t] + kk * 32 * 128 + 16 * 128)
                        dh_, dw_ = t % 3 - 1, t // 3 - 1
                        if dh_ < 0:
                            lo[mlo, 0] = 0; hi[mlo, 0] = 0
                        if dh_ > 0:
                            lo[mhi, 3] = 0; hi[mhi, 3] = 0
                        if dw_ != 0 and bnd:
                            bad = 0 if dw_ < 0 else W - 1
                            lo[w0 == bad] = 0; hi[w1 == bad] = 0
                        Af = np.concatenate([lo, hi], 1)
                        assert not np.isnan(Af).any()
                        for c in range(4):
                            assert not np.isnan(Bf[c]).any()
                            # D[i][j] = sum over (lane group, slot) A[lane(g, i)][s] * B[lane(g, j)][s]
                            A3 = Af.reshape(4, 16, 8)            # [g][i][s]
                            B3 = Bf[c].reshape(4, 16, 8)         # [g][j][s]
                            D = np.einsum('gis,gjs->ij', A3, B3)
                            for r in range(4):
                                acc[wave, t, c, r] += D[4 * g4 + r, L]   # lane (g4, L) register r holds D[4 g4 + r][L]
            # column sums by threads: source chunk tid & 7, rows tid >> 3 and + 64
            tid = np.arange(512)
            csq, csr = tid & 7, tid >> 3
            offC = NRp * 128 + csr * 128 + (((((csq >> 1) ^ ((csr >> 1) & 3)) << 1) | (csq & 1)) << 4)
            for th in range(512):
                for extra in (0, 8192):
                    a = (offC[th] + extra) // 2
                    cs[csq[th] * 8: csq[th] * 8 + 8] += lds[a: a + 8]
            wc = (wc + (128 >> hs)) % W
        tot = acc[:4] + acc[4:]
        for cb in range(4):
            for t, c, r in itertools.product(range(9), range(4), range(4)):
                ci = ci0 + cb * 16 + g4 * 4 + r
                co = co0 + c * 16 + L
                part[split, t, ci, co] = tot[cb, t, c, r]
        if ti == 0:
            cs_part[split, co0: co0 + 64] = cs
    assert not np.isnan(part).any() and not np.isnan(cs_part).any()
    return part.sum(0), cs_part.sum(0)


def check(Nb, W, H, Cin, Cout, force=None, seed=0):
    rng = np.random.RandomState(seed)
    M = Nb * W * H
    X = rng.randint(-3, 4, (M, Cin)).astype(np.float64)
    dY = rng.randint(-3, 4, (M, Cout)).astype(np.float64)
    got, cs = kernel_model(X, dY, W, H, force)
    ref = reference(X, dY, W, H)
    ok = np.array_equal(got, ref) and np.array_equal(cs, dY.sum(0))
    print('Nb %d W %d H %d Cin %d Cout %d plan %s: %s' % (Nb, W, H, Cin, Cout, force or plan(M, W, H, Cin, Cout), 'OK' if ok else 'MISMATCH'))
    if not ok:
        bad = np.argwhere(got != ref)
        print('  first mismatches (tap, ci, co):', bad[:5].tolist(), 'of', len(bad))
    return ok


if __name__ == '__main__':
    res = [check(2, 16, 4, 64, 64), check(1, 8, 16, 64, 64), check(3, 12, 8, 64, 128, force=(2, 256, 0)),
           check(2, 40, 4, 128, 64, force=(8, 128, 1)), check(5, 13, 4, 128, 64, force=(4, 128, 2), seed=3), check(2, 9, 8, 64, 64, seed=5)]
    sys.exit(0 if all(res) else 1)
