"""CPU replay of conv_ws.hip's index algebra (weight-stationary persistent 3x3 convolution): the formulas below are TRANSCRIBED from the
kernel — DMA lane -> (source element, LDS position), fragment-read addresses (lane register + immediate), the wave-local staging
swizzle of the write-out, the mask rows' DMA, the pool windows and the workgroup -> (channel tile, pixel tiles) map — and checked against
what a 3x3 SAME convolution over [N, W, H, C] must read and write.  No GPU: tests/test_ws_plane_model.py runs it for every instance."""
import itertools

Z = ('zero',)


class Cfg(object):
    def __init__(self, H, NC, KSPLIT):
        self.H, self.NC, self.KSPLIT = H, NC, KSPLIT
        self.BM = NC * H
        self.PG = 4 // KSPLIT
        self.CF = min(NC, 16)
        self.HF = 16 // self.CF
        self.HW = 4 * self.HF
        self.PS = NC + 2
        self.CHB = (H + 2) * self.PS * 128
        self.BUFB = KSPLIT * self.CHB
        self.PPC = H * self.PS // 8
        self.PI = KSPLIT * self.PPC // 4
        self.NFE = 4 // KSPLIT
        self.HWE = self.NFE * self.HF
        self.NPE = NC * self.HWE
        self.XOFF = 2 * self.BUFB
        self.SOFF = self.XOFF + (4 * 8192 if KSPLIT == 2 else 0)
        self.STW = self.NPE * 128
        self.TOFF = self.SOFF + 4 * self.STW
        self.BOFF = self.TOFF + 4 * self.PI * 128
        self.MOFF = self.BOFF + 256
        self.LDS_PLAIN, self.LDS_MASK = self.MOFF, self.MOFF + 4 * self.STW
        assert NC == self.CF and self.PG * self.HW == H and (H * self.PS) % 8 == 0 and (KSPLIT * self.PPC) % 4 == 0
        assert self.LDS_MASK <= 163840 and (H + 1) * self.PS * 128 < 65536
        assert ((NC + 1) * H + H) * 128 * 2 // 16 * 4 < 65536


INSTANCES = [(16, 16, 1), (16, 8, 2), (8, 16, 2)]


def dma_tile(g, t, buf, W, M, lds):
    """halo of pixel tile t -> LDS units (16 B each), as issue_halo() + the lane table do it.  lds: dict unit address -> content."""
    H, NC, PS, C = g.H, g.NC, g.PS, 64 * g.KSPLIT
    col0 = t * NC
    edge_l, edge_r = col0 % W == 0, (col0 + NC) % W == 0
    base = (col0 - 1) * H * C * 2
    em = (1 if edge_l else 0) | (2 if edge_r else 0)
    for wave in range(4):
        for j in range(g.PI):
            q = j * 4 + wave
            c, pc = q // g.PPC, q % g.PPC
            for lane in range(64):
                rsub = lane >> 3
                r = pc * 8 + rsub
                h, cp = r // PS, r % PS
                off = (((cp * H + h) * C + c * 64) + (((lane & 7) ^ (cp & 7)) << 3)) * 2
                assert off % 16 == 0 and ((off >> 2) | 3) < 65536
                e = (off >> 2) | (1 if cp == 0 else 0) | (2 if cp == NC + 1 else 0)
                dst = buf * g.BUFB + c * g.CHB + PS * 128 + pc * 1024 + lane * 16
                if e & em:
                    lds[dst] = Z
                else:
                    src = base + ((e & ~3) << 2)
                    assert 0 <= src and src + 16 <= M * C * 2, (t, wave, j, lane, src)
                    el = src // 2
                    lds[dst] = ('px', el // C, el % C)          # eight channels from (pixel, first channel)


def zero_planes(g, lds):
    ZU = g.PS * 8
    for i in range(2 * g.KSPLIT * 2 * ZU):
        img, r = i // (2 * ZU), i % (2 * ZU)
        off = img * g.CHB + (0 if r < ZU else (g.H + 1) * g.PS * 128) + (r % ZU) * 16
        lds[off] = Z


def check_fragment_reads(g, t, buf, W, lds):
    """every ds_read_b128 of the K loop returns the eight channels the MFMA lane must multiply"""
    H, NC, PS, CF, HF = g.H, g.NC, g.PS, g.CF, g.HF
    col0 = t * NC
    img = col0 // W
    for wave in range(4):
        pg, kh = (wave, 0) if g.KSPLIT == 1 else (wave & 1, wave >> 1)
        for lane in range(64):
            frow, fq = lane & 15, lane >> 4
            colf, hsub = frow % CF, frow // CF
            for s in range(18):
                tap, j = s >> 1, s & 1
                dw, dh = tap // 3, tap % 3 - 1
                cp = colf + dw
                pb = buf * g.BUFB + kh * g.CHB + ((pg * 4 * HF + hsub) * PS + cp) * 128 + (((j * 4 + fq) ^ (cp & 7)) << 4)
                for b in range(4):
                    imm = ((b * HF + 1 + dh) * PS) * 128
                    assert 0 <= imm < 65536
                    got = lds.get(pb + imm)
                    col = col0 + colf + dw - 1
                    h = (pg * 4 + b) * HF + hsub + dh
                    if h < 0 or h >= H or col // W != img or col < 0:
                        want = Z
                    else:
                        want = ('px', col * H + h, kh * 64 + (j * 4 + fq) * 8)
                    assert got == want, (g.H, g.NC, g.KSPLIT, t, wave, lane, s, b, got, want)


def check_write_out(g, t, N, n0):
    """accumulator lane -> staging -> 16-byte row units -> global rows: every (pixel, channel) lands where [M][N] wants it, once"""
    H, NC, CF, HF, HWE = g.H, g.NC, g.CF, g.HF, g.HWE
    col0 = t * NC
    written = {}
    for wave in range(4):
        pg, kh = (wave, 0) if g.KSPLIT == 1 else (wave & 1, wave >> 1)
        B0 = 2 * kh if g.KSPLIT == 2 else 0
        h_lo = pg * g.HW + (kh * HWE if g.KSPLIT == 2 else 0)
        S = {}
        for lane in range(64):
            frow, fq = lane & 15, lane >> 4
            colf, hsub = frow % CF, frow // CF
            for i in range(g.NFE):
                row = colf * HWE + i * HF + hsub
                b = B0 + i
                m_acc = (col0 + colf) * H + (pg * 4 + b) * HF + hsub            # the pixel this accumulator fragment lane holds
                for a in range(4):
                    slot = a * 4 + fq
                    addr = row * 128 + ((slot ^ ((colf & 3) << 2)) << 3)
                    assert addr not in S and 0 <= addr < g.STW
                    S[addr] = (m_acc, n0 + a * 16 + fq * 4)
        assert len(S) == g.NPE * 16
        for it in range(g.NPE // 8):
            for lane in range(64):
                idx = it * 64 + lane
                row, u = idx >> 3, idx & 7
                col = row // HWE
                addr = row * 128 + ((u ^ ((col & 3) << 1)) << 4)
                lo, hi = S[addr], S[addr + 8]
                m = (col0 + col) * H + h_lo + row % HWE
                assert lo == (m, n0 + u * 8) and hi == (m, n0 + u * 8 + 4), (wave, it, lane, lo, hi, m)
                key = (m, n0 + u * 8)
                assert key not in written
                written[key] = True
                # the mask rows' DMA of a MASK instance delivers row `idx` of this wave's write-out from the same place
                rsub = lane >> 3
                mrow = it * 8 + rsub
                mbase = (col0 * H + h_lo) * N + n0
                v = (((mrow // HWE) * H + mrow % HWE) * N + (lane & 7) * 8)
                assert mbase + v == m * N + n0 + u * 8
        # fused pools read whole windows inside the wave's own rows
        for kind in (1, 2):
            if kind == 1:
                for q in range(NC * (HWE // 2)):
                    col, ph = q // (HWE // 2), q % (HWE // 2)
                    r0 = col * HWE + 2 * ph
                    m0 = (col0 + col) * H + h_lo + 2 * ph
                    assert m0 % 2 == 0
                    pm = (col0 + col) * (H // 2) + (h_lo >> 1) + ph
                    assert pm == m0 // 2 and r0 + 1 < g.NPE
            else:
                for q in range((NC // 2) * (HWE // 2)):
                    pc, ph = q // (HWE // 2), q % (HWE // 2)
                    c0 = 2 * pc
                    rows = [c0 * HWE + 2 * ph, c0 * HWE + 2 * ph + 1, (c0 + 1) * HWE + 2 * ph, (c0 + 1) * HWE + 2 * ph + 1]
                    assert max(rows) < g.NPE and col0 % 2 == 0 and h_lo % 2 == 0
                    pm = ((col0 >> 1) + pc) * (H // 2) + (h_lo >> 1) + ph
                    # reference: pooled tensor [N, W/2, H/2, C] row of window (column pair, row pair)
                    gcol, gh = col0 + c0, h_lo + 2 * ph
                    assert pm == (gcol // 2) * (H // 2) + gh // 2
    return written


def grid_map(mtiles, ntiles, cus):
    """launch_ws(): slots, tiles per slot, grid, XCD map; returns the list of (channel tile, pixel tiles) per workgroup id"""
    slots = max(1, cus // ntiles)
    slots = min(slots, mtiles)
    per_slot = (mtiles + slots - 1) // slots
    slots = (mtiles + per_slot - 1) // per_slot
    grid = slots * ntiles
    xcd = grid % 8 == 0 and (grid // 8) % ntiles == 0
    out = []
    for wid in range(grid):
        if xcd:
            per_xcd, j = grid >> 3, wid >> 3
            nt, slot = j % ntiles, (wid & 7) * (per_xcd // ntiles) + j // ntiles
        else:
            nt, slot = wid % ntiles, wid // ntiles
        out.append((nt, list(range(slot * per_slot, min(mtiles, slot * per_slot + per_slot)))))
    return out


def check_instance(H, NC, KSPLIT, Nb, W):
    g = Cfg(H, NC, KSPLIT)
    C = 64 * KSPLIT
    M = Nb * W * H
    assert W % NC == 0 and M % g.BM == 0
    mtiles = M // g.BM
    lds = {}
    zero_planes(g, lds)
    frozen = dict(lds)
    tiles = sorted(set([0, 1, W // NC - 1, W // NC, mtiles - 1, mtiles // 2]) & set(range(mtiles)))
    for k, t in enumerate(tiles):
        buf = k & 1
        dma_tile(g, t, buf, W, M, lds)
        for a, v in frozen.items():
            assert lds[a] == v                           # the zero planes are never overwritten
        check_fragment_reads(g, t, buf, W, lds)
    N = 128
    for n0 in (0, 64):
        w = check_write_out(g, tiles[-1], N, n0)
        assert len(w) == g.BM * 8                        # 64 channels = eight 16-byte units per pixel
    return True


if __name__ == '__main__':
    for H, NC, K in INSTANCES:
        for Nb, W in ((2, 2 * NC), (3, 3 * NC), (1, NC)):
            check_instance(H, NC, K, Nb, W)
        print('instance', (H, NC, K), 'ok: LDS', Cfg(H, NC, K).LDS_PLAIN, Cfg(H, NC, K).LDS_MASK)
    for mt, nt, cus in ((512, 2, 256), (1024, 1, 256), (256, 4, 256), (32, 2, 256), (7, 3, 256), (1000, 3, 304)):
        wg = grid_map(mt, nt, cus)
        seen = sorted((n, t) for n, ts in wg for t in ts)
        assert seen == sorted(itertools.product(range(nt), range(mt))), (mt, nt, cus)
    print('grid maps ok')
