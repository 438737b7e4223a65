"""Lane-level replay (numpy / plain Python, no GPU) of the index arithmetic of conv_k3_kernel (lstm_ctc_ocr_amd/csrc/conv_k3.hip):
the plane layout of the halo tile in LDS, the LDS-DMA fill that produces it, the fragment reads (lane register + immediate) of every
(wave, fragment, tap), the taps that are skipped, and the staged write-out swizzle.  tests/test_k3_plane_model.py asserts on it:
every fragment read returns exactly the pixel / channel chunk the convolution needs (or zeros outside the image), every read is
bank-conflict free, and the staged tile is written once and read back in order.  The formulas below restate the kernel's; keep them in step.
"""
import numpy as np

NW = 8
# ds_read_b128 is serviced in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table)
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


class Geometry(object):
    def __init__(self, H, BN):
        self.H, self.BN = H, BN
        self.BM = 256
        self.WN = BN // 64
        self.WMW = 4 // self.WN
        self.FM = 16 // self.WMW
        self.NC = self.BM // H
        self.NCB = self.NC // 16
        self.PS = (self.NC + 2 + 7) // 8 * 8
        self.PPIECES = (H * self.PS // 8 + NW - 1) // NW * NW
        self.PI = self.PPIECES // NW
        self.CBW = max(1, self.NCB // self.WMW)
        self.HW = self.FM // self.CBW
        self.WGC = self.NCB // self.CBW
        self.static_h = self.HW == H


def dma_fill(g, col0, W, chunk_of_pixel):
    """The halo buffer after the DMA of one chunk: dict (row, pos) -> (pixel column, feature row, source 16-byte chunk) or None (zeros).
    chunk_of_pixel is unused structure-wise (the kernel adds chunk * 128 bytes as a scalar offset); kept for readability."""
    H = g.H
    edge_l, edge_r = col0 % W == 0, (col0 + g.NC) % W == 0
    lds, writers = {}, {}
    for wave in range(NW):
        for j in range(g.PI):
            u = j * NW + wave
            for lane in range(64):
                rsub, pos = lane >> 3, lane & 7
                r = u * 8 + rsub
                h, cp = r // g.PS, r % g.PS
                ok = h < H and cp < g.NC + 2 and not (cp == 0 and edge_l) and not (cp == g.NC + 1 and edge_r)
                key = (r, pos)
                assert key not in writers, "LDS position written twice"
                writers[key] = (wave, j, lane)
                lds[key] = (col0 - 1 + cp, h, pos ^ rsub) if ok else None
    return lds


def fragment_reads(g, wave, lane):
    """For one lane: list of (b, tap, live, row, pos) of the pixel-fragment reads of a K step; row / pos address the halo buffer."""
    kh, wm = wave >> 2, (wave & 3) // g.WN
    cb0, hbase = (wm % g.WGC) * g.CBW, (wm // g.WGC) * g.HW
    top, bot = g.static_h or hbase == 0, g.static_h or hbase + g.HW == g.H
    frow, fq = lane & 15, lane >> 4
    kq = (kh << 2) | fq
    out = []
    for tap in range(9):
        d, dh = tap // 3, tap % 3 - 1
        cp = cb0 * 16 + frow + d
        base_row, pos = (hbase - 1) * g.PS + cp, kq ^ (cp & 7)
        for b in range(g.FM):
            hl = b // g.CBW + dh
            live = not ((hl < 0 and top) or (hl >= g.HW and bot))
            row = base_row + (b // g.CBW + dh + 1) * g.PS + (b % g.CBW) * 16
            out.append((b, tap, live, row, pos, kq))
    return out


def expected_pixel(g, col0, wave, lane, b, tap):
    """(column, feature row) the MFMA operand of (fragment b, tap) must hold for this lane."""
    wm = (wave & 3) // g.WN
    cb0, hbase = (wm % g.WGC) * g.CBW, (wm // g.WGC) * g.HW
    return col0 + (cb0 + b % g.CBW) * 16 + (lane & 15) + (tap // 3 - 1), hbase + b // g.CBW + (tap % 3 - 1)


def output_pixels(g, col0):
    """Every (wave of K half 0, fragment, lane) -> local pixel index lp = local column * H + h; must be a permutation of 0..255 per channel half."""
    seen = {}
    for wave in range(4):
        wm, wn = (wave & 3) // g.WN, (wave & 3) % g.WN
        cb0, hbase = (wm % g.WGC) * g.CBW, (wm // g.WGC) * g.HW
        for b in range(g.FM):
            for frow in range(16):
                lp = ((cb0 + b % g.CBW) * 16 + frow) * g.H + hbase + b // g.CBW
                seen.setdefault(wn, []).append(lp)
    return seen


def staged_roundtrip(g):
    """Writer (wave, lane, fragment, a) -> physical 8-byte slot of row lp; reader (lp, 16-byte unit u) -> physical unit.  Returns
    (#positions written twice, #reader units whose two slots are not the logical ones)."""
    U, SWM = g.BN * 2 // 16, (7 if g.BN == 128 else 3)
    phys = {}
    dup = 0
    for wave in range(NW):
        kh, wm, wn = wave >> 2, (wave & 3) // g.WN, (wave & 3) % g.WN
        cb0, hbase = (wm % g.WGC) * g.CBW, (wm // g.WGC) * g.HW
        FH = g.FM // 2
        for bb in range(FH):
            b = kh * FH + bb
            for lane in range(64):
                frow = lane & 15
                lp = ((cb0 + b % g.CBW) * 16 + frow) * g.H + hbase + b // g.CBW
                for a in range(4):
                    slot = wn * 16 + a * 4 + (lane >> 4)
                    key = (lp, slot ^ ((frow & SWM) << 2))
                    dup += key in phys
                    phys[key] = slot
    bad = 0
    for lp in range(256):
        for u in range(U):
            pu = u ^ (((lp // g.H) & SWM) << 1)
            if phys.get((lp, 2 * pu)) != 2 * u or phys.get((lp, 2 * pu + 1)) != 2 * u + 1:
                bad += 1
    return dup, bad, len(phys)


def bank_conflicts(g, wave):
    """Number of ds_read_b128 lane groups of this wave's pixel-fragment reads (all taps, live fragments) that touch a 16-byte bank slot twice."""
    per_lane = [fragment_reads(g, wave, lane) for lane in range(64)]
    bad = 0
    for i in range(len(per_lane[0])):
        if not per_lane[0][i][2]:
            continue
        for grp in B128_GROUPS:
            slots = [((per_lane[l][i][3] * 128 + per_lane[l][i][4] * 16) % 256) // 16 for l in grp]
            bad += len(set(slots)) != 16
    return bad


# ---------------------------------------------------------------------------------------------------------------------------------
# General image width (GENW): a tile's NC columns may cross image boundaries.  One zero row is inserted in every plane in front of each
# local column c in [1, NC) with (col0 + c) % W == 0, so that the dw = +-1 taps of the columns on either side of the boundary read
# zeros; local column c sits at plane row 1 + c + k(c), k(c) = number of such boundaries in [1, c].
def genw_k(col0, W, c):
    return (col0 + c) // W - col0 // W if c >= 1 else 0


def genw_fits(g, W):
    return g.NC + 2 + (g.NC + W - 1) // W <= g.PS


def genw_dma_fill(g, col0, W):
    """As dma_fill for any W: dict (row, pos) -> (column, h, chunk) or None."""
    H, NC = g.H, g.NC
    edge_l, edge_r = col0 % W == 0, (col0 + NC) % W == 0
    last_row = 1 + NC + genw_k(col0, W, NC - 1)                       # the right halo column's row (plane-relative)
    lds = {}
    for wave in range(NW):
        for j in range(g.PI):
            u = j * NW + wave
            for lane in range(64):
                rsub, pos = lane >> 3, lane & 7
                r = u * 8 + rsub
                h, p = r // g.PS, r % g.PS
                val = None
                if h < H:
                    if p == 0:
                        val = None if edge_l else (col0 - 1, h, pos ^ rsub)
                    elif p == last_row:
                        val = None if edge_r else (col0 + NC, h, pos ^ rsub)
                    elif p < last_row:
                        for k in range(0, 8):                      # the kernel tries the few possible boundary counts
                            c = p - 1 - k
                            if 0 <= c < NC and genw_k(col0, W, c) == k:
                                val = (col0 + c, h, pos ^ rsub)
                                break
                assert (r, pos) not in lds
                lds[(r, pos)] = val
    return lds


def genw_fragment_reads(g, col0, W, wave, lane):
    kh, wm = wave >> 2, (wave & 3) // g.WN
    cb0, hbase = (wm % g.WGC) * g.CBW, (wm // g.WGC) * g.HW
    top, bot = g.static_h or hbase == 0, g.static_h or hbase + g.HW == g.H
    frow, fq = lane & 15, lane >> 4
    kq = (kh << 2) | fq
    out = []
    for tap in range(9):
        d, dh = tap // 3, tap % 3 - 1
        for b in range(g.FM):
            c = (cb0 + b % g.CBW) * 16 + frow                     # local column of this lane's pixel
            p = 1 + c + genw_k(col0, W, c) + (d - 1)               # lane register pbase[d][b % CBW]
            hl = b // g.CBW + dh
            live = not ((hl < 0 and top) or (hl >= g.HW and bot))
            row = (hbase - 1) * g.PS + p + (b // g.CBW + dh + 1) * g.PS
            out.append((b, tap, live, row, kq ^ (p & 7), kq))
    return out
