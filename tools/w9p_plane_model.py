"""Lane-level replay (plain Python, no GPU) of the index arithmetic of wgrad9p_kernel (lstm_ctc_ocr_amd/csrc/wgrad9.hip): the plane layout
of a 128-pixel step in LDS (X halo planes + dY rows in (feature row, column) order), the LDS-DMA fill, and the transposing fragment reads
of every (wave, K block, tap, half).  tests/test_w9p_plane_model.py asserts that the k-th contraction element of the X fragment of tap
(dw, dh) is the pixel (column + dw, feature row + dh) of the dY fragment's k-th pixel — or zero outside the image — for every lane, and
that the MFMAs the kernel skips are exactly those whose 32 pixels are all padding.  The formulas restate the kernel's; keep them in step.

ds_read_b64_tr_b16 (pinned on hardware by tests/test_gpu_kernels.py::test_probe_tr16): lane (g4 = lane >> 4, L = lane & 15) SUPPLIES the
8-byte piece (row 4 g4 + (L >> 2), channels 4 (L & 3) .. + 3) of a 4 x 16 block and RECEIVES elements (row 4 g4 + e, channel L), e = 0..3.
"""


class Geometry(object):
    def __init__(self, H):
        self.H = H
        self.NC = 128 // H
        self.PS = (self.NC + 2 + 7) // 8 * 8
        self.XROWS = H * self.PS
        self.XPIECES = self.XROWS // 8
        self.NDMA = 5


def dma_fill(g, step_col0, W):
    """Stage contents after the DMA of one step: dict row -> ('x', column, h) | ('y', column, h) | None (zeros); rows < XROWS are the X planes."""
    H = g.H
    edge_l, edge_r = step_col0 % W == 0, (step_col0 + g.NC) % W == 0
    rows = {}
    for wave in range(8):
        for i in range(g.NDMA):
            u = wave + 8 * i
            for rr in range(8):
                if u < g.XPIECES:
                    r = 8 * u + rr
                    h, cp = r // g.PS, r % g.PS
                    ok = cp < g.NC + 2 and not (cp == 0 and edge_l) and not (cp == g.NC + 1 and edge_r)
                    val = ('x', step_col0 - 1 + cp, h) if ok else None
                elif u < g.XPIECES + 16:
                    r2 = 8 * (u - g.XPIECES) + rr
                    h, col = r2 // g.NC, r2 % g.NC
                    r, val = g.XROWS + r2, ('y', step_col0 + col, h)
                else:
                    continue
                assert r not in rows, "stage row written twice"
                rows[r] = val
    return rows


def planes_of(g, kh, kk, dh):
    """Planes the two 16-pixel halves of K block kk of wave group kh read at feature shift dh (kernel macro W9P_PLANE)."""
    if g.H == 4:
        p = kh * 2 + kk + dh
        return p, p
    p = kh * 4 + kk * 2 + dh
    return p, p + 1


def a_fragment_rows(g, kh, kk, tap, lane):
    """Rows (or None = zero block / not read) whose elements this lane RECEIVES for the X fragment of (kk, tap): 8 contraction elements
    = 4 of the first read + 4 of the second.  Returns (live, [row or None] * 8)."""
    dw, dh = tap // 3 - 1, tap % 3 - 1
    g4 = lane >> 4
    pl, ph = planes_of(g, kh, kk, dh)
    okl, okh = 0 <= pl < g.H, 0 <= ph < g.H
    out = []
    for e in range(4):                                  # first read: supplied rows cp = 1 + 4 g4 + (L >> 2) + dw; received rows 4 g4 + e
        cp = 1 + 4 * g4 + e + dw
        out.append(pl * g.PS + cp if okl else None)
    for e in range(4):                                  # second read: + 16 rows (H = 4) or the next plane (H = 8)
        cp = 1 + 4 * g4 + e + dw + (16 if g.H == 4 else 0)
        out.append(ph * g.PS + cp if okh else None)
    return (okl or okh), out


def b_fragment_rows(g, kh, kk, lane):
    """dY rows of the same 8 contraction elements: rows kh * 64 + kk * 32 + 4 g4 + e (+ 16) of the dY tile."""
    g4 = lane >> 4
    base = g.XROWS + kh * 64 + kk * 32
    return [base + 4 * g4 + e for e in range(4)] + [base + 16 + 4 * g4 + e for e in range(4)]


# ---- general widths (round 6, wgrad9p_kernel<H, LA, CONT, GENW = true>): a step's NC columns may cross ONE image boundary (W >= NC) ---------------------
# The planes are staged as for whole-image steps (dma_fill above works for any W: plane row 1 + c = local column c).  Where local column b of the step
# is the first column of the next image, two contraction elements change: the +1 tap of column b - 1 and the -1 tap of column b read a ZERO ROW of the
# plane (rows NC + 2 .. PS - 1 arrive as zeros with every stage) — the lane that supplies that pixel is redirected for that tap.
def boundary_of(g, step_col0, W):
    """first local column of the NEXT image inside this step, or NC if the step stays inside one image (W >= NC: at most one)"""
    assert W >= g.NC
    b = W - step_col0 % W
    return b if b < g.NC else g.NC


def zero_rows(g):
    """(row read by a redirected FIRST read, base row of a redirected SECOND read before its immediate) — kernel constants ZLO, ZHI"""
    zlo = g.NC + 4
    return zlo, (zlo - 16 if g.H == 4 else zlo)


def a_fragment_rows_genw(g, kh, kk, tap, lane, b):
    """a_fragment_rows() with the boundary redirect: the lane SUPPLYING pixel column c = 4 g4 + e (+ 16 in the second read at H = 4) reads a zero row
    when (c == b and dw == -1) or (c == b - 1 and dw == +1)"""
    dw, dh = tap // 3 - 1, tap % 3 - 1
    g4 = lane >> 4
    pl, ph = planes_of(g, kh, kk, dh)
    okl, okh = 0 <= pl < g.H, 0 <= ph < g.H
    zlo, zhi = zero_rows(g)
    nb = b < g.NC

    def redirected(c):
        return nb and ((c == b and dw == -1) or (c == b - 1 and dw == 1))
    out = []
    for e in range(4):
        c = 4 * g4 + e
        out.append((pl * g.PS + (zlo if redirected(c) else 1 + c + dw)) if okl else None)
    for e in range(4):
        c = 4 * g4 + e + (16 if g.H == 4 else 0)
        imm = 16 if g.H == 4 else 0
        out.append((ph * g.PS + ((zhi + imm) if redirected(c) else 1 + c + dw)) if okh else None)
    return (okl or okh), out
