"""wgrad9p_kernel's plane layout replayed on the CPU (tools/w9p_plane_model.py): the X fragment of tap (dw, dh) must hold, element by
element of the contraction, the pixel (column + dw, feature row + dh) of the dY fragment's pixel — zeros outside the image — and the
skipped MFMAs must be exactly the all-padding ones.  The kernel itself is checked against torch autograd in tests/test_gpu_kernels.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import w9p_plane_model as wm   # noqa: E402


@pytest.mark.parametrize("H", [4, 8])
def test_stage_geometry(H):
    g = wm.Geometry(H)
    assert g.PS % 8 == 0 and g.XPIECES + 16 <= 8 * g.NDMA                  # planes + dY tile fit the 40 KiB stage
    assert (g.PS, g.XROWS) == ((40, 160) if H == 4 else (24, 192))


@pytest.mark.parametrize("H", [4, 8])
@pytest.mark.parametrize("where", ["first", "middle", "last", "whole"])
def test_x_fragment_is_the_shifted_dy_pixel(H, where):
    g = wm.Geometry(H)
    W = g.NC if where == "whole" else 4 * g.NC
    col0 = {"first": 3 * W, "middle": 3 * W + 2 * g.NC, "last": 3 * W + 3 * g.NC, "whole": 5 * W}[where]
    st = wm.dma_fill(g, col0, W)
    assert sum(1 for r in range(g.XROWS + 128) if r in st) == g.XROWS + 128     # every row of the planes and of the dY tile is filled
    img = col0 // W
    skipped = 0
    for kh in range(2):
        for kk in range(2):
            for tap in range(9):
                dw, dh = tap // 3 - 1, tap % 3 - 1
                all_padding = True
                for lane in range(0, 64, 16):           # the row pattern depends on g4 = lane >> 4 only
                    live, arows = wm.a_fragment_rows(g, kh, kk, tap, lane)
                    brows = wm.b_fragment_rows(g, kh, kk, lane)
                    for ar, br in zip(arows, brows):
                        kind, col, h = st[br]
                        assert kind == 'y'
                        want_col, want_h = col + dw, h + dh
                        inside = 0 <= want_h < H and want_col // W == img and want_col >= 0
                        got = st.get(ar) if ar is not None else None
                        if inside:
                            all_padding = False
                            assert got == ('x', want_col, want_h), (kh, kk, tap, lane, got, (want_col, want_h))
                        else:
                            assert got is None              # zero block, zero halo column, or a plane that is not read
                    if not live:
                        assert all(a is None for a in arows)
                live_any = wm.a_fragment_rows(g, kh, kk, tap, 0)[0]
                assert live_any == (not all_padding) or (live_any and H == 8)   # H = 8 never skips: one half always exists
                skipped += not live_any
    assert skipped == (6 if H == 4 else 0)              # H = 4: 3 taps x (first K block of kh 0, last of kh 1) = a sixth of 36


@pytest.mark.parametrize("H", [4, 8])
def test_general_width_boundary_redirect(H):
    """wgrad9p's GENW form (round 6): a step's NC columns may cross one image boundary; the planes are staged as for whole-image steps and the two
    lanes whose +1 / -1 neighbour belongs to the other image read a zero row of the plane for that tap.  For every width NC <= W <= 2 NC + 1 (and a
    few wider ones) and EVERY step of three images: each contraction element of every tap's X fragment is the shifted pixel of the same image, or
    zero — at the boundary, at the image edges, outside the feature range."""
    g = wm.Geometry(H)
    zlo, zhi = wm.zero_rows(g)
    assert g.NC + 2 <= zlo < g.PS and g.NC + 2 <= zhi + (16 if H == 4 else 0) < g.PS          # zero rows of every plane (never DMA'd with data)
    for W in list(range(g.NC, 2 * g.NC + 2)) + [3 * g.NC - 1, 3 * g.NC + 5, 80 if H == 4 else 79, 79 if H == 4 else 78]:
        total = 3 * W
        total -= total % g.NC                                        # whole steps only (the kernel requires M % 128 == 0)
        crossed = 0
        for col0 in range(0, total, g.NC):
            b = wm.boundary_of(g, col0, W)
            crossed += b < g.NC
            st = wm.dma_fill(g, col0, W)                             # the whole-image staging, unchanged
            assert sum(1 for r in range(g.XROWS + 128) if r in st) == g.XROWS + 128
            for kh in range(2):
                for kk in range(2):
                    for tap in range(9):
                        dw, dh = tap // 3 - 1, tap % 3 - 1
                        for lane in range(0, 64, 16):
                            live, arows = wm.a_fragment_rows_genw(g, kh, kk, tap, lane, b)
                            brows = wm.b_fragment_rows(g, kh, kk, lane)
                            for ar, br in zip(arows, brows):
                                kind, col, h = st[br]
                                assert kind == 'y'
                                want_col, want_h = col + dw, h + dh
                                inside = 0 <= want_h < H and want_col >= 0 and want_col // W == col // W
                                got = st.get(ar) if ar is not None else None
                                if inside:
                                    assert got == ('x', want_col, want_h), (W, col0, b, kh, kk, tap, lane, got, (want_col, want_h))
                                else:
                                    assert got is None, (W, col0, b, kh, kk, tap, lane, got)
        assert crossed > 0 or W % g.NC == 0
