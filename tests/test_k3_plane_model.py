"""conv_k3_kernel's plane layout replayed lane by lane on the CPU (tools/k3_plane_model.py restates the kernel's index arithmetic):
what the LDS-DMA puts where, what every fragment read of every wave / lane / tap fetches, which taps are skipped, bank conflicts of the
reads, and the staged write-out swizzle.  The GPU parity tests (tests/test_gpu_kernels.py) check the kernel itself; this keeps the
address algebra pinned where no GPU is needed."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import k3_plane_model as km   # noqa: E402

CONFIGS = [(4, 128), (8, 128), (16, 128), (4, 64), (8, 64), (16, 64)]       # (feature rows, channels per tile): tiles A and D


@pytest.mark.parametrize("H,BN", CONFIGS)
def test_geometry_matches_the_kernel_constants(H, BN):
    g = km.Geometry(H, BN)
    assert g.FM * g.WMW == 16 and g.PS % 8 == 0 and H * g.PS <= g.PPIECES * 8 and g.PI <= 8
    assert (g.PPIECES, g.PI) == ((40, 5) if H in (4, 8) else (48, 6))
    assert g.CBW * g.HW == g.FM and g.WGC * (H // g.HW) == g.WMW        # the wave groups tile (column blocks) x (plane groups)
    assert (g.FM // 2) % (2 * g.CBW) == 0                                # the feature-axis pool partner b ^ CBW stays in one K half
    # the largest immediate of a fragment read fits ds_read's 16-bit offset field
    assert ((g.HW + 1) * g.PS + (g.CBW - 1) * 16) * 128 < 65536


@pytest.mark.parametrize("H,BN", CONFIGS)
@pytest.mark.parametrize("tile_in_image", ["first", "middle", "last", "whole"])
def test_every_fragment_read_fetches_the_pixel_the_convolution_needs(H, BN, tile_in_image):
    g = km.Geometry(H, BN)
    W = g.NC if tile_in_image == "whole" else 3 * g.NC
    col0 = {"first": 5 * W, "middle": 5 * W + g.NC, "last": 5 * W + 2 * g.NC, "whole": 7 * W}[tile_in_image]
    lds = km.dma_fill(g, col0, W, 0)
    assert len(lds) == g.PPIECES * 64                       # every position of the padded buffer is written exactly once per chunk
    img = col0 // W
    for wave in range(km.NW):
        for lane in range(0, 64, 5):                        # a third of the lanes is plenty (the pattern has period 16 / 4)
            for (b, tap, live, row, pos, kq) in km.fragment_reads(g, wave, lane):
                col, h = km.expected_pixel(g, col0, wave, lane, b, tap)
                if not live:                                # skipped: the plane does not exist
                    assert not (0 <= h < H)
                    continue
                assert 0 <= h < H and 0 <= row < g.PPIECES * 8
                got = lds[(row, pos)]
                if col // W != img or col < 0:              # the neighbour column belongs to another image: must be zeros
                    assert got is None
                else:
                    assert got == (col, h, kq), (wave, lane, b, tap, got, (col, h, kq))


@pytest.mark.parametrize("H,BN", CONFIGS)
def test_fragment_reads_are_bank_conflict_free(H, BN):
    g = km.Geometry(H, BN)
    assert sum(km.bank_conflicts(g, wave) for wave in range(km.NW)) == 0


@pytest.mark.parametrize("H,BN", CONFIGS)
def test_output_pixels_and_staged_write_out(H, BN):
    g = km.Geometry(H, BN)
    for wn, lps in km.output_pixels(g, 0).items():
        assert sorted(lps) == list(range(256))             # the wave groups of a channel half cover the tile's 256 pixels once
    dup, bad, n = km.staged_roundtrip(g)
    assert dup == 0 and bad == 0 and n == 256 * BN // 4    # every 8-byte slot of the staged image written once, read back in order


GENW_CASES = [(4, 128, 40, 0), (4, 128, 40, 64), (4, 64, 23, 128), (8, 128, 24, 32), (8, 64, 50, 96), (2, 128, 64, 128), (2, 64, 64, 0),
              (16, 64, 20, 16), (4, 128, 64, 64), (8, 64, 32, 64)]


@pytest.mark.parametrize("H,BN,W,col0", GENW_CASES)
def test_general_width_layout_reads_the_right_pixels(H, BN, W, col0):
    """Tiles that cross image boundaries (W not a multiple of the tile's columns): a zero row in front of every interior boundary column."""
    g = km.Geometry(H, BN)
    assert km.genw_fits(g, W)
    lds = km.genw_dma_fill(g, col0, W)
    assert len(lds) == g.PPIECES * 64
    for wave in range(km.NW):
        for lane in range(64):
            for (b, tap, live, row, pos, kq) in km.genw_fragment_reads(g, col0, W, wave, lane):
                col, h = km.expected_pixel(g, col0, wave, lane, b, tap)
                own = km.expected_pixel(g, col0, wave, lane, b, 4)[0]        # the pixel's own column (centre tap)
                if not live:
                    assert not (0 <= h < H)
                    continue
                got = lds[(row, pos)]
                if col < 0 or col // W != own // W:                          # the neighbour column belongs to another image
                    assert got is None, (wave, lane, b, tap, got)
                else:
                    assert got == (col, h, kq), (wave, lane, b, tap, got, (col, h, kq))
