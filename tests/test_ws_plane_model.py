"""conv_ws.hip's index algebra replayed on the CPU (tools/ws_plane_model.py): DMA lane -> LDS plane rows (swizzle key carried by the lane),
the zero planes, every fragment read of the K loop against the pixel a 3x3 SAME convolution must multiply there (image edges, feature-axis
padding, both K halves), the write-out's staging swizzle / row order / mask-row DMA / pool windows, and the workgroup -> tiles map."""
import itertools
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ws_plane_model as wm  # noqa: E402


@pytest.mark.parametrize("H,NC,K", wm.INSTANCES)
def test_instance_geometry_reads_and_write_out(H, NC, K):
    for Nb, W in ((2, 2 * NC), (3, 3 * NC), (1, NC), (2, 8 * NC)):
        assert wm.check_instance(H, NC, K, Nb, W)


def test_every_tile_is_covered_once_by_the_grid():
    for mt, nt, cus in ((512, 2, 256), (1024, 1, 256), (256, 4, 256), (32, 2, 256), (7, 3, 256), (1000, 3, 304), (1, 1, 256), (300, 1, 256)):
        wg = wm.grid_map(mt, nt, cus)
        assert sorted((n, t) for n, ts in wg for t in ts) == sorted(itertools.product(range(nt), range(mt))), (mt, nt, cus)
        assert len(wg) <= max(cus, nt) and all(ts for _, ts in wg)          # no workgroup without a tile


def test_model_mirrors_the_kernel_source():
    """The instances and the LDS budget the model checks are the ones the kernel file instantiates."""
    src = open(os.path.join(ROOT, 'lstm_ctc_ocr_amd', 'csrc', 'conv_ws.hip')).read()
    inst = sorted(set(tuple(int(v) for v in m) for m in re.findall(r'launch_ws<(\d+), (\d+), (\d+)>', src)))
    assert inst == sorted(wm.INSTANCES)
    for H, NC, K in wm.INSTANCES:
        g = wm.Cfg(H, NC, K)
        assert g.LDS_MASK <= 160 * 1024 and g.PI in (9, 10)
    for expr in ('static constexpr int PS = NC + 2;', 'static constexpr int CHB = (H + 2) * PS * 128;', 'static constexpr int PPC = H * PS / 8;',
                 "(((lane & 7) ^ (cp & 7)) << 3)", "(((j * 4 + fq) ^ (cp & 7)) << 4)", "((slot ^ ((colf & 3) << 2)) << 3)", "((u ^ ((col & 3) << 1)) << 4)"):
        assert expr in src, expr
