"""Which convolution kernel the dispatcher takes for a shape — a host-only query of libocrhip.so (ocr_conv3x3_kernel_choice: the real
dispatch code with the launch cut off), so the policy that the round's measurements settled is pinned without a GPU: the headline net's
ten launches (profiles/r03_final_kernel_stats.md), the variable-width and configs[4] shapes, and the refusals."""
import os

import pytest

from lstm_ctc_ocr_amd import _native as nat
from lstm_ctc_ocr_amd import ops

pytestmark = pytest.mark.skipif(not os.path.exists(nat.LIB_PATH), reason="libocrhip.so not built")

HEADLINE = [   # (layer, Nb, W, H, Cin, Cout, kwargs) -> kernel        (batch 64, 32 x 256 images: LSTM_train.py:22-38)
    ("conv2 forward + pool2", 64, 128, 16, 64, 128, dict(pool=(2, 2)), "conv_ws"),            # round 5: weights in registers, 4 tiles per workgroup (40.7 -> 29.7 us)
    ("conv2 data gradient", 64, 128, 16, 128, 64, dict(bias=False, relu=False, mask=True), "conv_k3/D"),
    ("conv3_1 forward", 64, 64, 8, 128, 256, {}, "conv_k3/A"),
    ("conv3_1 data gradient", 64, 64, 8, 256, 128, dict(bias=False, relu=False, mask=True), "conv_k3/D"),   # A would give 128 tiles, D 256
    ("conv3_2 forward + pool3", 64, 64, 8, 256, 256, dict(pool=(1, 2)), "conv_k3/A"),
    ("conv3_2 data gradient", 64, 64, 8, 256, 256, dict(bias=False, relu=False, mask=True), "conv_k3/A"),
    ("conv4_1 forward", 64, 64, 4, 256, 512, dict(relu=False), "conv_k3/A"),
    ("conv4_1 data gradient", 64, 64, 4, 512, 256, dict(bias=False, relu=False, mask=True), "conv_k3/D"),
    ("conv4_2 forward", 64, 64, 4, 512, 512, dict(relu=False), "conv_k3/A"),
    ("conv4_2 data gradient", 64, 64, 4, 512, 512, dict(bias=False, relu=False, mask=True), "conv_k3/A"),
]


@pytest.mark.parametrize("layer,Nb,W,H,Ci,Co,kw,want", HEADLINE)
def test_headline_layers(layer, Nb, W, H, Ci, Co, kw, want):
    assert ops.conv3x3_kernel_choice(Nb, W, H, Ci, Co, **kw) == want, layer


def test_other_workloads_and_refusals():
    c = ops.conv3x3_kernel_choice
    assert c(64, 80, 8, 256, 256) == "conv_k3w/A"            # variable width: 80 columns, 32-column tiles cross image boundaries
    assert c(64, 50, 4, 512, 512) == "conv_k3w/D"            # 200 tiles of 256 x 128 do not fill the chip, 400 of 256 x 64 do
    assert c(32, 64, 2, 512, 512) == "conv_k3w/D"            # configs[4], H = 2: two images per 128-column tile; half-filling tiles accepted
    assert c(32, 64, 4, 256, 256) == "conv_k3/D"             # configs[4] stage 3: 128 tiles
    assert c(32, 128, 16, 64, 64) == "conv_halo"             # configs[4] stage 1: 9 K steps, one tile per workgroup only (conv_ws starts at two)
    assert c(64, 80, 16, 64, 128, pool=(2, 2)) == "conv_ws"  # configs[3]: a padded width that is a multiple of the 16-column tile
    assert c(64, 84, 16, 64, 128, pool=(2, 2)) == "conv_halo"        # ... and one that is not
    assert c(64, 128, 16, 128, 64, bias=False, relu=False, mask=True) == "conv_k3/D"      # conv_ws's K-split instances are measured slower: not chosen
    assert c(33, 62, 8, 256, 128) == "conv_k2/D"             # M % 256 != 0: the flat-pixel kernel (128 tiles of 256 x 64)
    assert c(17, 62, 4, 128, 128) == "conv_halo"             # 34 tiles: conv_halo's 128-pixel workgroups
    assert c(4, 16, 8, 64, 128) == "gemm"                    # fewer than 1024 pixels: generic engines
    assert c(64, 64, 4, 96, 128) == "gemm"                   # C_in % 64 != 0
    assert c(32, 64, 4, 256, 256, bias=False, relu=False, mask=True, accumulate=True) == "conv_k3/D"     # accumulate form exists there
    assert c(64, 64, 8, 256, 256, pool=(2, 1)) == "gemm"     # no fused pool of that window
