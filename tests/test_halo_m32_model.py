"""CPU: the index model of the experimental 32x32x16-MFMA convolution kernel (tools/halo_m32_model.py) — swizzle, fragment
addresses, MFMA operand / result layout and epilogue mapping replayed lane by lane against a direct 3x3 SAME convolution.
-m gpu (only with OCR_TEST_EXPERIMENTAL=1): the kernel itself behind OCR_HALO_MFMA32=1 through the ordinary convolution parity tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import halo_m32_model as model          # noqa: E402


def test_row_swizzle_is_bank_conflict_free_for_every_tap_shift():
    assert model.max_bank_conflict() == 1


@pytest.mark.parametrize("kw", [dict(Nb=2, cW=20, cH=4, C=64, N=128, BN=128, NW=4, seed=0),            # two pixel tiles, ragged tail
                                dict(Nb=1, cW=12, cH=8, C=128, N=64, BN=64, NW=4, seed=1),             # two channel chunks, one fragment per wave
                                dict(Nb=1, cW=6, cH=16, C=64, N=72, BN=128, NW=4, seed=2),             # H = 16, channel tail (N % 64 != 0)
                                dict(Nb=3, cW=22, cH=4, C=64, N=128, BN=128, NW=8, seed=3, tiles=[(0, 0), (1, 0)])])   # 8-wave tiles
def test_lane_level_replay_equals_direct_convolution(kw):
    assert model.check(**kw) > 0


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get('OCR_TEST_EXPERIMENTAL') != '1', reason='experimental kernel: set OCR_TEST_EXPERIMENTAL=1')
def test_mfma32_kernel_through_the_convolution_parity_tests(dev):
    env = dict(os.environ, OCR_HALO_MFMA32='1')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_kernels.py'), '-q', '-k', 'conv3x3'],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]
