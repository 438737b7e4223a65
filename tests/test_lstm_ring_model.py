"""The persistent LSTM kernels' hand-off ring (lstm_seq.hip, protocol 4) as a CPU model, explored exhaustively over interleavings of
{sub-poll passes, payload / refill store parts land}: tools/lstm_ring_model.py.  Three workgroups x four slots x T <= 6 x every assignment of
sequence lengths 0 .. T to two rows of the batch tile — all-inactive TAILS in the forward kernels, all-inactive HEADS in the backward ones.
The rule the kernels follow since round 5 ('live': a workgroup none of whose rows is inside its sequence leaves the ring alone) never lets a
piece be overwritten before its last reader, never lets a reader take a stale payload, never leaves a reader waiting; the rule of rounds 3-4
('always') fails exactly where the hardware did (profiles/r05p_race.log: expired waits and silently wrong h on tiles whose sequences end
before T) — and, in the backward kernels, where it was never caught on hardware (a free head iteration's zero payload taken for a gradient).
Replaces: the tf.while_loop of /root/reference/lib/networks/network.py:104-109 (no hand-off there: one kernel per op per step)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import lstm_ring_model as rm  # noqa: E402


@pytest.mark.parametrize("direction,T,waves", [('fwd', 4, 1), ('fwd', 6, 1), ('bwd', 6, 1), ('fwd', 4, 4), ('bwd', 4, 4)])
def test_live_rule_is_clean_for_every_length_pattern(direction, T, waves):
    npat, states, found = rm.sweep('live', direction, T, W=3, R=2, waves=waves)
    assert npat == (T + 1) ** 2 and states > 10 * npat
    assert not found, found


def test_live_rule_is_clean_for_the_four_wave_kernels_at_six_steps_on_the_ragged_patterns():
    # the patterns with a free stretch and a ring wrap (heads in the backward kernels need T - max(len) >= 1 and the slot of iteration 0 to come round)
    for direction in ('fwd', 'bwd'):
        for lens in ((0, 2), (2, 2), (1, 2), (2, 3), (1, 5), (3, 3)):
            assert rm.explore('live', direction, 6, lens, W=3, waves=4) > 0


@pytest.mark.parametrize("waves", [1, 4])
def test_round4_rule_fails_in_the_forward_kernels(waves):
    """free-running workgroups of a finished tile refill / overwrite slots a slower workgroup still reads"""
    npat, _, found = rm.sweep('always', 'fwd', 4, W=3, R=2, waves=waves)
    assert 'overwrite' in found and found['overwrite'][1] >= 10, found
    # the live pipeline's case: every sequence one step shorter than T (W = 88: 20 of 21 steps)
    for T in (4, 5, 6):
        with pytest.raises(rm.Violation):
            rm.explore('always', 'fwd', T, (T - 1, T - 1), W=3, waves=waves)
        assert rm.explore('live', 'fwd', T, (T - 1, T - 1), W=3, waves=waves) > 0
    # full-length tiles never had the problem under either rule
    assert rm.explore('always', 'fwd', 6, (6, 6), W=3, waves=waves) > 0


@pytest.mark.parametrize("waves", [1, 4])
def test_round4_rule_fails_in_the_backward_kernels_once_the_ring_wraps(waves):
    """a late workgroup's zero payload of free head iteration 0 sits in slot 0 until its refill at iteration 2; a faster workgroup polls slot 0 at
    iteration 5 for the payload of iteration 4 and takes the zeros (needs T - max(len) >= 4: never seen on hardware, closed in 9fec851)"""
    with pytest.raises(rm.Violation) as e:
        rm.explore('always', 'bwd', 6, (2, 2), W=3, waves=waves)
    assert e.value.kind == 'stale'
    assert rm.explore('live', 'bwd', 6, (2, 2), W=3, waves=waves) > 0
    for T in (4, 5):                                       # shorter than the wrap: the old rule was safe there
        assert not rm.sweep('always', 'bwd', T, W=3, R=2, waves=1)[2]


def test_model_mirrors_the_kernel_source():
    src = open(os.path.join(ROOT, 'lstm_ctc_ocr_amd', 'csrc', 'lstm_seq.hip')).read()
    assert '#define RING %d' % rm.RING in src
    # every protocol-4 kernel guards BOTH ring stores with the tile-wide activity: 5 kernels (fwd_seq, bwd_seq, fwd_seq4, fwd_seq4x, bwd_seq4)
    assert src.count('const bool ring_live = RING_LIVE(active);') == 5
    # ... and the product library compiles the 'live' rule in: the only other definition sits behind OCR_EXPERIMENTS (tests/test_gpu_stress.py shows that one failing)
    assert '#else\n#define RING_LIVE(act) (__any(act))\n#endif' in src and '#ifdef OCR_EXPERIMENTS\n#define RING_LIVE(act) (__any(act) || a.ring_always)' in src
    assert src.count('>= 2 && ring_live)') == 5
    # the polls wait for rows that need the hand-off only: `active` forward, `has_next` backward
    assert src.count('__any(active && holds_fill(m))') == 3 and src.count('__any(has_next && holds_fill(m))') == 2
