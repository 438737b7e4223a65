"""-m gpu: race detector.  The forward pass contains no atomics, so repeated runs on the same batch must give bit-identical
logits and per-sample CTC costs; the backward pass accumulates with fp32 atomics, so its gradients must repeat to rounding.
(A hand-placed `s_waitcnt lgkmcnt` that was two reads short in one template instance of the convolution kernel passed every
parity test and showed up here as ~1 % of repetitions with garbage gradients.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network


@pytest.mark.parametrize("N,W,ragged,reps", [(8, 88, False, 300), (8, 88, True, 300), (64, 256, False, 60)])
def test_repeated_runs_agree(dev, N, W, ragged, reps):
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
    rng = np.random.RandomState(0)
    x = rng.rand(N, W, 32).astype(np.float32)
    sl = rng.randint(W // 8, W // 4, N).astype(np.int32) if ragged else np.full(N, W // 4 - 1, np.int32)
    ll = np.full(N, 4, np.int32)
    labels = rng.randint(1, 63, N * 4).astype(np.int32)
    ref = eng.forward(x, sl).clone()
    for i in range(reps):
        assert torch.equal(eng.forward(x, sl), ref), "forward repetition %d deviates" % i
    sp = eng.plan(N, W)
    eng._bind(sp, x, sl, labels, ll)
    eng._run(sp, 'fb')
    torch.cuda.synchronize()
    costs, grads = sp.costs.clone(), eng.grads.clone()
    assert bool(torch.isfinite(grads).all())
    scale = float(grads.abs().max())
    for i in range(reps):
        eng._bind(sp, x, sl, labels, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        assert torch.equal(sp.costs, costs), "costs of repetition %d deviate" % i
        assert float((eng.grads - grads).abs().max()) < 1e-3 * scale, "gradients of repetition %d deviate" % i
    for word in getattr(sp, 'lstm_sync', ()):             # error word of the persistent LSTM kernels: 1 = a wait timed out (0, or -1 in a
        assert int(word[-1].item()) != 1                   # block the step prepared inside its first kernel: nothing happened)


# ------------------------------------------------------------------------------------------------------------------------------------------
# The persistent LSTM kernels' hand-off ring under the conditions in which it was silently wrong for three rounds: ragged
# tiles (every sequence ends before T), an HBM-copy stream beside the launches, and — new — unit block 0 of every group held back by a
# deterministic number of clocks per iteration (ocr_lstm_seq_test_skew), which turns the one-in-10^4 interleaving into every launch's.
# tools/lstm_tail_race_probe.py is the runner, tools/lstm_ring_model.py the exhaustive CPU model of the same protocol.
# Replaces: the dependent op sequence of /root/reference/lib/networks/network.py:104-109 (nothing to race there).
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

HANDOFF_CASES = [
    # width, short, train, ragged, graph launches beside the copy stream, eager launches per skew
    (88, 1, False, False, 5000, 100), (88, 4, False, False, 5000, 100), (88, 6, False, False, 5000, 100),
    (88, 1, True, False, 5000, 80), (88, 4, True, False, 5000, 80), (88, 6, True, False, 5000, 80),
    (320, 0, True, True, 5000, 25),                                   # one configs[3] plan: per-sample lengths, tiles end at different steps
]


@pytest.mark.parametrize("width,short,train,ragged,reps,skew_reps", HANDOFF_CASES)
def test_lstm_handoff_is_right_for_any_relative_speed_of_the_workgroups(dev, width, short, train, ragged, reps, skew_reps):
    import lstm_tail_race_probe as probe
    last = width // 4 - 2 - short                   # the last active step of the uniform tiles (ragged: some step in the middle of the tails)
    skews = ((24, -1), (64, -1), (64, last), (128, last), (256, last), (48, 0), (96, 0), (64, 3))
    r = probe.run_case(width=width, short=short, train=train, reps=reps, skews=skews, skew_reps=skew_reps, ragged=ragged)
    print('HANDOFF ' + json.dumps(r))
    assert r['launches'] == reps + len(skews) * skew_reps
    assert r['differed'] == 0, r                    # hout (and dz) bit-identical to the first quiet launch, every time
    assert r['expired'] == 0, r                     # no error word ever read 1


def _exp_library():
    from lstm_ctc_ocr_amd import _native as nat
    exp = os.path.join(os.path.dirname(nat.LIB_PATH), 'libocrhip_exp.so')
    if not os.path.exists(exp):
        pytest.skip('experiments flavour not built (make -C lstm_ctc_ocr_amd/csrc EXPERIMENTS=1)')
    import ctypes
    try:
        fn = ctypes.CDLL(exp).ocr_build_id
        fn.restype = ctypes.c_char_p
        have = fn().decode()
    except (OSError, AttributeError):
        have = None
    if have != nat.source_build_id(experiments=True):
        pytest.skip('libocrhip_exp.so is stale (%s, tree %s): rebuild it with make EXPERIMENTS=1' % (have, nat.source_build_id(experiments=True)))
    return exp


@pytest.mark.parametrize("train,short,skews", [
    (False, 1, '64:last,96:last,128:last,192:last,256:last'),           # the live pipeline's case: an expired wait (the refill of slot s - 2 lands first)
    (False, 4, '96:last,128:last,192:last,256:last,384:last'),          # ... or a silently wrong h (the zero payload of free step s + 3 lands first)
    (True, 6, '160:0,224:0,288:0,384:0,96:0,128:0,512:0'),                  # the backward twin: a free HEAD iteration's zero payload taken for a gradient
])
def test_the_handoff_cases_fail_on_the_round4_ring_rule(dev, train, short, skews):
    """The detector detects: the same runner against the experiments library told to follow the rule of rounds 3-4 again
    (OCR_LSTM_RING_RULE=always: every iteration stores its payload and refills slot i - 2, whether or not any row of the tile is still inside its
    sequence — lstm_seq.hip at 7a5d3057), with unit block 0 held back at ONE iteration: the tile's last active step in the forward kernels (the
    others run free through the tail and destroy what it still reads), iteration 0 in the backward kernels (the others run free through the head
    and find its stale zero payload in the slot of their first real poll)."""
    exp = _exp_library()
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'lstm_tail_race_probe.py'), '--width', '88', '--short', str(short), '--reps', '0',
           '--skews', skews, '--skew-reps', '20', '--stop-at-first', '--json'] + (['--train'] if train else [])
    out = subprocess.run(cmd, env=dict(os.environ, OCR_NATIVE_LIB=exp, OCR_LSTM_RING_RULE='always'), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('RESULT ')][-1]
    r = json.loads(line[7:])
    print('OLD-RULE ' + line[7:])
    assert r['build'].endswith('-exp')
    assert r['differed'] + r['expired'] > 0, 'the round-4 ring rule went undetected: %s' % r
    # ... and the same library WITHOUT the switch (the rule the product compiles in) passes the same launches
    cmd[cmd.index('--stop-at-first')] = '--json'
    out = subprocess.run(cmd, env=dict(os.environ, OCR_NATIVE_LIB=exp), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    print('NEW-RULE ' + json.dumps(r))
    assert r['differed'] == 0 and r['expired'] == 0, r
