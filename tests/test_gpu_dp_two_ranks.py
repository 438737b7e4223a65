"""-m gpu: TWO data-parallel ranks of the real engine on the one GPU a test box has (gloo transport instead of RCCL; everything
else — rank-local data, 1/(N*world) scaling, the two gradient buckets, the side-stream exchange between the backward graphs, the
optimiser on the exchanged gradient — is the product path).  SURVEY 8e."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_pair(tmp_path, overlap):
    port = _free_port()
    outs = [str(tmp_path / ('rank%d_%s.npz' % (r, overlap))) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_two_rank_worker.py'), str(r), str(port), overlap, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [np.load(o) for o in outs]


def test_two_ranks_exchange_the_sum_of_their_local_gradients_and_stay_identical(dev, tmp_path):
    res = {ov: _run_pair(tmp_path, ov) for ov in ('1', '0')}
    for ov, (r0, r1) in res.items():
        # the ranks saw different data ...
        assert float(r0['local_differs']) > 1e-6
        # ... every rank's gradient buffer after the step holds the SUM of the two local gradients (each pre-scaled by 1/(N*world):
        # the global-batch mean gradient), in both buckets (the weight gradients of conv5 / LSTM / FC use fp32 atomics: order noise)
        for r in (r0, r1):
            ex, sm = r['exchanged'], r['sum_local']
            scale = float(np.abs(sm).max())
            lb = int(r['late_begin'])
            assert float(np.abs(ex[:lb] - sm[:lb]).max()) <= 2e-5 * scale, ov          # early bucket
            assert float(np.abs(ex[lb:] - sm[lb:]).max()) <= 2e-5 * scale, ov          # late bucket
        assert np.array_equal(r0['exchanged'], r1['exchanged'])
        # ... and the replicas are bit-identical after three steps
        assert bool(r0['replicas_equal']) and bool(r1['replicas_equal'])
        assert np.array_equal(r0['params'], r1['params'])
        assert np.all(np.isfinite(r0['losses'])) and not np.allclose(r0['losses'], r1['losses'])     # local losses differ
        # a time-out injected on rank 0 ONLY: both replicas dropped that step (nothing moved, NaN reported on both), counted it once, took the next
        # step normally and are still bit-identical (the drop flag rides on the late bucket: engine._publish_guard, ocr_optim_step_guarded2)
        for r in (r0, r1):
            assert bool(r['fault_unchanged']) and np.isnan(float(r['fault_loss'])), ov
            assert int(r['fault_dropped']) == 1 and int(r['fault_timeouts']) == 1, ov
            assert np.isfinite(float(r['post_loss'])) and bool(r['post_changed']) and float(r['post_steps']) == 1.0, ov
            assert bool(r['post_replicas_equal']), ov
    # overlapped two-bucket schedule == plain one-exchange schedule (same sums; atomics order noise only)
    a, b = res['1'][0]['params'], res['0'][0]['params']
    assert float(np.abs(a - b).max()) < 5e-3 and float(np.abs(a - b).mean()) < 2e-5
