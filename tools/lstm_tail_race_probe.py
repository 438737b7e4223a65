"""Reproducer for the persistent LSTM kernels' hand-off races (round 5: once every row of a batch tile is past its length, a workgroup of the
forward recurrence polls nothing and runs free; under the rule of rounds 3-4 its slot refill and, three steps later, its payload landed in ring
slots that a slower workgroup, still at the tile's last active step, was reading -> a time-out (refill) or a silently wrong h (payload); the
backward kernels have the same hole at the HEAD of a tile whose sequences are all shorter than T).  tools/lstm_ring_model.py is the CPU model.

run_case(): the same launch (inference forward, or forward + CTC + backward) is repeated on one plan
  * `reps` times as the captured graph beside a second stream that keeps HBM busy (the condition under which the live pipeline hit the race:
    ~2 in 20 000 launches on the old rule — profiles/r05p_race.log), and
  * `skew_reps` times per entry (n, at) of `skews` eagerly with ocr_lstm_seq_test_skew(n, at): unit block 0 of every group sleeps n x 64 clocks
    at iteration `at` (-1: at every iteration), which makes the interleaving the race needs DETERMINISTIC (a correct hand-off is right for any
    relative speed; the delay at ONE iteration — the tile's last active step forward, iteration 0 backward — is the one that lets the other
    workgroups run ahead: a delay in every iteration is absorbed by their adaptive pre-poll sleep),
and every launch's BiLSTM output (hout; with --train also the backward recurrence's dz) is compared bit for bit with the first quiet launch.
tests/test_gpu_stress.py runs these cases in `pytest -m gpu`, and once against the experiments library with OCR_LSTM_RING_RULE=always (the
round-4 rule), where they must FAIL.

    python tools/lstm_tail_race_probe.py [--reps 20000] [--short 4] [--width 88] [--train] [--skews 24:-1,96:last,64:0] [--skew-reps 300] [--ragged]"""
import argparse
import json
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class HbmNoise:
    """a side stream that copies 256 MB blocks back and forth while the block is open"""
    def __init__(self, dev):
        self.stop = threading.Event()
        self.side = torch.cuda.Stream(device=dev)
        self.a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        self.b = torch.empty_like(self.a)

    def _loop(self):
        with torch.cuda.stream(self.side):
            n = 0
            while not self.stop.is_set():
                self.b.copy_(self.a, non_blocking=True)
                n += 1
                if n % 8 == 0:
                    self.side.synchronize()

    def __enter__(self):
        self.th = threading.Thread(target=self._loop)
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join()
        self.side.synchronize()


def run_case(width=88, short=4, train=False, reps=5000, skews=((24, -1), (64, -1)), skew_reps=300, ragged=False, N=64, stop_at_first=False, seed=0):
    """Returns dict(launches, differed, expired, ...).  ragged: per-sample lengths as a configs[3] batch has them (W_i in [80, width], so every
    16-row tile ends before T and the tiles end at different steps); else every sequence `short` steps shorter than T."""
    from lstm_ctc_ocr_amd import _native as nat
    from lstm_ctc_ocr_amd import ops
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    dev = torch.device('cuda:0')
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
    T = width // 4 - 1
    rng = np.random.RandomState(seed)
    x = torch.from_numpy(rng.rand(N, width, 32).astype(np.float32)).to(dev)
    if ragged:
        wi = rng.randint(80, width - 8, N)
        sl_host = (wi // 4 - 1).astype(np.int32)
        cols = torch.arange(width, device=dev)[None, :, None]
        x = x * (cols < torch.from_numpy(wi).to(dev)[:, None, None])        # right-padded with 0 like gen.py:59-65
    else:
        sl_host = np.full(N, T - short, np.int32)
    sl = torch.from_numpy(sl_host).to(dev)
    sp = eng.plan(N, width)
    mode = 'fb' if train else 'fwd'
    if train:
        ll = rng.randint(3, 5, N).astype(np.int32)
        labels = rng.randint(1, 63, int(ll.sum())).astype(np.int32)
        eng._bind(sp, x, sl, labels, ll)
    else:
        eng._bind(sp, x, sl)
    ops.lstm_seq_test_skew(0, -1)
    eng._run(sp, mode)                                   # (captures the graph: the skew is baked in at capture — 0 here)
    torch.cuda.synchronize()
    keys = [k for k in sp.buf if k.endswith('/hout')][:1] + ([k for k in sp.buf if k.endswith('/dz') and 'logits' in k] if train else [])
    refs = [sp.buf[k].clone() for k in keys]
    for _ in range(2):                                   # the reference itself: three quiet launches agree
        eng._run(sp, mode)
        torch.cuda.synchronize()
        assert all(torch.equal(sp.buf[k], r) for k, r in zip(keys, refs)), 'quiet launches disagree'
    diff = torch.zeros(len(keys), dtype=torch.int64, device=dev)       # launches in which output k differed (hout | dz)
    res = dict(build=nat.build_id(), width=width, T=T, mode=mode, lens='ragged' if ragged else 'T-%d' % short, launches=0, differed=0, expired=0,
               by_skew={})
    exp_words = [0] * len(sp.lstm_sync)                  # expired waits seen per error word (forward launch | backward launch)

    def expired():
        n = 0
        for i, w in enumerate(sp.lstm_sync):
            if int(w[-1].item()) == 1:
                exp_words[i] += 1
                n += 1
        return n

    def launch_and_compare():
        eng._run(sp, mode)
        diff.add_(torch.stack([(sp.buf[k] != r).any() for k, r in zip(keys, refs)]).to(torch.int64))

    def ndiff():
        return int(diff.max().item())

    with HbmNoise(dev):
        i = -1
        for i in range(reps):
            launch_and_compare()
            if i % 500 == 499 or i == reps - 1:
                torch.cuda.synchronize()
                res['expired'] += expired()
                if stop_at_first and (ndiff() or res['expired']):
                    break
        res['launches'] += i + 1
        res['by_skew']['0'] = [ndiff(), res['expired']]
        graphs = eng.use_graphs
        eng.use_graphs = False                           # eager: the skew is a launch argument
        try:
            for sk, at in skews:
                if at == 'last':                         # the last active step of the (uniform-length) tiles
                    at = T - short - 1
                d0, e0 = ndiff(), res['expired']
                ops.lstm_seq_test_skew(sk, at)
                for i in range(skew_reps):
                    launch_and_compare()
                    if i % 10 == 9 or i == skew_reps - 1:
                        torch.cuda.synchronize()
                        res['expired'] += expired()
                        if stop_at_first and (ndiff() > d0 or res['expired'] > e0):
                            break
                res['launches'] += i + 1
                res['by_skew']['%d@%d' % (sk, at)] = [ndiff() - d0, res['expired'] - e0]
                if stop_at_first and (ndiff() or res['expired']):
                    break
        finally:
            ops.lstm_seq_test_skew(0, -1)
            eng.use_graphs = graphs
    torch.cuda.synchronize()
    res['differed'] = ndiff()
    res['differed_by_output'] = {k.split('/')[-1]: int(v) for k, v in zip(keys, diff.tolist())}
    res['expired_by_launch'] = dict(zip(('forward', 'backward'), exp_words))
    return res


def parse_skews(text):
    out = []
    for item in text.split(','):
        if item:
            u, _, at = item.partition(':')
            out.append((int(u), at if at == 'last' else int(at or -1)))
    return tuple(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20000)
    ap.add_argument('--short', type=int, default=4)
    ap.add_argument('--width', type=int, default=88)
    ap.add_argument('--train', action='store_true', help='forward + CTC + backward: compare the backward recurrence\'s dz too')
    ap.add_argument('--ragged', action='store_true', help='per-sample lengths of a configs[3] batch instead of T - short everywhere')
    ap.add_argument('--skews', default='24:-1,64:-1,96:last,64:0', help='units:iteration pairs; iteration -1 = every, last = T - short - 1')
    ap.add_argument('--skew-reps', type=int, default=300)
    ap.add_argument('--stop-at-first', action='store_true')
    ap.add_argument('--json', action='store_true')
    a = ap.parse_args()
    skews = parse_skews(a.skews)
    r = run_case(a.width, a.short, a.train, a.reps, skews, a.skew_reps, a.ragged, stop_at_first=a.stop_at_first)
    if a.json:
        print('RESULT ' + json.dumps(r), flush=True)
    else:
        print('build %s: W=%d T=%d len=%s, %s: %d launches (%d as a graph beside an HBM-copy stream, then %s x <= %d eager with unit block 0 held back, units@iteration): '
              'output differed from the first quiet launch in %d, expired waits seen at %d checks; per skew [differed, expired]: %s'
              % (r['build'], r['width'], r['T'], r['lens'], 'forward + CTC + backward (hout and dz compared)' if a.train else 'inference forward (hout compared)',
                 r['launches'], a.reps, list(skews), a.skew_reps, r['differed'], r['expired'], r['by_skew']), flush=True)


if __name__ == '__main__':
    main()
