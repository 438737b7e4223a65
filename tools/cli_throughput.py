"""End-to-end training-loop throughput through the REAL input path (live captcha generator -> shared-memory ring -> pinned async
H2D -> Engine.train_step with the loss fetched every iteration, exactly what lib/lstm/train.py's loop does) next to the
device-resident rate of the same shape.   python tools/cli_throughput.py [--iters 2000] [--legacy]   (on the GPU box)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd.config import cfg, cfg_from_file  # noqa: E402
from lstm_ctc_ocr_amd.engine import Engine  # noqa: E402
from lstm_ctc_ocr_amd.models import get_network  # noqa: E402


LAG = True


def run(eng, stream, iters, warm):
    it = iter(stream)
    losses = []
    for i in range(warm):
        b = next(it)
        eng.train_step(*(b if torch.is_tensor(b[0]) else [np.array(a) for a in b]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    pending = None
    for i in range(iters):
        b = next(it)
        x = b[0] if torch.is_tensor(b[0]) else np.array(b[0])
        rest = b[1:] if torch.is_tensor(b[0]) else [np.array(a) for a in b[1:]]
        if LAG:                                                     # what lstm_ctc_ocr_amd/train.py does by default: the loss of
            eng.train_step(x, *rest, fetch_loss=False)              # iteration k is read while iteration k + 1 runs
            h = eng.report_async()
            if pending is not None:
                try:
                    losses.append(eng.report_wait(pending))
                except Exception as e:                              # diagnostic (round 5): what does the state look like when a time-out is reported?
                    torch.cuda.synchronize()
                    sp = eng.last_plan
                    fin = lambda t: bool(torch.isfinite(t.float()).all())
                    bufs = {k: fin(v) for k, v in sp.buf.items() if torch.is_tensor(v) and v.is_floating_point() and k.endswith(('/y', '/hout', '/z', '/dy'))}
                    print('TIMEOUT-DIAG iteration %d: last losses %s | params finite %s grads finite %s | non-finite activation buffers %s | seq_len %s..%s labels_len %s..%s | %s'
                          % (i, [round(v, 3) for v in losses[-5:]], fin(eng.params), fin(eng.grads), [k for k, ok in bufs.items() if not ok][:8],
                             int(sp.seq_len.min()), int(sp.seq_len.max()), int(sp.labels_len.min()), int(sp.labels_len.max()), str(e)[:100]), flush=True)
                    raise
            pending = h
        else:
            losses.append(eng.train_step(x, *rest))                 # one host sync per iteration at once (sess.run semantics)
        n += x.shape[0]
    if pending is not None:
        losses.append(eng.report_wait(pending))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dropped = [k for k, v in enumerate(losses) if v != v]
    if dropped:
        print('steps dropped by the time-out guard at iterations %s' % dropped[:8], flush=True)
    return n / dt, dt / iters * 1e3, float(np.nanmean(losses[-50:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2000)
    ap.add_argument('--legacy', action='store_true')
    ap.add_argument('--synth', action='store_true', help='GPU-side synthesis (utils/synth.DeviceSynthStream) instead of the PIL worker ring')
    ap.add_argument('--workers', type=int, default=0)
    ap.add_argument('--pool', type=int, default=0)
    ap.add_argument('--only', default='', help='run one configuration only: W88, W256 or var')
    ap.add_argument('--var', action='store_true', help='also run variable-width batches (3-12 characters at 48 px each: W = 76..312 after the resize, '
                                                      'padded per batch - BASELINE configs[3] as the generators produce it)')
    ap.add_argument('--no-lag', action='store_true', help='wait for every loss at once (OCR_LOSS_LAG=0 of the training loop)')
    a = ap.parse_args()
    global LAG
    LAG = not a.no_lag
    cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lstm', 'lstm.yml'))
    out = {'source': 'gpu synthesis' if a.synth else 'PIL workers', 'cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0)), 'pool': a.pool, 'loss_lag': LAG}
    try:
        out['cgroup_cpu_max'] = open('/sys/fs/cgroup/cpu.max').read().strip()
    except Exception:
        pass
    confs = [('W88_4to6char', dict()), ('W256_10char', dict(min_len=10, max_len=10, width=480))]
    if a.var or a.only == 'var':
        confs.append(('varwidth_3to12char', dict(min_len=3, max_len=12, px_per_char=48)))
    for name, kw in confs:
        if a.only and not name.startswith(a.only):
            continue
        eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
        eng.setup_optimizer('Adam', 1e-4)
        if a.legacy:
            from lstm_ctc_ocr_amd.utils.gen import get_batch
            stream = get_batch(num_workers=12, batch_size=64, vis=False, **kw)
        elif a.synth:
            from lstm_ctc_ocr_amd.utils.synth import DeviceSynthStream
            stream = DeviceSynthStream('cuda:0', 64, **kw)
        else:
            from lstm_ctc_ocr_amd.utils.pipeline import DeviceBatchStream
            stream = DeviceBatchStream('cuda:0', 64, workers=a.workers or None, pool=a.pool, **kw)
            time.sleep(3.0)                                          # let the ring fill
        nworkers = len(stream.ring.procs) if hasattr(stream, 'ring') else (0 if a.synth else 12)
        ips, ms, loss = run(eng, stream, a.iters, 50)
        # device-resident rate of the same shape: replay ONE batch that is already in HBM
        b = next(iter(stream))
        if not torch.is_tensor(b[0]):
            b = [torch.from_numpy(np.array(b[0], np.float32)).cuda()] + [torch.from_numpy(np.array(x, np.int32)).cuda() for x in b[1:]]
        else:
            b = [t.clone() for t in b]
        for _ in range(20):
            eng.train_step(*b, fetch_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            eng.train_step(*b, fetch_loss=False)
        torch.cuda.synchronize()
        dev = 300 * 64 / (time.perf_counter() - t0)
        if hasattr(stream, 'close'):
            stream.close()
        out[name] = {'pipeline_images_per_s': ips, 'ms_per_iter': ms, 'device_resident_images_per_s': dev, 'ratio': ips / dev,
                     'loss_tail': loss, 'workers': nworkers,
                     'pinned': getattr(stream, 'pinned', None)}
        print(name, json.dumps(out[name]), flush=True)
        del eng
    print('RESULT ' + json.dumps(out))


if __name__ == '__main__':
    main()
