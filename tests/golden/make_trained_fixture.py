"""Trained-weights parity fixture: rendered captcha batches + a trained CRNN + the fp32 oracle's outputs on them.

The reference holds no golden vectors and TensorFlow cannot run here, so the end-to-end parity anchor is:
the SAME trained weights and the SAME PIL-rendered captcha batches go through (a) the fp32 CPU oracle (oracle/graph.py, no
bf16 rounding anywhere) and (b) the MI355X path; tests/test_trained_fixture.py compares logits, per-sample CTC costs, the
loss, greedy strings (blank 0) and the reference's beam-search strings (beam 100, blank C-1, zeros stripped).

Three stages (each one a sub-command; the files they write are committed):

  batches   CPU, this container.  Renders the batches with lstm_ctc_ocr_amd.utils.gen (PIL) -> tests/golden/captcha_batches.npz
            C1 : N = 8,  4 characters on the stock 160x60 canvas            -> W = 88,  T = 21 (20 valid)   BASELINE configs[0]
            C2 : N = 64, 10 characters on 480x60                            -> W = 256, T = 63              BASELINE configs[1]
            V0..V2 : N = 64, 3..12 / 3..9 / 3..6 characters, 50 px per char -> W = 320 / ~240 / ~160, ragged  BASELINE configs[3]
                 (the shortest sample of V0 carries an INFEASIBLE label: more characters than time steps -> cost 0, gradient 0)
  train     GPU box (gpurun).  Trains LSTM_train with the product's own engine on the live generator (three streams: stock,
            10-character, variable width), rounds the trained weights to bf16-representable values (the device's MFMA operand
            precision — the fp32 oracle then sees exactly the weights the device multiplies with) and writes
            gpurun_out/fixture/trained_weights.npz (+ the device's own outputs on the batches, for the record).
  oracle    CPU, this container.  fp32 oracle on (weights, batches) -> tests/golden/trained_expect.npz

    python tests/golden/make_trained_fixture.py batches
    gpurun -- python tests/golden/make_trained_fixture.py train --iters 24000
    cp gpurun_out/fixture/trained_weights.npz tests/golden/ && python tests/golden/make_trained_fixture.py oracle
"""
import argparse
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)

BATCHES = os.path.join(HERE, 'captcha_batches.npz')
WEIGHTS = os.path.join(HERE, 'trained_weights.npz')
EXPECT = os.path.join(HERE, 'trained_expect.npz')
NAMES = ('C1', 'C2', 'V0', 'V1', 'V2')
VARIANTS = {           # name: (N, generator kwargs)
    'C1': (8, dict(min_len=4, max_len=4, width=160)),
    'C2': (64, dict(min_len=10, max_len=10, width=480)),
    'V0': (64, dict(min_len=3, max_len=12, px_per_char=50)),
    'V1': (64, dict(min_len=3, max_len=9, px_per_char=50)),
    'V2': (64, dict(min_len=3, max_len=6, px_per_char=50)),
}


# ------------------------------------------------------------------------------------------------ bf16 helpers (numpy)
def to_bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7fff + ((u >> 16) & 1)                       # round to nearest even
    return (u >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def load_weights(path=WEIGHTS):
    d = np.load(path)
    return {k[2:]: from_bf16_bits(d[k]).reshape(d['s/' + k[2:]]) for k in d.files if k.startswith('w/')}


def load_batch(d, name):
    """-> x float32 [N, W, 32] in [0, 1] (gen.py:59-65), flat labels, label lengths, time_step_len"""
    x = d[name + '/x_u8'].astype(np.float32) / 255.
    return x, d[name + '/labels'], d[name + '/label_len'], d[name + '/seq_len']


# ------------------------------------------------------------------------------------------------ stage 1
def stage_batches():
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.utils import gen
    out = {}
    for i, name in enumerate(NAMES):
        n, kw = VARIANTS[name]
        random.seed(20260925 + i)
        np.random.seed(20260925 + i)
        imgs, labels, label_len, steps = next(gen.generator(batch_size=n, **kw))
        x = np.stack(imgs)                                          # [N, W, 32] float32 = uint8 / 255
        u8 = np.rint(x * 255.).astype(np.uint8)
        assert np.array_equal(u8.astype(np.float32) / 255., x)
        labels, label_len, steps = np.array(labels, np.int32), np.array(label_len, np.int32), np.array(steps, np.int32)
        if name == 'V0':                # one INFEASIBLE sample: 31 characters on the shortest image (warp-ctc: cost 0, zero gradient)
            k = int(np.argmin(steps))
            assert steps[k] < 31
            pos = int(label_len[:k].sum())
            extra = np.array([(7 * j) % 62 + 1 for j in range(31)], np.int32)
            labels = np.concatenate([labels[:pos], extra, labels[pos + label_len[k]:]])
            label_len[k] = 31
        out.update({name + '/x_u8': u8, name + '/labels': labels, name + '/label_len': label_len, name + '/seq_len': steps})
        print('%s: x %s, %d labels, T valid %d..%d, font %s' % (name, u8.shape, len(labels), steps.min(), steps.max(),
                                                                  os.path.basename(gen.resolve_font())))
    np.savez_compressed(BATCHES, **out)
    print('wrote', BATCHES, os.path.getsize(BATCHES), 'bytes')


# ------------------------------------------------------------------------------------------------ stage 2 (GPU)
def stage_train(iters, out_dir, workers):
    import torch
    from lstm_ctc_ocr_amd.config import cfg, cfg_from_file
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    from lstm_ctc_ocr_amd.utils.gen import get_batch
    from lstm_ctc_ocr_amd.utils.training import accuracy_calculation
    cfg_from_file(os.path.join(ROOT, 'lstm', 'lstm.yml'))
    os.makedirs(out_dir, exist_ok=True)
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=cfg.RNG_SEED, max_label_len=31)
    eng.setup_optimizer('Adam', 2e-4)
    workers = workers or max(4, min(48, (os.cpu_count() or 8) - 4))
    share = max(2, workers // 3)
    streams = [get_batch(num_workers=share, seed=1000 * (i + 1), batch_size=64, **kw)
               for i, kw in enumerate((dict(), dict(min_len=10, max_len=10, width=480), dict(min_len=3, max_len=12, px_per_char=50)))]
    d = np.load(BATCHES)
    t0 = time.time()
    for it in range(iters):
        if it in (iters // 2, 3 * iters // 4):
            eng.scale_lr(0.5)
        images, labels, label_lens, steps = next(streams[it % 3])
        loss = eng.train_step(np.array(images), np.array(labels), np.array(label_lens), np.array(steps), fetch_loss=(it % 200 == 0))
        if it % 1000 == 0:
            print('iter %d loss %.4f lr %.2e  %.1f s' % (it, eng.last_loss(), eng.lr, time.time() - t0), flush=True)
    print('trained %d iterations in %.1f s' % (iters, time.time() - t0), flush=True)
    # bf16-representable weights: what the MFMA operands hold; reload them so device and oracle see the same numbers
    arrays = eng.state_arrays()
    bits = {k: to_bf16_bits(v) for k, v in arrays.items()}
    eng.load_arrays({k: from_bf16_bits(b).reshape(arrays[k].shape) for k, b in bits.items()})
    save = {}
    for k, b in bits.items():
        save['w/' + k] = b.reshape(-1)
        save['s/' + k] = np.array(arrays[k].shape, np.int64)
    np.savez_compressed(os.path.join(out_dir, 'trained_weights.npz'), **save)
    dev = {}
    for name in NAMES:
        x, labels, ll, sl = load_batch(d, name)
        truth, pos = [], 0
        for n in ll:
            truth.append(labels[pos:pos + n].tolist()); pos += n
        logits = eng.forward(x, sl).float().cpu().numpy()
        greedy = eng.decode(x, sl, method='greedy')
        beam = eng.decode(x, sl, method='beam')
        sp = eng.plan(x.shape[0], x.shape[1])
        eng._bind(sp, x, sl, labels, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        costs = sp.costs.cpu().numpy()
        acc_g = accuracy_calculation(truth, greedy, isPrint=False)
        acc_b = accuracy_calculation(truth, beam, isPrint=False)
        print('%s: device accuracy greedy %.3f beam %.3f, mean cost %.5f, greedy != beam on %d samples'
              % (name, acc_g, acc_b, float(costs.mean()), sum(a != b for a, b in zip(greedy, beam))), flush=True)
        dev[name + '/logits'] = logits
        dev[name + '/costs'] = costs
        dev[name + '/greedy'] = np.array([g + [0] * (40 - len(g)) for g in greedy], np.int32)
        dev[name + '/beam'] = np.array([g + [0] * (40 - len(g)) for g in beam], np.int32)
    np.savez_compressed(os.path.join(out_dir, 'device_outputs.npz'), **dev)
    print('wrote', out_dir, flush=True)


# ------------------------------------------------------------------------------------------------ stage 3
def oracle_outputs(params, d, name, have=None):
    """fp32 oracle (the parity anchor) and the bf16-simulating oracle (rounds where the device stores bf16) on one batch.
    `have`: outputs of an earlier run — the (slow, pure-Python) beam search is skipped when the fp32 logits are unchanged."""
    import torch
    from oracle import decode as odec
    from oracle import graph as og
    x, labels, ll, sl = load_batch(d, name)
    out = {}
    with torch.no_grad():
        for tag, sim in (('', False), ('_bf16sim', True)):
            logits = og.forward(params, torch.from_numpy(x), sl.tolist(), sim_bf16=sim)
            out['logits' + tag] = logits.numpy().astype(np.float32)
            out['costs' + tag] = og._CTC.apply(logits, labels, ll, sl).numpy().astype(np.float64)
    lg = out['logits']
    pad = lambda seqs: np.array([s + [0] * (40 - len(s)) for s in seqs], np.int32)
    out['greedy'] = pad([[v for v in s if v != 0] for s in odec.greedy_decode(lg, sl)])
    if have is not None and np.array_equal(have.get('logits'), lg):
        out['beam'] = have['beam']
    else:
        out['beam'] = pad([[v for v in s if v != 0] for s in odec.reference_decode(lg, sl, beam_width=100)])
    return out


def stage_oracle():
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    params = {k: torch.from_numpy(v.copy()) for k, v in load_weights().items()}
    d = np.load(BATCHES)
    old = np.load(EXPECT) if os.path.exists(EXPECT) else None
    out = {}
    for name in NAMES:
        t0 = time.time()
        have = {k: old[name + '/' + k] for k in ('logits', 'beam')} if old is not None and name + '/logits' in old.files else None
        r = oracle_outputs(params, d, name, have)
        truth, pos = [], 0
        for n in d[name + '/label_len']:
            truth.append(d[name + '/labels'][pos:pos + n].tolist()); pos += n
        strip = lambda row: [int(v) for v in row if v != 0]
        acc = np.mean([strip(g) == t for g, t in zip(r['greedy'], truth)])
        dis = sum(strip(g) != strip(b) for g, b in zip(r['greedy'], r['beam']))
        print('%s: oracle fp32 mean cost %.6f (bf16-sim %.6f), greedy accuracy %.3f, greedy != beam on %d of %d samples (%.1f s)'
              % (name, r['costs'].mean(), r['costs_bf16sim'].mean(), acc, dis, len(truth), time.time() - t0))
        out.update({name + '/' + k: v for k, v in r.items()})
    np.savez_compressed(EXPECT, **out)
    print('wrote', EXPECT, os.path.getsize(EXPECT), 'bytes')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('stage', choices=['batches', 'train', 'oracle'])
    ap.add_argument('--iters', type=int, default=24000)
    ap.add_argument('--workers', type=int, default=0)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'fixture'))
    a = ap.parse_args()
    if a.stage == 'batches':
        stage_batches()
    elif a.stage == 'train':
        stage_train(a.iters, a.out, a.workers)
    else:
        stage_oracle()
