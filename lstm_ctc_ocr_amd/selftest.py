"""smoke(): one tiny train step + greedy decode of the CRNN hot path on cuda:0, checked against the oracle.
(The oracle import lives here and in tests/ only — it is the checker, never the thing that runs the model.)"""
import numpy as np
import torch


def tiny_batch(N=4, W=32, L=2, seed=0, nclasses=64):
    rng = np.random.RandomState(seed)
    x = rng.rand(N, W, 32).astype(np.float32)
    T = W // 4 - 1
    labels = rng.randint(1, nclasses - 1, size=N * L).astype(np.int32)
    return x, labels, np.full(N, L, np.int32), np.full(N, T, np.int32)


def smoke():
    from .engine import Engine
    from .models import get_network
    from oracle import decode as odec
    from oracle import graph as og

    torch.manual_seed(0)
    net = get_network('LSTM_train')
    eng = Engine(net, device='cuda:0', seed=3)
    x, labels, ll, sl = tiny_batch()
    params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
    # forward + greedy decode vs the oracle with bf16 rounding at the same points
    logits = eng.forward(x, sl).float().cpu()
    ref = og.forward(params, torch.from_numpy(x), sl.tolist(), sim_bf16=True)
    err = float((logits - ref).abs().max())
    assert err < 5e-3, 'logits differ from the oracle by %g' % err
    dec = eng.decode(x, sl, method='greedy')
    assert dec == odec.greedy_decode(logits.numpy(), sl), 'greedy decode differs from the oracle'
    assert eng.decode(x, sl, method='beam') == odec.reference_decode(logits.numpy(), sl), 'beam decode differs from the oracle'
    # one optimisation step: loss against the oracle's loss on the same batch
    wd = float(eng.cfg.TRAIN.WEIGHT_DECAY)
    loss = eng.train_step(x, labels, ll, sl)
    total, ctc, _ = og.loss_fn(params, torch.from_numpy(x), labels, ll, sl.tolist(), wd, sim_bf16=True)
    rel = abs(loss - float(total)) / abs(float(total))
    assert rel < 1e-3, 'loss %g vs oracle %g' % (loss, float(total))
    loss2 = eng.train_step(x, labels, ll, sl)
    assert np.isfinite(loss2)
    # the data side: four captchas composed on the GPU against Pillow driven with the same parameters (everything but the noise arc is Pillow's own
    # arithmetic and must be identical: oracle/synth_ref.py)
    from . import ops
    from .utils import gen, synth
    from oracle import synth_ref
    atlas = synth.GlyphAtlas()
    P = synth.draw_params(np.random.default_rng(3), 4, atlas, strings=False)
    P['packed'][:, 8] = P['packed'][:, 6]                                # empty arc box
    W = gen.padded_width(int(P['nw_out'].max()))
    pix = torch.zeros((4, W, 32), dtype=torch.uint8, device='cuda:0')
    ops.captcha_synth(torch.from_numpy(P['packed'].reshape(-1).copy()).cuda(), 4, P['packed'].shape[1], torch.from_numpy(atlas.data).cuda(),
                      torch.from_numpy(synth.dot_stamp().reshape(-1)).cuda(), pix, W, max_glyphs=P['max_glyphs'],
                      canvas_cap=int(P['canvas_w'].max()), width_cap=int(P['widths'].max()))
    assert np.array_equal(pix.cpu().numpy(), synth_ref.render_batch(P, atlas, W, arc=False)), 'synthesised captchas differ from Pillow'
    print('smoke ok: max|logits - oracle| = %.2e, loss %.6f (oracle %.6f), next-step loss %.6f' % (err, loss, float(total), loss2))
