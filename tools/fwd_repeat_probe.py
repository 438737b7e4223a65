"""Are two inference forwards of the same batch bit-identical, layer by layer?  (diagnostic for a decode mismatch seen with conv_ws on)
    python tools/fwd_repeat_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from lstm_ctc_ocr_amd.engine import Engine  # noqa: E402
from lstm_ctc_ocr_amd.models import get_network  # noqa: E402
from oracle import decode as odec  # noqa: E402
import test_golden as tg  # noqa: E402

d, x, labels, ll, sl = tg.c2_inputs()
_, params = tg.load_graph_fixture()
for graphs in (False, True):
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=1, use_graphs=graphs)
    eng.load_arrays({k: v.numpy() for k, v in params.items()})
    snaps = []
    for rep in range(3):
        logits = eng.forward(x, sl)
        torch.cuda.synchronize()
        sp = eng.plan(64, 256)
        snaps.append({k: v.clone() for k, v in sp.buf.items() if k.endswith('/y') or k.endswith('/hout') or k.endswith('/z')})
        if rep == 1:
            eng._bind(sp, x, sl, labels, ll); eng._run(sp, 'fb'); torch.cuda.synchronize()
    for k in snaps[0]:
        same = [bool(torch.equal(snaps[0][k], s[k])) for s in snaps[1:]]
        if not all(same):
            df = (snaps[0][k].float() - snaps[2][k].float()).abs()
            print('graphs=%s %-24s differs between repeats %s: %d elements, max %.3e' % (graphs, k, same, int((df > 0).sum()), float(df.max())), flush=True)
    lg = eng.forward(x, sl).float().cpu().numpy()
    dev_dec = eng.decode(x, sl, method='greedy')
    ora = odec.greedy_decode(lg, sl)
    nbad = sum(1 for a, b in zip(odec.dense(dev_dec).tolist(), odec.dense(ora).tolist()) if a != b)
    # exact ties at the arg max of the logits the decoders look at?
    top2 = np.sort(lg, axis=2)[:, :, -2:]
    ties = int((top2[:, :, 0] == top2[:, :, 1]).sum())
    print('graphs=%s: repeats compared, decode mismatches %d / 64, exact top-2 ties in the logits %d, logits dtype %s' % (graphs, nbad, ties, eng.forward(x, sl).dtype), flush=True)
