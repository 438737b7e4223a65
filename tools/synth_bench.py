"""GPU-side captcha synthesis alone: microseconds per batch of 64 and images/s of ocr_captcha_synth with nothing else on the GPU (in the training
loop the launch runs on a side stream beside the step's kernels and its duration in a kernel trace includes waiting for CUs).
    python tools/synth_bench.py        (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops  # noqa: E402
from lstm_ctc_ocr_amd.utils import gen, synth  # noqa: E402

CONFIGS = {'W88_4to6char': dict(), 'W256_10char': dict(min_len=10, max_len=10, width=480), 'varwidth_2to12char': dict(min_len=2, max_len=12, px_per_char=48)}


def main():
    dev = 'cuda:0'
    atlas = synth.GlyphAtlas()
    d_atlas = torch.from_numpy(atlas.data).to(dev)
    d_stamp = torch.from_numpy(synth.dot_stamp().reshape(-1)).to(dev)
    out = {}
    for name, kw in CONFIGS.items():
        B = 64
        P = synth.draw_params(np.random.default_rng(1), B, atlas, strings=False, **kw)
        W = gen.padded_width(int(P['nw_out'].max()))
        packed = torch.from_numpy(P['packed'].reshape(-1).copy()).to(dev)
        pix = torch.empty((B, W, 32), dtype=torch.uint8, device=dev)
        args = dict(max_glyphs=P['max_glyphs'], canvas_cap=int(P['canvas_w'].max()), width_cap=int(P['widths'].max()))
        for _ in range(20):
            ops.captcha_synth(packed, B, P['packed'].shape[1], d_atlas, d_stamp, pix, W, **args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 500
        e0.record()
        for _ in range(n):
            ops.captcha_synth(packed, B, P['packed'].shape[1], d_atlas, d_stamp, pix, W, **args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        out[name] = {'us_per_batch_of_64': round(us, 2), 'images_per_s': round(B / us * 1e6), 'W': W, 'lds_bytes': 60 * (((args['canvas_cap'] + 15) & ~15) + ((args['width_cap'] + 15) & ~15)) + 34816,
                     'hbm_bytes_written': B * W * 32, 'param_bytes_read': int(packed.numel() * 4)}
        print(name, json.dumps(out[name]), flush=True)
    print('RESULT ' + json.dumps(out))


if __name__ == '__main__':
    main()
