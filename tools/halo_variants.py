"""Timing ablations / variants of conv_halo_kernel (forward of the five 3x3 layers at batch 64, W = 256), one child process per
setting because the knobs are read once per process.   python tools/halo_variants.py [KEY=VAL,KEY=VAL ...]   (on the GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = {}
g = torch.Generator(device="cpu").manual_seed(1)
for name, W, H, Ci, Co in [("conv2", 128, 16, 64, 128), ("conv3_1", 64, 8, 128, 256), ("conv3_2", 64, 8, 256, 256), ("conv4_1", 64, 4, 256, 512), ("conv4_2", 64, 4, 512, 512)]:
    x = torch.randn(64, W, H, Ci, generator=g).to(dev).to(BF)
    wp = (torch.randn(Co, 3, 3, Ci, generator=g) * 0.05).to(dev).to(BF)
    b = torch.zeros(Co, device=dev); y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    us = timeit(lambda: ops.conv3x3(x, wp, out=y, bias=b, relu=True))
    out[name] = {"us": us, "tflops": 2.0 * 64 * W * H * 9 * Ci * Co / us / 1e6, "checksum": float(y.float().abs().sum())}
print("RESULT " + json.dumps(out))
''' % ROOT


def main():
    settings = sys.argv[1:] or ["", "OCR_HALO_ABL=1", "OCR_HALO_ABL=2", "OCR_HALO_ABL=5", "OCR_HALO_NW=8", "OCR_HALO_NW=8,OCR_HALO_ABL=5",
                                "OCR_HALO_NW=4", "OCR_HALO_NW=4,OCR_HALO_ABL=5"]
    for st in settings:
        env = dict(os.environ)
        for kv in filter(None, st.split(',')):
            k, v = kv.split('=')
            env[k] = v
        p = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
        if not line:
            print('%-32s FAILED %s' % (st, p.stderr[-600:]))
            continue
        r = json.loads(line[0][7:])
        print('%-32s ' % (st or 'default') + '  '.join('%s %.1f us %.0f TF' % (k, r[k]['us'], r[k]['tflops']) for k in r) +
              '   sum %.1f us' % sum(v['us'] for v in r.values()), flush=True)


if __name__ == '__main__':
    main()
