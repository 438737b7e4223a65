"""A/B of the wgrad9 read-stream variants (OCR_W9_VARIANT is read once per process, so each variant runs in a child process)
and the s_memtime stamps of the debug variants.  python tools/w9_variants.py   (on the GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from lstm_ctc_ocr_amd import ops, _native as nat
import ctypes
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
for name, W, H, Ci, Co in [("conv2", 128, 16, 64, 128), ("conv3_2", 64, 8, 256, 256), ("conv4_2", 64, 4, 512, 512)]:
    x = torch.randn(64, W, H, Ci, generator=g).to(dev).to(BF); y = torch.randn(64, W, H, Co, generator=g).to(dev).to(BF)
    dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
    ws = torch.empty(ops.conv3x3_wgrad_workspace_bytes(64, W, H, Ci, Co), dtype=torch.uint8, device=dev)
    us = timeit(lambda: ops.conv3x3_wgrad(x, y, dw, dbias=db, workspace=ws))
    dw.zero_(); ops.conv3x3_wgrad(x, y, dw, dbias=db, workspace=ws); torch.cuda.synchronize()
    out[name] = {"us": us, "checksum": float(dw.double().abs().sum())}
    if int(os.environ.get("OCR_W9_VARIANT", "0")) >= 10 and name == "conv4_2":
        lib = nat.lib(); lib.ocr_wgrad9_debug.argtypes = [ctypes.c_void_p]; lib.ocr_wgrad9_debug.restype = ctypes.c_int
        dbg = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
        lib.ocr_wgrad9_debug(dbg.data_ptr())
        ops.conv3x3_wgrad(x, y, dw, dbias=db, workspace=ws); torch.cuda.synchronize()
        lib.ocr_wgrad9_debug(None)
        d = dbg.cpu().numpy().reshape(8, 64, 8)[:, :32, :6]
        out["stamps"] = d.tolist()
print("RESULT " + json.dumps(out))
''' % ROOT


def main():
    res = {}
    for v in [int(a) for a in sys.argv[1:]] or (0, 3, 6, 7, 8, 9, 16, 17):
        env = dict(os.environ, OCR_W9_VARIANT=str(v))
        p = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
        if not line:
            print('variant', v, 'FAILED', p.stderr[-800:])
            continue
        r = json.loads(line[0][7:])
        res[v] = r
        print('variant %2d: ' % v + '  '.join('%s %.1f us (chk %.6e)' % (k, r[k]['us'], r[k]['checksum']) for k in r if k != 'stamps'), flush=True)
        if 'stamps' in r:
            import numpy as np
            d = np.array(r['stamps'], dtype=np.float64)       # [wave][step][st0 top, st1 after vmcnt, st2 after barrier, st3 after DMA issue, st4 first tap ready, st5 end]
            for w in (0, 4, 3):
                s = d[w, 2:30]
                if v >= 16:          # continuous-stream kernel: st1 = arrival at the mid-step barrier, st2 = released, st3 = last DMA piece issued
                    print('   wave %d (cycles, median over steps): first tap ready after %.0f  barrier wait %.0f  barrier->last DMA issued %.0f  step total %.0f  (step-to-step %.0f)' % (
                        w, np.median(s[:, 4] - s[:, 0]), np.median(s[:, 2] - s[:, 1]), np.median(s[:, 3] - s[:, 2]), np.median(s[:, 5] - s[:, 0]),
                        np.median(s[1:, 0] - s[:-1, 0])))
                    continue
                print('   wave %d (cycles, median over steps): vmcnt wait %.0f  barrier %.0f  ->DMA issued %.0f  ->first tap ready(from barrier) %.0f  step total %.0f  (step-to-step %.0f)' % (
                    w, np.median(s[:, 1] - s[:, 0]), np.median(s[:, 2] - s[:, 1]), np.median(s[:, 3] - s[:, 2]), np.median(s[:, 4] - s[:, 2]),
                    np.median(s[:, 5] - s[:, 0]), np.median(s[1:, 0] - s[:-1, 0])))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump({str(k): {kk: vv for kk, vv in v.items() if kk != 'stamps'} for k, v in res.items()}, open(os.path.join(ROOT, 'gpurun_out', 'w9_variants.json'), 'w'))


if __name__ == '__main__':
    main()
