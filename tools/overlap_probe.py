"""Do a data-gradient convolution (conv_halo, MFMA-bound) and a weight-gradient launch (gemm_tn2, latency-bound) overlap
when issued on two streams?  (diagnostic for running the weight gradients beside the backward chain)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (W, H, Ci, Co) in [(64, 4, 512, 512), (64, 8, 256, 256), (128, 16, 64, 128)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); y = torch.randn(64, W, H, Co, device=dev).to(BF)
    wd = (torch.randn(Ci, 3, 3, Co, device=dev) * 0.05).to(BF)
    ox = torch.empty_like(x); dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
    dgrad = lambda: ops.conv3x3(y, wd, out=ox, mask=x)
    wgrad = lambda: ops.conv3x3_wgrad(x, y, dw, dbias=db)
    def run(mode, iters=20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            if mode == 'dgrad': dgrad()
            elif mode == 'wgrad': wgrad()
            elif mode == 'serial': dgrad(); wgrad()
            else:
                s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s1): dgrad()
                with torch.cuda.stream(s2): wgrad()
                torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    for m in ('dgrad', 'wgrad', 'serial', 'two-streams'):
        run(m, 3)
    print((W, H, Ci, Co), " ".join("%s %.1f us" % (m, run(m)) for m in ('dgrad', 'wgrad', 'serial', 'two-streams')), flush=True)
