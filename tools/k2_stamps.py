"""s_memtime stamps of conv_k2_kernel's workgroup 0 (experiments build: make EXPERIMENTS=1 TARGET=../libocrhip_exp.so), per interval and wave:
start of the interval -> MFMAs start -> MFMAs done -> in front of the barrier.  Waves 0-3 load first and multiply second, waves 4-7 the
other way round.    OCR_NATIVE_LIB=.../libocrhip_exp.so OCR_CONV_K2=1 OCR_K2_CFG=A OCR_K2_ABL=8 python tools/k2_stamps.py   (GPU box)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = nat.lib(); lib.ocr_conv_k2_debug.argtypes = [ctypes.c_void_p]; lib.ocr_conv_k2_debug.restype = ctypes.c_int
for name, W, H, Ci, Co in [("conv4_2", 64, 4, 512, 512), ("conv3_2", 64, 8, 256, 256)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); wp = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    b = torch.zeros(Co, device=dev); y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    for _ in range(3): ops.conv3x3(x, wp, out=y, bias=b, relu=True)
    dbg = torch.zeros(8 * 80 * 4, dtype=torch.int32, device=dev)
    lib.ocr_conv_k2_debug(dbg.data_ptr())
    ops.conv3x3(x, wp, out=y, bias=b, relu=True); torch.cuda.synchronize()
    lib.ocr_conv_k2_debug(None)
    d = dbg.cpu().numpy().view(np.uint32).reshape(8, 80, 4).astype(np.int64)
    n = min(9 * Ci // 64, 80)
    md = lambda a: float(np.median(a & 0xffffffff if False else a))
    print(name, 'intervals', n)
    for w in (0, 1, 4, 5):
        s = d[w, 4:n - 2]
        nxt = d[w, 5:n - 1, 0]
        dif = lambda a, b: np.median((a - b) & 0xffffffff)
        if w < 4:
            print('  wave %d (load, multiply): LOAD + read wait %.0f | MFMAs %.0f | vmwait %.0f | barrier %.0f | interval %.0f' % (
                w, dif(s[:, 1], s[:, 0]), dif(s[:, 2], s[:, 1]), dif(s[:, 3], s[:, 2]), dif(nxt, s[:, 3]), dif(nxt, s[:, 0])))
        else:
            print('  wave %d (multiply, load): read wait %.0f | MFMAs %.0f | LOAD + vmwait + barrier %.0f | interval %.0f' % (
                w, dif(s[:, 1], s[:, 0]), dif(s[:, 2], s[:, 1]), dif(nxt, s[:, 2]), dif(nxt, s[:, 0])))
