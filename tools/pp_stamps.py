"""s_memtime stamps of conv_pp_kernel (workgroup 0) on the conv4_2 / conv3_2 forward shapes.  python tools/pp_stamps.py (GPU box)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = nat.lib(); lib.ocr_conv_pp_debug.argtypes = [ctypes.c_void_p]; lib.ocr_conv_pp_debug.restype = ctypes.c_int
for name, W, H, Ci, Co in [("conv4_2", 64, 4, 512, 512), ("conv3_2", 64, 8, 256, 256)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); wp = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    b = torch.zeros(Co, device=dev); y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    for _ in range(3): ops.conv3x3(x, wp, out=y, bias=b, relu=True)
    dbg = torch.zeros(8 * 80 * 8, dtype=torch.int64, device=dev)
    lib.ocr_conv_pp_debug(dbg.data_ptr())
    ops.conv3x3(x, wp, out=y, bias=b, relu=True); torch.cuda.synchronize()
    lib.ocr_conv_pp_debug(None)
    d = dbg.cpu().numpy().reshape(8, 80, 8).astype(np.float64)
    n = 9 * Ci // 64
    print(name, 'steps', n)
    for w in (0, 4, 1, 5):
        s = d[w, 3:n - 2]
        md = lambda a: np.median(a)
        print('  wave %d: DMA issue %.0f | reads issue %.0f | wait(vm,lgkm) %.0f | barrier %.0f | MFMA %.0f | vmwait %.0f | barrier %.0f | step %.0f' % (
            w, md(s[:, 1] - s[:, 0]), md(s[:, 2] - s[:, 1]), md(s[:, 3] - s[:, 2]), md(s[:, 4] - s[:, 3]), md(s[:, 5] - s[:, 4]),
            md(s[:, 6] - s[:, 5]), md(s[:, 7] - s[:, 6]), md(s[1:, 0] - s[:-1, 0])))
