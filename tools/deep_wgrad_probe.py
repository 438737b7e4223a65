import sys, torch
sys.path.insert(0, '/root/repo')
from lstm_ctc_ocr_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (Nb, W, H, Ci, Co) in ((32, 64, 2, 512, 512), (32, 64, 4, 256, 256), (32, 64, 8, 128, 128), (32, 128, 16, 64, 64)):
    x = torch.randn(Nb, W, H, Ci, device=dev).to(BF); y = torch.randn(Nb, W, H, Co, device=dev).to(BF)
    dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
    ws = torch.empty(max(16, ops.conv3x3_wgrad_workspace_bytes(Nb, W, H, Ci, Co)), dtype=torch.uint8, device=dev)
    fl = 2.0 * Nb * W * H * 9 * Ci * Co
    ta = timeit(lambda: ops.conv3x3_wgrad(x, y, dw, dbias=db)); ts = timeit(lambda: ops.conv3x3_wgrad(x, y, dw, dbias=db, workspace=ws))
    print('%s atomics %.1f us (%.0f TF)  slab %.1f us (%.0f TF)' % ((Nb, W, H, Ci, Co), ta, fl / ta / 1e6, ts, fl / ts / 1e6))
