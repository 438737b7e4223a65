"""Phase breakdown of the fast CTC kernel (diagnostic)."""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device('cuda:0')
T, N, C, L = 63, 64, 64, 10
rng = np.random.RandomState(0)
act = torch.from_numpy(rng.randn(T, N, C).astype(np.float32)).to(dev)
labels = torch.from_numpy(rng.randint(1, 63, N * L).astype(np.int32)).to(dev)
ll = torch.full((N,), L, dtype=torch.int32, device=dev); sl = torch.full((N,), T, dtype=torch.int32, device=dev)
costs = torch.empty(N, device=dev); g = torch.empty(N, T, C, dtype=torch.bfloat16, device=dev)
for _ in range(3): ops.ctc_loss_train(act, g, 1.0 / N, labels, ll, sl, 31, costs, blank=0)
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
nat.call('ocr_ctc_debug', dbg.data_ptr())
ops.ctc_loss_train(act, g, 1.0 / N, labels, ll, sl, 31, costs, blank=0)
torch.cuda.synchronize()
nat.call('ocr_ctc_debug', 0)
t = dbg.cpu().numpy()
print("phase1 lse %.1f us | phase2 gather %.1f | phase3 alpha/beta %.1f | phase4 grad %.1f" % tuple((t[i + 1] - t[i]) / 100.0 for i in range(4)))
