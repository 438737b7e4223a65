import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network
from lstm_ctc_ocr_amd.selftest import tiny_batch
for graphs in (False, True):
    for N, W in ((8, 88), (64, 256)):
        eng = Engine(get_network('LSTM_train'), seed=3, use_graphs=graphs)
        x, labels, ll, sl = tiny_batch(N=N, W=W, L=3)
        sp = eng.plan(N, W)
        def flags(): 
            torch.cuda.synchronize(); return [int(w[-1]) for w in sp.lstm_sync], [w[::64][:9].tolist() for w in sp.lstm_sync]
        for i in range(3):
            eng.forward(x, sl); print(graphs, N, 'fwd', i, flags())
        eng._bind(sp, x, sl, labels, ll)
        for i in range(3):
            eng._run(sp, 'fb'); print(graphs, N, 'fb', i, flags())
