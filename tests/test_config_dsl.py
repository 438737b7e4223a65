"""Host-side interface parity with the reference (no GPU): config keys / merge rules (lib/lstm/config.py), the layer
DSL's chaining and error behaviour (lib/networks/network.py), the factory, and the import surface the reference's own
CLI scripts need."""
import os

import numpy as np
import pytest


def test_cfg_defaults_and_yaml_merge(tmp_path):
    from lstm_ctc_ocr_amd import config
    cfg = config.cfg
    assert cfg.NCLASSES == 64 and len(cfg.CHARSET) == 62 and cfg.POOL_SCALE == 4 and cfg.OFFSET_TIME_STEP == -1
    assert cfg.TRAIN.NUM_HID == 512 and cfg.TRAIN.BATCH_SIZE == 64 and cfg.VAL.BATCH_SIZE == 128 and cfg.RNG_SEED == 3
    enc, dec = config.get_encode_decode_dict()
    assert enc['0'] == 1 and enc['Z'] == 62 and enc[''] == 0 and dec[0] == '' and dec[11] == 'a'
    old = (cfg.TRAIN.LEARNING_RATE, cfg.EXP_DIR)
    yml = tmp_path / "a.yml"
    yml.write_text("EXP_DIR: lstm_ctc\nTRAIN:\n  LEARNING_RATE: 0.0001\n")
    config.cfg_from_file(str(yml))
    assert cfg.TRAIN.LEARNING_RATE == 1e-4 and cfg.EXP_DIR == 'lstm_ctc'
    bad = tmp_path / "b.yml"; bad.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        config.cfg_from_file(str(bad))
    bad.write_text("TRAIN:\n  LEARNING_RATE: 1\n")              # int where a float is expected
    with pytest.raises(ValueError):
        config.cfg_from_file(str(bad))
    config.cfg_from_list(['TRAIN.LEARNING_RATE', '0.01', 'EXP_DIR', 'default'])
    assert cfg.TRAIN.LEARNING_RATE == 0.01 and cfg.EXP_DIR == 'default'
    with pytest.raises(AssertionError):
        config.cfg_from_list(['TRAIN.DISPLAY', '0.5'])           # type must match exactly
    cfg.TRAIN.LEARNING_RATE, cfg.EXP_DIR = old
    # the shipped yml parses and carries the reference's values
    config.cfg_from_file(os.path.join(os.path.dirname(__file__), '..', 'lstm', 'lstm.yml'))
    assert cfg.TRAIN.SOLVER == 'Adam' and cfg.TRAIN.WEIGHT_DECAY == 1e-5 and cfg.TRAIN.SNAPSHOT_ITERS == 2000


def test_layer_dsl_contract():
    from lstm_ctc_ocr_amd.models import LSTM_test, LSTM_train, get_network, list_networks
    net = get_network('LSTM_train')
    assert isinstance(net, LSTM_train) and isinstance(get_network('LSTM_test'), LSTM_test)
    with pytest.raises(KeyError):
        get_network('LSTM_bogus')
    assert {'LSTM_train', 'LSTM_test'} <= set(list_networks())          # plus the RESNET_* models of BASELINE configs[4]
    for attr in ('data', 'labels', 'time_step_len', 'labels_len', 'keep_prob', 'layers', 'inputs', 'trainable'):
        assert hasattr(net, attr)
    for name in ('logits', 'time_step_len', 'labels', 'labels_len'):
        net.get_output(name)
    with pytest.raises(KeyError):
        net.get_output('nope')
    with pytest.raises(KeyError):
        net.feed('nope')
    net.inputs = []
    with pytest.raises(RuntimeError):
        net.max_pool(2, 2, 2, 2, name='p')
    with pytest.raises(AssertionError):
        net.feed('conv1').max_pool(2, 2, 2, 2, name='p', padding='FULL')
    assert net.feed('conv1').get_unique_name('conv') == 'conv_8'          # counts existing names with the prefix, +1
    loss, dense = net.build_loss()
    assert loss.op == 'ctc_loss' and dense.op == 'ctc_decode'
    # parameters carry the TF variable names / layouts a reference checkpoint would hold
    specs = net.param_specs
    assert specs['conv1/weights'].shape == (3, 3, 1, 64) and specs['conv5/weights'].shape == (2, 2, 512, 512)
    assert specs['conv4_1/conv4_1/gamma'].shape == (512,) and specs['logits/fw/weights'].shape == (768, 1024)
    assert specs['logits/weights'].shape == (512, 64)
    assert sum(int(np.prod(s.shape)) for s in specs.values()) == 7158592
    reg = sorted(k for k, s in specs.items() if s.regularized)
    assert reg == sorted(['conv1/weights', 'conv2/weights', 'conv3_1/weights', 'conv3_2/weights', 'conv4_1/weights',
                          'conv4_2/weights', 'conv5/weights', 'logits/weights'])


def test_reference_import_surface():
    import lib.lstm                                           # eager config+train import like lib/lstm/__init__.py:8-9
    from easydict import EasyDict as edict
    from lib.lstm.config import cfg, cfg_from_file, cfg_from_list, get_log_dir, get_output_dir  # noqa: F401
    from lib.lstm.test import test_net  # noqa: F401
    from lib.lstm.train import train_net  # noqa: F401
    from lib.lstm.utils.gen import get_batch  # noqa: F401
    from lib.lstm.utils.timer import Timer
    from lib.lstm.utils.training import accuracy_calculation
    from lib.networks.factory import get_network  # noqa: F401
    from lib.networks.network import Network  # noqa: F401
    from lib.utils.data_util import GeneratorEnqueuer  # noqa: F401
    d = edict({'a': {'b': 1}})
    assert d.a.b == 1
    t = Timer(); t.tic(); assert t.toc(average=False) >= 0 and t.calls == 1
    assert accuracy_calculation([[1, 2, 0], [3]], [[1, 2], [4, 0]], ignore_value=0, isPrint=False) == 0.5
    assert accuracy_calculation([[1]], [[1], [2]], isPrint=False) == 0       # length mismatch -> 0 (training.py:27-29)


def test_reference_cli_script_runs_against_this_lib(monkeypatch, tmp_path, capsys):
    """Executes the reference's OWN lstm/train_net.py source (when the checkout is present) with this repo's lib/:
    argument parsing, cfg merge, directory creation and network construction must all work; the final train_net call
    is intercepted because this host has no GPU."""
    ref = '/root/reference/lstm/train_net.py'
    if not os.path.exists(ref):
        pytest.skip('reference checkout not present')
    import sys
    import lib.lstm.train as lt
    from lstm_ctc_ocr_amd.config import cfg
    called = {}
    monkeypatch.setattr(lt, 'train_net', lambda network, imgdb, **kw: called.update(net=network, kw=kw, db=imgdb))
    monkeypatch.setitem(cfg, 'ROOT_DIR', str(tmp_path))
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lstm', 'train_net.py')
    monkeypatch.setattr(sys, 'argv', ['train_net.py', '--network=LSTM_train', '--cfg=' + os.path.join(os.path.dirname(here), 'lstm.yml'),
                                      '--restore=0', '--iters', '5'])
    src = open(ref).read()
    exec(compile(src, here, 'exec'), {'__file__': here, '__name__': '__main__'})
    assert called['kw']['max_iters'] == 5 and called['kw']['restore'] is False and called['db'].name == 'lstm_train'
    assert type(called['net']).__name__ == 'LSTM_train'
    assert os.path.isdir(os.path.join(str(tmp_path), 'output', 'lstm_ctc'))


def test_flat_layout_buckets():
    """Flat parameter layout of the engine: one contiguous regularised range, late layers (>= 75 % of the parameters) on top."""
    from lstm_ctc_ocr_amd.layout import FlatLayout
    from lstm_ctc_ocr_amd.models import get_network
    net = get_network('LSTM_train')
    specs = list(net.param_specs.values())
    lay = FlatLayout(specs, 64)
    assert lay.split_layer == 'conv4_1'
    n = {s.name: int(np.prod(s.shape)) for s in specs}
    assert sum(n.values()) == 7158592 and lay.n_total >= 7158592 and lay.n_total % 64 == 0
    r0, r1 = lay.reg_range
    for s in specs:
        o = lay.offsets[s.name]
        assert o % 64 == 0
        assert (r0 <= o < r1) == bool(s.regularized), s.name                       # regularised <=> inside the range
        late = s.name.split('/')[0] in ('conv4_1', 'conv4_2', 'conv5', 'logits')
        assert (o >= lay.late_begin) == late, s.name                               # late <=> upper bucket
    spans = sorted((lay.offsets[k], lay.offsets[k] + n[k]) for k in n)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))                     # no overlap
    late_params = sum(v for k, v in n.items() if lay.offsets[k] >= lay.late_begin)
    assert late_params >= 0.75 * 7158592 and 0 < r0 < lay.late_begin < r1 < lay.n_total
    # a one-layer net has nothing to split
    one = FlatLayout([s for s in specs if s.name.startswith('conv1/')], 64)
    assert one.split_layer is None and one.late_begin == one.n_total


def test_flat_layout_follows_execution_order_on_residual_graphs():
    """A residual block's 1x1 projection is DEFINED after its two convolutions but EXECUTED before them (it is the first
    input of the add).  The late bucket must be a suffix of the execution order, otherwise the early backward graph would
    write a gradient that the concurrent all-reduce of the late bucket is reading (round-1 advisor finding)."""
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.layout import FlatLayout, execution_order, layer_of
    from lstm_ctc_ocr_amd.models import get_network
    old = (cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID)
    try:
        for hid in (512, 1024, 2048):
            cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID = 96, 2, hid
            net = get_network('RESNET_train')
            order = [nd.name for nd in execution_order(net.get_output('logits'))]
            i_proj, i_a = order.index('res4_0_proj'), order.index('res4_0_a')
            assert i_proj < i_a                                                   # the premise
            specs = list(net.param_specs.values())
            lay = FlatLayout(specs, 64, order=order)
            assert lay.split_layer is not None
            split = order.index(lay.split_layer)
            for s in specs:
                own = layer_of(s.name, order)
                assert own == lay.owner[s.name]
                assert (lay.offsets[s.name] >= lay.late_begin) == (order.index(own) >= split), s.name
            n_late = sum(int(np.prod(s.shape)) for s in specs if lay.offsets[s.name] >= lay.late_begin)
            assert n_late >= 0.75 * sum(int(np.prod(s.shape)) for s in specs)
            # stacked BiLSTM variables belong to their own operator, not to the final 'logits' one
            assert lay.owner['logits/stack0/fw/weights'] == 'logits/stack0' and lay.owner['logits/fw/weights'] == 'logits'
    finally:
        cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID = old
    # the shipped chain: execution order == definition order, same layout as before
    net = get_network('LSTM_train')
    a = FlatLayout(list(net.param_specs.values()), 64)
    b = FlatLayout(list(net.param_specs.values()), 64, order=[nd.name for nd in execution_order(net.get_output('logits'))])
    assert a.offsets == b.offsets and a.late_begin == b.late_begin and a.split_layer == b.split_layer == 'conv4_1'


def test_own_cli_drivers_parse_and_open_a_session(monkeypatch, tmp_path, capsys):
    """This repository's lstm/train_net.py and lstm/test_net.py: the reference's flag set, config overlay from lstm.yml and
    --set, directory creation, network construction; the final train_net / test_net call is intercepted (it needs a GPU)."""
    import importlib
    lstm_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lstm')
    monkeypatch.syspath_prepend(lstm_dir)
    monkeypatch.setenv('OCR_ROOT_DIR', str(tmp_path))
    from lstm_ctc_ocr_amd.config import cfg
    monkeypatch.setattr(cfg, 'ROOT_DIR', str(tmp_path))
    import copy
    saved = copy.deepcopy({k: dict(v) if isinstance(v, dict) else v for k, v in cfg.items()})
    import lib.lstm.test as lt_test
    import lib.lstm.train as lt_train
    seen = {}
    monkeypatch.setattr(lt_train, 'train_net', lambda network, imgdb, **kw: seen.update(train=(network, imgdb, kw)))
    monkeypatch.setattr(lt_test, 'test_net', lambda network, imgdb, **kw: seen.update(test=(network, imgdb, kw)))
    try:
        train_cli = importlib.import_module('train_net')
        train_cli.main(['--network=LSTM_train', '--cfg=' + os.path.join(lstm_dir, 'lstm.yml'), '--restore=1', '--iters', '7',
                        '--set', 'TRAIN.DISPLAY', '3'])
        net, db, kw = seen['train']
        assert type(net).__name__ == 'LSTM_train' and db.name == 'lstm_train' and kw['max_iters'] == 7 and kw['restore'] is True
        assert cfg.TRAIN.DISPLAY == 3 and cfg.TRAIN.LEARNING_RATE == 1e-4            # --set and lstm.yml both applied
        assert os.path.isdir(kw['output_dir']) and os.path.isdir(kw['log_dir']) and str(tmp_path) in kw['output_dir']
        test_cli = importlib.import_module('test_net')
        test_cli.main(['--network=LSTM_test', '--cfg=' + os.path.join(lstm_dir, 'lstm.yml')])
        net, db, kw = seen['test']
        assert type(net).__name__ == 'LSTM_test' and kw['restore'] is True and kw['testDir'] == './data/val/'
        out = capsys.readouterr().out
        assert 'Called with args:' in out and 'Using config:' in out and 'Use network `LSTM_test` in training' in out
    finally:
        for k, v in saved.items():                      # the drivers overlay lstm.yml on the global configuration
            if isinstance(v, dict):
                cfg[k].update(v)
            else:
                cfg[k] = v


def test_unidirectional_lstm_layer_plan_and_oracle():
    """Network.lstm (network.py:130-152): MultiRNNCell of LSTMCell(num_hids) under dynamic_rnn + FC.  Plan, TF variable names /
    shapes / initialisers, ownership of the cells' variables (the hidden layer's op, for the data-parallel buckets) and the
    plan-walking oracle (zeros past each length, logits time-major)."""
    import torch
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.layout import FlatLayout, execution_order, host_parameters
    from lstm_ctc_ocr_amd.network import Network
    from oracle import plan_exec

    class Net(Network):
        def __init__(self):
            self.inputs = []
            self.layers = {'data': self.placeholder('data', 'float32', [None, None, 32]),
                           'time_step_len': self.placeholder('time_step_len', 'int32', [None])}
            self.trainable = True
            self.setup()

        def setup(self):
            (self.feed('data').conv_single(3, 3, 64, 1, 1, name='c1', c_i=1).max_pool(2, 2, 2, 2, padding='VALID', name='p1')
                 .max_pool(2, 2, 2, 2, padding='VALID', name='p2').max_pool(1, 2, 1, 2, padding='VALID', name='p3')
                 .max_pool(1, 2, 1, 2, padding='VALID', name='p4')
                 .conv_single(2, 2, 64, 1, 1, padding='VALID', name='c5', relu=False).reshape_squeeze_layer(d=64, name='rs'))
            self.feed('rs', 'time_step_len').lstm(32, 3, name='logits')

    net = Net()
    order = [nd.name for nd in execution_order(net.get_output('logits'))]
    assert order[-3:] == ['logits/rnn/multi_rnn_cell/cell_0', 'logits/rnn/multi_rnn_cell/cell_1', 'logits']
    sp = net.param_specs
    assert tuple(sp['logits/rnn/multi_rnn_cell/cell_0/lstm_cell/weights'].shape) == (64 + 32, 128)
    assert tuple(sp['logits/rnn/multi_rnn_cell/cell_2/lstm_cell/weights'].shape) == (32 + 32, 128)
    assert tuple(sp['logits/weights'].shape) == (32, cfg.NCLASSES) and sp['logits/weights'].regularized
    assert sp['logits/weights'].init == ('truncated_normal', 0.1)
    lay = FlatLayout(sp.values(), 64, order=order)
    assert lay.owner['logits/rnn/multi_rnn_cell/cell_1/lstm_cell/biases'] == 'logits/rnn/multi_rnn_cell/cell_1'
    assert lay.owner['logits/rnn/multi_rnn_cell/cell_2/lstm_cell/weights'] == 'logits'
    params = host_parameters(net, 5)
    assert float(params['logits/weights'].abs().max()) <= 0.2 + 1e-6          # truncated at two standard deviations
    x = torch.rand(3, 64, 32, generator=torch.Generator().manual_seed(0))
    lens = [15, 4, 9]
    logits, inter = plan_exec.forward(net, params, x, lens, sim_bf16=False, keep=True)
    assert tuple(logits.shape) == (15, 3, cfg.NCLASSES)
    h0 = inter['logits/rnn/multi_rnn_cell/cell_0']
    assert float(h0[1, 4:].abs().max()) == 0.0 and float(h0[1, :4].abs().max()) > 0.0
    # past its length a sample's logits are the FC bias alone
    assert torch.allclose(logits[4:, 1], params['logits/biases'].expand(11, -1), atol=1e-6)


def test_fc_layer_plan_and_oracle():
    """Network.fc (network.py:415-447) on the row tensor: TF variable names / shapes / initialisers / regulariser, its place in the execution
    order and the parameter layout, the refusal of a 4-D map of dynamic width, and the plan-walking oracle against a plain torch product."""
    import torch
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.layout import FlatLayout, execution_order, host_parameters
    from lstm_ctc_ocr_amd.network import Network
    from oracle import plan_exec

    class Net(Network):
        def __init__(self):
            self.inputs = []
            self.layers = {'data': self.placeholder('data', 'float32', [None, None, 32]),
                           'time_step_len': self.placeholder('time_step_len', 'int32', [None])}
            self.trainable = True
            self.setup()

        def setup(self):
            (self.feed('data').conv_single(3, 3, 64, 1, 1, name='c1', c_i=1).max_pool(2, 2, 2, 2, padding='VALID', name='p1')
                 .max_pool(2, 2, 2, 2, padding='VALID', name='p2').max_pool(1, 2, 1, 2, padding='VALID', name='p3')
                 .max_pool(1, 2, 1, 2, padding='VALID', name='p4')
                 .conv_single(2, 2, 64, 1, 1, padding='VALID', name='c5', relu=False).reshape_squeeze_layer(d=64, name='rs')
                 .fc(48, name='fc1').fc(40, name='bbox_pred', relu=False))
            self.feed('bbox_pred', 'time_step_len').bi_lstm(32, 1, name='logits')

    net = Net()
    order = [nd.name for nd in execution_order(net.get_output('logits'))]
    assert order[-4:] == ['rs', 'fc1', 'bbox_pred', 'logits']
    sp = net.param_specs
    assert tuple(sp['fc1/weights'].shape) == (64, 48) and sp['fc1/weights'].regularized and sp['fc1/weights'].init == ('truncated_normal', 0.01)
    assert sp['bbox_pred/weights'].init == ('truncated_normal', 0.001) and tuple(sp['bbox_pred/biases'].shape) == (40,)
    assert not sp['fc1/biases'].regularized and sp['fc1/biases'].init == 'zeros'
    assert tuple(sp['logits/fw/weights'].shape) == (40 + 16, 64)           # the BiLSTM sees fc's num_out channels
    lay = FlatLayout(sp.values(), 64, order=order)
    assert lay.owner['fc1/weights'] == 'fc1'
    with pytest.raises(NotImplementedError):
        net.feed('p2').fc(10, name='bad')
    with pytest.raises(RuntimeError):                                       # no inputs: the @layer contract (network.py:24-25)
        net.feed('rs'); net.inputs = []; net.fc(10, name='bad2')
    params = host_parameters(net, 5)
    assert float(params['fc1/weights'].abs().max()) <= 0.02 + 1e-7         # truncated at two standard deviations
    params['fc1/weights'] = params['fc1/weights'] * 20
    params['fc1/biases'] = torch.linspace(-0.1, 0.1, 48)
    x = torch.rand(3, 64, 32, generator=torch.Generator().manual_seed(0))
    logits, inter = plan_exec.forward(net, params, x, [15, 4, 9], sim_bf16=False, keep=True)
    assert tuple(logits.shape) == (15, 3, cfg.NCLASSES)
    want = torch.relu(inter['rs'] @ params['fc1/weights'] + params['fc1/biases'])
    assert tuple(inter['fc1'].shape) == (3, 15, 48) and torch.allclose(inter['fc1'], want, atol=1e-6)
    assert torch.allclose(inter['bbox_pred'], want @ params['bbox_pred/weights'] + params['bbox_pred/biases'], atol=1e-6)      # relu=False: xw_plus_b
