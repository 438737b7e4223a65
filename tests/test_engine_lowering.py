"""Host-side lowering (Engine._lower) of the reference DSL's graphs, no GPU: which fusions the plan records for which layer chains.
The kernels behind the fusions are covered by tests/test_gpu_kernels.py / test_gpu_engine.py; this file pins the PREDICATES."""
import torch

from lstm_ctc_ocr_amd.config import cfg
from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network
from lstm_ctc_ocr_amd.network import Network


def _lower(net, **env):
    e = Engine.__new__(Engine)          # lowering only: no device memory, no native calls
    e.fuse_conv1_pool = True
    e.device = torch.device('cpu')
    e._lower(net)
    return {op.key.split(':', 1)[1]: op for op in e.ops}, e


class _BnNoReluPool(Network):
    """conv + batch norm WITHOUT ReLU followed by a 1 x 2 max-pool as the only consumer, then the reference's tail."""
    def __init__(self, relu):
        self.inputs = []
        self.data = self.placeholder('data', 'float32', [None, None, cfg.NUM_FEATURES])
        self.labels = self.placeholder('labels', 'int32', [None])
        self.time_step_len = self.placeholder('time_step_len', 'int32', [None])
        self.labels_len = self.placeholder('labels_len', 'int32', [None])
        self.keep_prob = self.placeholder('keep_prob', 'float32', [])
        self.layers = {'data': self.data, 'labels': self.labels, 'time_step_len': self.time_step_len, 'labels_len': self.labels_len}
        self.trainable = True
        (self.feed('data').conv_single(3, 3, 64, 1, 1, name='c1', c_i=1).max_pool(2, 2, 2, 2, padding='VALID', name='p1')
             .conv_single(3, 3, 64, 1, 1, name='cb', bn=True, relu=relu).max_pool(1, 2, 1, 2, padding='VALID', name='pb')
             .conv_single(3, 3, 64, 1, 1, name='cc', bn=True)
             .max_pool(1, 2, 1, 2, padding='VALID', name='pc').max_pool(1, 2, 1, 2, padding='VALID', name='pd')
             .conv_single(2, 2, 128, 1, 1, padding='VALID', name='c5', relu=False).reshape_squeeze_layer(d=128, name='rs'))
        self.feed('rs', 'time_step_len').bi_lstm(64, 1, name='logits')


def test_shipped_graph_fusions():
    ops, _ = _lower(get_network('LSTM_train'))
    assert ops['conv1'].fused_pool is ops['pool1'] and ops['conv2'].pool_after is not None and ops['conv3_2'].pool_after is not None
    assert ops['conv4_2'].bn_pool is ops['pool3'] and ops['conv4_1'].bn_pool is None
    assert ops['conv3_1'].pool_after is None and ops['conv2'].bn_pool is None


def test_bn_pool_fusion_predicate_with_and_without_relu():
    """the BN + 1 x 2 pool fusion does not ask for a ReLU — the relu flag travels with the layer into ocr_bn_train_fwd2 /
    bwd2 (tests/test_gpu_kernels.py::test_batchnorm* cover relu on and off with the pooled form) — and it is the ONLY form in which
    such a layer's backward sums are taken: the following layer's data gradient must not also deliver partial rows for it."""
    for relu in (True, False):
        ops, _ = _lower(_BnNoReluPool(relu))
        cb, cc = ops['cb'], ops['cc']
        assert cb.bn and cb.relu is relu and cb.bn_pool is ops['pb'] and ops['pb'].bn_fused_into is cb
        assert cc.bn_pool is ops['pc']                      # the first pool behind cc; the second one (pd) stays a pool launch
        assert getattr(ops['pd'], 'bn_fused_into', None) is None
        # the alloc-time rule of _ConvOp.alloc: a producer with bn_pool never gets partial rows from its consumer's data gradient
        for p in (cb, cc):
            fusable_rows = (p.bn and p.relu and p.consumers == 1 and p.tail_into is None and p.mask_from is None and p.bn_pool is None
                            and p.kind != 'c1')
            assert not fusable_rows
