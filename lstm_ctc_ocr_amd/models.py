"""The two shipped graphs and the factory — /root/reference/lib/networks/{LSTM_train,LSTM_test,factory}.py.

Same constructor signature, attribute names (.data .labels .time_step_len .labels_len .keep_prob .layers .inputs
.trainable) and layer chain; the attributes are named input slots of the plan instead of TF placeholders."""
from .config import cfg
from .network import Network


def _vgg_crnn(net):
    # LSTM_train.py:22-38 / LSTM_test.py:18-34
    (net.feed('data')
        .conv_single(3, 3, 64, 1, 1, name='conv1', c_i=cfg.NCHANNELS)
        .max_pool(2, 2, 2, 2, padding='VALID', name='pool1')
        .conv_single(3, 3, 128, 1, 1, name='conv2')
        .max_pool(2, 2, 2, 2, padding='VALID', name='pool2')
        .conv_single(3, 3, 256, 1, 1, name='conv3_1')
        .conv_single(3, 3, 256, 1, 1, name='conv3_2')
        .max_pool(1, 2, 1, 2, padding='VALID', name='pool2')      # name reused by the reference (overwrites the LUT entry)
        .conv_single(3, 3, 512, 1, 1, name='conv4_1', bn=True)
        .conv_single(3, 3, 512, 1, 1, name='conv4_2', bn=True)
        .max_pool(1, 2, 1, 2, padding='VALID', name='pool3')
        .conv_single(2, 2, 512, 1, 1, padding='VALID', name='conv5', relu=False)
        .reshape_squeeze_layer(d=512, name='reshaped_layer'))
    (net.feed('reshaped_layer', 'time_step_len')
        .bi_lstm(cfg.TRAIN.NUM_HID, cfg.TRAIN.NUM_LAYERS, name='logits'))


class LSTM_train(Network):
    def __init__(self, trainable=True):
        self.inputs = []
        self.data = self.placeholder('data', 'float32', [None, None, cfg.NUM_FEATURES])   # N * time * features
        self.labels = self.placeholder('labels', 'int32', [None])
        self.time_step_len = self.placeholder('time_step_len', 'int32', [None])
        self.labels_len = self.placeholder('labels_len', 'int32', [None])
        self.keep_prob = self.placeholder('keep_prob', 'float32', [])
        self.layers = dict({'data': self.data, 'labels': self.labels, 'time_step_len': self.time_step_len,
                            'labels_len': self.labels_len})
        self.trainable = trainable
        self.setup()

    def setup(self):
        _vgg_crnn(self)


class LSTM_test(Network):
    def __init__(self, trainable=True):
        self.inputs = []
        self.data = self.placeholder('data', 'float32', [None, None, cfg.NUM_FEATURES])
        self.time_step_len = self.placeholder('time_step_len', 'int32', [None])
        self.keep_prob = self.placeholder('keep_prob', 'float32', [])
        self.layers = dict({'data': self.data, 'time_step_len': self.time_step_len})
        self.trainable = trainable
        self.setup()

    def setup(self):
        _vgg_crnn(self)


def _basic_block(net, src, c, name):
    """ResNet BasicBlock through the reference DSL: conv-bn-relu, conv-bn, (+ 1x1 conv-bn projection when the width
    changes), add, relu.  `src` is the name of the block input layer; returns the name of the block output."""
    cin = net.layers[src].channels
    (net.feed(src)
        .conv_single(3, 3, c, 1, 1, name=name + '_a', bn=True)
        .conv_single(3, 3, c, 1, 1, name=name + '_b', bn=True, relu=False))
    skip = src
    if cin != c:
        net.feed(src).conv_single(1, 1, c, 1, 1, name=name + '_proj', bn=True, relu=False)
        skip = name + '_proj'
    net.feed(skip, name + '_b').add(name=name + '_add').relu(name=name + '_out')
    return name + '_out'


def _resnet_crnn(net, blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)):
    """BASELINE configs[4]: ResNet-34-style extractor (BasicBlocks [3,4,6,3], stride-1 convs + the reference's pooling
    schedule so that T = W/4 - 1) + 2 stacked BiLSTM(num_hid) + CTC — not in the reference, expressed in its DSL."""
    (net.feed('data')
        .conv_single(3, 3, widths[0], 1, 1, name='conv1', c_i=cfg.NCHANNELS)
        .max_pool(2, 2, 2, 2, padding='VALID', name='pool1'))
    cur = 'pool1'
    pools = {0: (2, 2, 'pool2'), 1: (1, 2, 'pool3'), 2: (1, 2, 'pool4')}
    for stage, (nb, c) in enumerate(zip(blocks, widths)):
        for b in range(nb):
            cur = _basic_block(net, cur, c, 'res%d_%d' % (stage + 1, b))
        if stage in pools:
            kh, kw, pname = pools[stage]
            net.feed(cur).max_pool(kh, kw, kh, kw, padding='VALID', name=pname)
            cur = pname
    (net.feed(cur)
        .conv_single(2, 2, widths[-1], 1, 1, padding='VALID', name='conv5', relu=False)
        .reshape_squeeze_layer(d=widths[-1], name='reshaped_layer'))
    (net.feed('reshaped_layer', 'time_step_len')
        .bi_lstm(cfg.TRAIN.NUM_HID, cfg.TRAIN.NUM_LAYERS, name='logits', honour_num_layers=True))


class RESNET_train(LSTM_train):
    blocks, widths = (3, 4, 6, 3), (64, 128, 256, 512)

    def setup(self):
        _resnet_crnn(self, self.blocks, self.widths)


class RESNET_test(LSTM_test):
    blocks, widths = (3, 4, 6, 3), (64, 128, 256, 512)

    def setup(self):
        _resnet_crnn(self, self.blocks, self.widths)


_NETWORKS = {'LSTM_train': LSTM_train, 'LSTM_test': LSTM_test, 'RESNET_train': RESNET_train, 'RESNET_test': RESNET_test}


def get_network(name):
    """'LSTM_train' / 'LSTM_test' -> network object; anything else under the LSTM_ prefix -> KeyError
    (factory.py:13-21; other prefixes return None there, kept)."""
    parts = name.split('_')
    if parts[0] == 'LSTM':
        if len(parts) > 1 and parts[1] == 'train':
            return LSTM_train()
        elif len(parts) > 1 and parts[1] == 'test':
            return LSTM_test()
        raise KeyError('Unknown dataset: {}'.format(name))
    if name in _NETWORKS:                 # configurations beyond the reference (deep CRNN)
        return _NETWORKS[name]()
    return None


def list_networks():
    return list(_NETWORKS.keys())
