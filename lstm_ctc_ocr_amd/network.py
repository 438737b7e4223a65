"""Layer DSL of the reference, re-hosted as a deferred plan for the MI355X engine.

Mirrors the public surface of /root/reference/lib/networks/network.py (class Network, the chainable @layer
decorator, feed / get_output / get_unique_name / make_var / validate_padding / load / build_loss /
l2_regularizer and the layer methods), with the same argument meaning and error behaviour:
  * unknown layer name fed  -> KeyError            (network.py:73-75)
  * layer called with no inputs -> RuntimeError    (network.py:24-25)
  * bad padding string -> AssertionError           (network.py:94-95)

Where the reference emits TensorFlow graph ops, a layer call here appends a `Node` to a static plan; the plan
is lowered by lstm_ctc_ocr_amd.engine.Engine onto hand-written gfx950 kernels, with buffers laid out once per
input shape and the step replayed from a hipGraph.  Placeholders (`net.data`, `net.labels`, ...) become named
input slots that the drivers bind device tensors to.
"""
import numpy as np

DEFAULT_PADDING = 'SAME'


class Node(object):
    """A value in the plan: an input slot, a parameter-free view, or the output of a layer."""

    def __init__(self, op, name, inputs=(), **attrs):
        self.op = op
        self.name = name
        self.inputs = list(inputs)
        self.attrs = attrs

    # static channel count where known (needed to size parameters at plan time, network.py:163)
    @property
    def channels(self):
        return self.attrs.get('channels')

    def __repr__(self):
        return 'Node(%s:%s)' % (self.op, self.name)


class ParamSpec(object):
    def __init__(self, name, shape, init, trainable=True, regularized=False):
        self.name, self.shape, self.init, self.trainable, self.regularized = name, tuple(shape), init, trainable, regularized


def layer(op):
    """Decorator for composable network layers (same contract as network.py:19-38)."""
    def layer_decorated(self, *args, **kwargs):
        # Automatically set a name if not provided.
        name = kwargs.setdefault('name', self.get_unique_name(op.__name__))
        if len(self.inputs) == 0:
            raise RuntimeError('No input variables found for layer %s.' % name)
        elif len(self.inputs) == 1:
            layer_input = self.inputs[0]
        else:
            layer_input = list(self.inputs)
        layer_output = op(self, layer_input, *args, **kwargs)
        self.layers[name] = layer_output
        self.feed(layer_output)
        return self
    return layer_decorated


class Network(object):
    def __init__(self, inputs, trainable=True):
        self.inputs = []
        self.layers = dict(inputs)
        self.trainable = trainable
        self.setup()

    # -- plan-level state shared by subclasses that bypass __init__ (LSTM_train does, like the reference)
    @property
    def param_specs(self):
        if not hasattr(self, '_param_specs'):
            self._param_specs = {}
        return self._param_specs

    def setup(self):
        raise NotImplementedError('Must be subclassed.')

    def placeholder(self, name, dtype, shape):
        return Node('input', name, dtype=dtype, shape=tuple(shape),
                    channels=(shape[-1] if len(shape) == 4 else None))

    def load(self, data_path, session=None, ignore_missing=False):
        """npy dict {scope: {var: array}} loader (network.py:50-63); `session` is the Engine (or None: the
        arrays are staged and picked up when the engine materialises the parameters)."""
        data_dict = np.load(data_path, encoding='latin1', allow_pickle=True).item()
        staged = {}
        for key in data_dict:
            for subkey in data_dict[key]:
                full = key + '/' + subkey
                if full in self.param_specs:
                    staged[full] = np.asarray(data_dict[key][subkey], np.float32)
                    print("assign pretrain model " + subkey + " to " + key)
                else:
                    print("ignore " + key)
                    if not ignore_missing:
                        raise ValueError('no variable %s in this network' % full)
        if session is not None:
            session.load_arrays(staged)
        else:
            self._staged_arrays = dict(getattr(self, '_staged_arrays', {}), **staged)

    def feed(self, *args):
        assert len(args) != 0
        self.inputs = []
        for lyr in args:
            if isinstance(lyr, str):
                try:
                    lyr = self.layers[lyr]
                except KeyError:
                    print(list(self.layers.keys()))
                    raise KeyError('Unknown layer name fed: %s' % lyr)
            self.inputs.append(lyr)
        return self

    def get_output(self, layer):
        try:
            layer = self.layers[layer]
        except KeyError:
            print(list(self.layers.keys()))
            raise KeyError('Unknown layer name fed: %s' % layer)
        return layer

    def get_unique_name(self, prefix):
        id = sum(t.startswith(prefix) for t, _ in list(self.layers.items())) + 1
        return '%s_%d' % (prefix, id)

    def make_var(self, name, shape, initializer=None, trainable=True, regularizer=None):
        """Registers a parameter `name` (already scope-qualified) in the plan."""
        spec = ParamSpec(name, shape, initializer, trainable, regularized=regularizer is not None)
        self.param_specs[name] = spec
        return spec

    def validate_padding(self, padding):
        assert padding in ('SAME', 'VALID')

    def l2_regularizer(self, weight_decay=0.0005, scope=None):
        """Marker object: variables created with it join the L2 term wd * sum(w^2)/2 (network.py:630-637)."""
        return ('l2', weight_decay)

    # ------------------------------------------------------------------------------------------ layers
    @layer
    def bi_lstm(self, input, num_hids, num_layers, name, img_shape=None, trainable=True, honour_num_layers=False):
        """honour_num_layers=False reproduces the reference (num_layers accepted and ignored, network.py:98,111-115).
        With honour_num_layers=True (new configurations, e.g. BASELINE configs[4]) num_layers BiLSTMs are stacked: the
        hidden layers feed their concatenated [N, T, num_hids] output forward, only the last one carries the FC."""
        from .config import cfg
        img, img_len = input[0], input[1]
        if honour_num_layers:
            for li in range(max(1, num_layers) - 1):
                din_l = img.channels
                u_l = num_hids // 2
                hname = '%s/stack%d' % (name, li)
                for d in ('fw', 'bw'):
                    self.make_var('%s/%s/weights' % (hname, d), [din_l + u_l, 4 * u_l], 'glorot_uniform', trainable)
                    self.make_var('%s/%s/biases' % (hname, d), [4 * u_l], 'zeros', trainable)
                img = Node('bi_lstm', hname, [img, img_len], num_hids=num_hids, num_layers=1, nclasses=cfg.NCLASSES,
                           din=din_l, with_fc=False, channels=num_hids)
                self.layers[hname] = img
        din = img.channels
        u = num_hids // 2
        for d in ('fw', 'bw'):
            self.make_var('%s/%s/weights' % (name, d), [din + u, 4 * u], 'glorot_uniform', trainable)
            self.make_var('%s/%s/biases' % (name, d), [4 * u], 'zeros', trainable)
        self.make_var(name + '/weights', [num_hids, cfg.NCLASSES], ('variance_scaling', 0.01), trainable,
                      regularizer=self.l2_regularizer(cfg.TRAIN.WEIGHT_DECAY))
        self.make_var(name + '/biases', [cfg.NCLASSES], 'zeros', trainable)
        # num_layers is accepted and ignored, exactly like network.py:98,111-115
        return Node('bi_lstm', name, [img, img_len], num_hids=num_hids, num_layers=num_layers,
                    nclasses=cfg.NCLASSES, din=din, channels=cfg.NCLASSES)

    @layer
    def lstm(self, input, num_hids, num_layers, name, img_shape=None, trainable=True):
        """Unidirectional stacked LSTM + FC (network.py:130-152): MultiRNNCell of num_layers LSTMCell(num_hids) under
        tf.nn.dynamic_rnn (forward in time, outputs zero past each length), then logits = h W + b, time-major.  Variable names
        as TF 1.0 creates them: <name>/rnn/multi_rnn_cell/cell_<i>/lstm_cell/{weights [D+U, 4U], biases [4U]}, <name>/weights
        (truncated_normal(0.1), L2-regularised), <name>/biases."""
        from .config import cfg
        img, img_len = input[0], input[1]
        nl = max(1, int(num_layers))
        for li in range(nl):
            din = img.channels
            cell = '%s/rnn/multi_rnn_cell/cell_%d/lstm_cell' % (name, li)
            self.make_var(cell + '/weights', [din + num_hids, 4 * num_hids], 'glorot_uniform', trainable)
            self.make_var(cell + '/biases', [4 * num_hids], 'zeros', trainable)
            last = li == nl - 1
            if last:
                self.make_var(name + '/weights', [num_hids, cfg.NCLASSES], ('truncated_normal', 0.1), trainable,
                              regularizer=self.l2_regularizer(cfg.TRAIN.WEIGHT_DECAY))
                self.make_var(name + '/biases', [cfg.NCLASSES], 'zeros', trainable)
            # a hidden layer is named like its cell's variable scope, so that layout.layer_of (longest layer-name prefix) attributes
            # the cell's variables to the op that produces their gradients
            lname = name if last else '%s/rnn/multi_rnn_cell/cell_%d' % (name, li)
            node = Node('lstm', lname, [img, img_len], num_hids=num_hids, num_layers=1, nclasses=cfg.NCLASSES, din=din,
                        with_fc=last, cells=[cell], fc=name, channels=cfg.NCLASSES if last else num_hids)
            if not last:
                self.layers[lname] = node
            img = node
        return img

    @layer
    def conv_single(self, input, k_h, k_w, c_o, s_h, s_w, name, c_i=None, bn=False, biased=True, relu=True,
                    padding=DEFAULT_PADDING, trainable=True):
        from .config import cfg
        self.validate_padding(padding)
        if not c_i:
            c_i = input.channels
        if c_i is None:
            raise ValueError('conv_single %s: input channel count unknown, pass c_i' % name)
        self.make_var(name + '/weights', [k_h, k_w, c_i, c_o], 'xavier_uniform', trainable,
                      regularizer=self.l2_regularizer(cfg.TRAIN.WEIGHT_DECAY))
        if biased:
            self.make_var(name + '/biases', [c_o], 'zeros', trainable)
        if bn:
            # tf.contrib.layers.batch_norm(scope=name) nested in variable_scope(name) -> name/name/{beta,gamma}
            self.make_var('%s/%s/beta' % (name, name), [c_o], 'zeros', trainable)
            self.make_var('%s/%s/gamma' % (name, name), [c_o], 'ones', trainable)
        return Node('conv', name, [input], k_h=k_h, k_w=k_w, c_o=c_o, s_h=s_h, s_w=s_w, c_i=c_i, bn=bn,
                    biased=biased, relu=relu, padding=padding, channels=c_o)

    @layer
    def conv(self, input, k_h, k_w, c_o, s_h, s_w, name, c_i=None, biased=True, relu=True,
             padding=DEFAULT_PADDING, trainable=True):
        from .config import cfg
        self.validate_padding(padding)
        if not c_i:
            c_i = input.channels
        self.make_var(name + '/weights', [k_h, k_w, c_i, c_o], 'xavier_uniform', trainable,
                      regularizer=self.l2_regularizer(cfg.TRAIN.WEIGHT_DECAY))
        if biased:
            self.make_var(name + '/biases', [c_o], 'zeros', trainable)
        return Node('conv', name, [input], k_h=k_h, k_w=k_w, c_o=c_o, s_h=s_h, s_w=s_w, c_i=c_i, bn=False,
                    biased=biased, relu=relu, padding=padding, channels=c_o)

    @layer
    def relu(self, input, name):
        return Node('relu', name, [input], channels=input.channels)

    @layer
    def max_pool(self, input, k_h, k_w, s_h, s_w, name, padding=DEFAULT_PADDING):
        self.validate_padding(padding)
        return Node('max_pool', name, [input], k_h=k_h, k_w=k_w, s_h=s_h, s_w=s_w, padding=padding,
                    channels=input.channels)

    @layer
    def avg_pool(self, input, k_h, k_w, s_h, s_w, name, padding=DEFAULT_PADDING):
        self.validate_padding(padding)
        return Node('avg_pool', name, [input], k_h=k_h, k_w=k_w, s_h=s_h, s_w=s_w, padding=padding,
                    channels=input.channels)

    @layer
    def reshape_squeeze_layer(self, input, d, name):
        # N,H,W,C -> N,H*W,C
        return Node('reshape_squeeze', name, [input], d=int(d), channels=int(d))

    @layer
    def concat(self, input, axis, name):
        ch = [i.channels for i in input]
        return Node('concat', name, list(input), axis=axis, channels=(sum(ch) if axis in (3, -1) and all(c is not None for c in ch) else
                                                                     input[0].channels))

    @layer
    def add(self, input, name):
        return Node('add', name, list(input), channels=input[0].channels)

    @layer
    def batch_normalization(self, input, name, relu=True, is_training=False):
        """tf.contrib.layers.batch_norm(scale, center, is_training, scope=name) [+ ReLU] — network.py:466-473.  is_training=False
        (the reference's default) normalises with the stored moving statistics, which the reference never updates (no
        UPDATE_OPS dependency: SURVEY Q2), i.e. moving_mean = 0, moving_variance = 1 unless a checkpoint says otherwise."""
        c = input.channels
        self.make_var('%s/beta' % name, [c], 'zeros')
        self.make_var('%s/gamma' % name, [c], 'ones')
        self.make_var('%s/moving_mean' % name, [c], 'zeros', trainable=False)
        self.make_var('%s/moving_variance' % name, [c], 'ones', trainable=False)
        return Node('batch_norm', name, [input], relu=relu, is_training=bool(is_training), channels=c)

    @layer
    def dropout(self, input, keep_prob, name):
        # the shipped graphs leave dropout commented out (LSTM_train.py:35); kept as an identity slot
        return Node('dropout', name, [input], keep_prob=keep_prob, channels=input.channels)

    @layer
    def fc(self, input, num_out, name, relu=True, trainable=True):
        """tf.nn.relu_layer / xw_plus_b over the LAST axis — network.py:415-447: `weights [dim, num_out]` truncated-normal(0.01) (0.001 for
        'bbox_pred'), L2-regularised, `biases [num_out]` zeros.  Lowered as the GEMMs of a 1 x 1 convolution (igemm forward / data gradient, the
        plain weight-gradient product).  What is covered is the reference's non-4-D branch (`feed_in, dim = input, input_shape[-1]`) on the row
        tensors this graph has: the [N, T, d] output of reshape_squeeze_layer (the same per-time-step product bi_lstm does itself at
        network.py:118-126), possibly behind dropout / relu / another fc.  A 4-D feature map is refused: the reference flattens the WHOLE static
        map there (`dim = prod(input_shape[1:])` after a transpose), and this graph's width is dynamic (placeholder [None, None, 32],
        LSTM_train.py:10) — TensorFlow raises on the `None` dimension as well."""
        from .config import cfg
        if isinstance(input, (list, tuple)):        # "only use the first input" (network.py:418-419)
            input = input[0]
        nd = input
        while nd.op in ('fc', 'dropout', 'relu'):
            nd = nd.inputs[0]
        if nd.op != 'reshape_squeeze':
            raise NotImplementedError('fc %r: the input must be a row tensor [N, T, d] (reshape_squeeze_layer, optionally behind dropout / relu / fc); '
                                      'a 4-D map of dynamic width has no static flattened size (network.py:421-425)' % name)
        dim = input.channels
        self.make_var(name + '/weights', [dim, num_out], ('truncated_normal', 0.001 if name == 'bbox_pred' else 0.01), trainable,
                      regularizer=self.l2_regularizer(cfg.TRAIN.WEIGHT_DECAY))
        self.make_var(name + '/biases', [num_out], 'zeros', trainable)
        return Node('fc', name, [input], k_h=1, k_w=1, c_o=num_out, s_h=1, s_w=1, c_i=dim, bn=False, biased=True, relu=relu, padding='VALID',
                    channels=num_out)

    @layer
    def softmax(self, input, name):
        return Node('softmax', name, [input], channels=input.channels)

    # ------------------------------------------------------------------------------------------ loss
    def build_loss(self):
        """Returns (loss, dense_decoded) plan nodes — warp-ctc mean cost + L2 terms, and the decoded labels
        (network.py:647-664).  The engine evaluates them in `train_step` / `decode`."""
        time_step_batch = self.get_output('time_step_len')
        logits_batch = self.get_output('logits')
        labels = self.get_output('labels')
        label_len = self.get_output('labels_len')
        loss = Node('ctc_loss', 'loss', [logits_batch, labels, label_len, time_step_batch])
        dense_decoded = Node('ctc_decode', 'dense_decoded', [logits_batch, time_step_batch])
        self.layers['loss'] = loss
        self.layers['dense_decoded'] = dense_decoded
        return loss, dense_decoded
