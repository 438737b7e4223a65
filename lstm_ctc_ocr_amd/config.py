"""Global configuration — same keys, defaults, merge rules and helpers as /root/reference/lib/lstm/config.py
(cfg :10-72, get_encode_decode_dict :73-81, get_output_dir :84-90, get_log_dir :92-97, _merge_a_into_b
:99-126, cfg_from_file :128-134, cfg_from_list :136-156).  Differences, all forced by this environment and
listed in INTEGRATION.md: yaml.safe_load (PyYAML >= 6 rejects yaml.load without a Loader), an in-tree
EasyDict, and ROOT_DIR overridable with $OCR_ROOT_DIR so outputs need not land inside the package."""
import os
import os.path as osp
from time import localtime, strftime

import numpy as np

try:                                    # the real package if the host has it, else the in-tree equivalent
    from easydict import EasyDict as edict
except ImportError:                     # pragma: no cover - depends on the host
    from .edict import EasyDict as edict

__C = edict()
cfg = __C

__C.GPU_ID = 1
__C.GPU_USAGE = 0.9
__C.OFFSET_TIME_STEP = -1
__C.POOL_SCALE = 4
__C.IMG_SHAPE = [32, 100]
__C.IMG_HEIGHT = 32
__C.MAX_CHAR_LEN = 6
__C.BLANK_TOKEN = 0
__C.CHARSET = '0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'
__C.NCLASSES = len(__C.CHARSET) + 2
__C.MIN_LEN = 4
__C.MAX_LEN = 6
__C.FONT = 'fonts/Ubuntu-M.ttf'
__C.NCHANNELS = 1
__C.NUM_FEATURES = __C.IMG_HEIGHT * __C.NCHANNELS

__C.NET_NAME = 'lstm'
__C.TRAIN = edict()
__C.TRAIN.SOLVER = 'Adam'              # Adam | Momentum | RMS
__C.TRAIN.TXT = 'annotation_train.txt'
__C.TRAIN.WEIGHT_DECAY = 0.0005
__C.TRAIN.LEARNING_RATE = 0.01
__C.TRAIN.MOMENTUM = 0.9
__C.TRAIN.GAMMA = 0.1
__C.TRAIN.STEPSIZE = 50000
__C.TRAIN.DISPLAY = 10
__C.TRAIN.LOG_IMAGE_ITERS = 100
__C.TRAIN.NUM_EPOCHS = 2000
__C.TRAIN.NUM_HID = 512
__C.TRAIN.NUM_LAYERS = 2
__C.TRAIN.BATCH_SIZE = 64
__C.TRAIN.SNAPSHOT_ITERS = 5000
__C.TRAIN.SNAPSHOT_PREFIX = 'lstm'
__C.TRAIN.SNAPSHOT_INFIX = ''

__C.VAL = edict()
__C.VAL.TXT = 'annotation_val.txt'
__C.VAL.VAL_STEP = 1000
__C.VAL.NUM_EPOCHS = 1000
__C.VAL.BATCH_SIZE = 128
__C.VAL.PRINT_NUM = 5

__C.RNG_SEED = 3
__C.ROOT_DIR = os.environ.get('OCR_ROOT_DIR', osp.abspath(osp.join(osp.dirname(__file__), '..')))
__C.TEST = edict()
__C.EXP_DIR = 'default'
__C.LOG_DIR = 'default'
__C.SPACE_INDEX = 0
__C.SPACE_TOKEN = ''


def get_encode_decode_dict():
    """char -> 1..len(CHARSET) by CHARSET position; '' <-> 0 (the CTC blank / padding value)."""
    encode_maps, decode_maps = {}, {}
    for i, char in enumerate(__C.CHARSET, 1):
        encode_maps[char] = i
        decode_maps[i] = char
    encode_maps[__C.SPACE_TOKEN] = __C.SPACE_INDEX
    decode_maps[__C.SPACE_INDEX] = __C.SPACE_TOKEN
    return encode_maps, decode_maps


def _ensure_dir(path):
    if not os.path.exists(path):
        os.makedirs(path)
    return path


def get_output_dir(imdb, weights_filename):
    outdir = osp.abspath(osp.join(__C.ROOT_DIR, 'output', __C.EXP_DIR))
    if weights_filename is not None:
        outdir = osp.join(outdir, weights_filename)
    return _ensure_dir(outdir)


def get_log_dir(imdb):
    stamp = strftime("%Y-%m-%d-%H-%M-%S", localtime())
    return _ensure_dir(osp.abspath(osp.join(__C.ROOT_DIR, 'logs', __C.LOG_DIR, imdb.name, stamp)))


def _merge_a_into_b(a, b):
    """Strict recursive merge: unknown key -> KeyError, type mismatch -> ValueError (numpy arrays are cast)."""
    if type(a) is not edict:
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            else:
                raise ValueError(('Type mismatch ({} vs. {}) for config key: {}').format(type(b[k]), type(v), k))
        if type(v) is edict:
            try:
                _merge_a_into_b(a[k], b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a YAML config file and merge it into the defaults."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = edict(yaml.safe_load(f))
    _merge_a_into_b(yaml_cfg, __C)


def cfg_from_list(cfg_list):
    """Set config keys from a flat [key, value, key, value, ...] list (the CLI's --set)."""
    from ast import literal_eval
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split('.')
        d = __C
        for subkey in key_list[:-1]:
            assert subkey in d
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d
        try:
            value = literal_eval(v)
        except Exception:
            value = v                   # plain string
        assert type(value) == type(d[subkey]), \
            'type {} does not match original type {}'.format(type(value), type(d[subkey]))
        d[subkey] = value
