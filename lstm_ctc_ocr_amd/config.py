"""Global configuration of the OCR path.

Contract taken from the reference's lib/lstm/config.py (keys and defaults :10-72, label maps :73-81, output / log directory
helpers :84-97, merge rules :99-126, file and list overrides :128-156): the same key tree, the same default values, the same
exceptions for bad overrides — user scripts and lstm/lstm.yml written for the reference keep working.  The implementation is
independent: the defaults are ONE nested table, and a single walker does both kinds of override.  Differences forced by this
environment: yaml.safe_load (PyYAML >= 6 rejects a bare yaml.load), an in-tree EasyDict when the PyPI one is
absent, ROOT_DIR overridable with $OCR_ROOT_DIR so outputs need not land inside the package.
"""
import ast
import os
import time

import numpy as np

try:                                    # the real package if the host has it, else the in-tree equivalent
    from easydict import EasyDict as edict
except ImportError:                     # pragma: no cover - depends on the host
    from .edict import EasyDict as edict

_ALPHABET = '0123456789' + 'abcdefghijklmnopqrstuvwxyz' + 'ABCDEFGHIJKLMNOPQRSTUVWXYZ'
_HEIGHT, _PLANES = 32, 1

_DEFAULTS = {
    # device / data geometry
    'GPU_ID': 1, 'GPU_USAGE': 0.9,
    'IMG_SHAPE': [32, 100], 'IMG_HEIGHT': _HEIGHT, 'NCHANNELS': _PLANES, 'NUM_FEATURES': _HEIGHT * _PLANES,
    'POOL_SCALE': 4,                    # two 2x2 pools along the width
    'OFFSET_TIME_STEP': -1,             # the 2x2 VALID conv5 eats one more step: T = W // POOL_SCALE + OFFSET_TIME_STEP
    # label alphabet: classes 1..62 are characters, 0 is blank / padding, 63 is the TF decoder's blank
    'CHARSET': _ALPHABET, 'NCLASSES': len(_ALPHABET) + 2, 'BLANK_TOKEN': 0, 'SPACE_INDEX': 0, 'SPACE_TOKEN': '',
    'MAX_CHAR_LEN': 6, 'MIN_LEN': 4, 'MAX_LEN': 6, 'FONT': 'fonts/Ubuntu-M.ttf',
    # bookkeeping
    'NET_NAME': 'lstm', 'RNG_SEED': 3, 'EXP_DIR': 'default', 'LOG_DIR': 'default',
    'ROOT_DIR': os.environ.get('OCR_ROOT_DIR', os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))),
    'TRAIN': {
        'SOLVER': 'Adam',               # Adam | Momentum | RMS
        'LEARNING_RATE': 0.01, 'MOMENTUM': 0.9, 'WEIGHT_DECAY': 0.0005, 'GAMMA': 0.1, 'STEPSIZE': 50000,
        'BATCH_SIZE': 64, 'NUM_EPOCHS': 2000, 'NUM_HID': 512, 'NUM_LAYERS': 2,
        'DISPLAY': 10, 'LOG_IMAGE_ITERS': 100,
        'SNAPSHOT_ITERS': 5000, 'SNAPSHOT_PREFIX': 'lstm', 'SNAPSHOT_INFIX': '',
        'TXT': 'annotation_train.txt',
    },
    'VAL': {'BATCH_SIZE': 128, 'VAL_STEP': 1000, 'NUM_EPOCHS': 1000, 'PRINT_NUM': 5, 'TXT': 'annotation_val.txt'},
    'TEST': {},
}


def _tree(table):
    node = edict()
    for key, value in table.items():
        node[key] = _tree(value) if isinstance(value, dict) else value
    return node


cfg = _tree(_DEFAULTS)
__C = cfg                               # the reference's private alias, kept for code that imports it


# ------------------------------------------------------------------------------------------------ label maps
def get_encode_decode_dict():
    """(char -> class, class -> char): characters are numbered from 1 in CHARSET order; '' <-> 0 is blank / padding."""
    chars = list(cfg.CHARSET)
    encode = dict(zip(chars, range(1, len(chars) + 1)))
    decode = {index: char for char, index in encode.items()}
    encode[cfg.SPACE_TOKEN] = cfg.SPACE_INDEX
    decode[cfg.SPACE_INDEX] = cfg.SPACE_TOKEN
    return encode, decode


# ------------------------------------------------------------------------------------------------ directories
def _made(path):
    os.makedirs(path, exist_ok=True)
    return path


def get_output_dir(imdb, weights_filename):
    """<ROOT_DIR>/output/<EXP_DIR>[/<weights_filename>], created on demand (snapshots go here)."""
    parts = [cfg.ROOT_DIR, 'output', cfg.EXP_DIR] + ([weights_filename] if weights_filename is not None else [])
    return _made(os.path.abspath(os.path.join(*parts)))


def get_log_dir(imdb):
    """<ROOT_DIR>/logs/<LOG_DIR>/<imdb.name>/<timestamp>, created on demand."""
    stamp = time.strftime('%Y-%m-%d-%H-%M-%S', time.localtime())
    return _made(os.path.abspath(os.path.join(cfg.ROOT_DIR, 'logs', cfg.LOG_DIR, imdb.name, stamp)))


# ------------------------------------------------------------------------------------------------ overrides
def _merge_a_into_b(a, b):
    """Overlay tree `a` on the configuration tree `b`.  Only existing keys may be set (KeyError otherwise) and a value must
    keep the type of the default (ValueError otherwise; array defaults adopt lists through their dtype)."""
    if type(a) is not edict:
        return
    for key in a:
        if key not in b:
            raise KeyError('{} is not a valid config key'.format(key))
        new, old = a[key], b[key]
        if type(new) is not type(old):
            if not isinstance(old, np.ndarray):
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(new), key))
            new = np.array(new, dtype=old.dtype)
        if type(new) is not edict:
            b[key] = new
            continue
        try:
            _merge_a_into_b(new, old)
        except Exception:
            print('Error under config key: {}'.format(key))
            raise


def cfg_from_file(filename):
    """Overlay a YAML file (e.g. lstm/lstm.yml) on the defaults."""
    import yaml
    with open(filename) as stream:
        _merge_a_into_b(edict(yaml.safe_load(stream)), cfg)


def cfg_from_list(cfg_list):
    """Overlay `KEY.SUBKEY value` pairs given as a flat list (the command line's --set)."""
    if len(cfg_list) % 2:
        raise AssertionError('--set takes KEY VALUE pairs')
    for dotted, text in zip(cfg_list[::2], cfg_list[1::2]):
        *path, leaf = dotted.split('.')
        node = cfg
        for part in path:
            assert part in node, dotted
            node = node[part]
        assert leaf in node, dotted
        try:
            value = ast.literal_eval(text)
        except Exception:               # not a Python literal: a plain string
            value = text
        assert type(value) == type(node[leaf]), 'type {} does not match original type {}'.format(type(value), type(node[leaf]))
        node[leaf] = value
