"""Reader for TensorFlow "tensor bundle" checkpoints (Saver write_version V2: `<prefix>.index` + `<prefix>.data-0000x-of-0000y`) WITHOUT
TensorFlow, and a converter to this package's `.npz` snapshots — so that a model trained with the reference
(`/root/reference/lib/lstm/train.py:18,23-37`: `tf.train.Saver(max_to_keep=100)` → `..._iter_<n>.ckpt.{index,data-*,meta}`) can be loaded
by `test_net` / resumed by `train_net` here.

STATUS: written from the published format (leveldb table format as used by tensorflow/core/lib/io/table, tensorflow/core/protobuf/
tensor_bundle.proto); no TensorFlow-written file exists in this environment, so the reader is pinned only on bundles produced by the
writer below (tests/test_tf_bundle.py) — treat a first real import as unverified and compare `--list` output with the graph.

Format, as read here:
  .index  = sorted string table.  Footer (last 48 bytes): metaindex handle, index handle (each: varint64 offset, varint64 size), zero
            padding, magic 0xdb4775248b80fb57 (little endian).  Block = entries {varint32 shared, varint32 unshared, varint32 value_len,
            key suffix, value} + uint32 restart offsets + uint32 restart count, followed by a 5-byte trailer {compression type: 0 none,
            1 snappy; masked crc32c}.  The index block maps separator keys to data-block handles.
            key ""  -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
            key var -> BundleEntryProto {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c, 7: slices}
  .data-* = raw little-endian tensor bytes at [offset, offset + size).
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16}


# ------------------------------------------------------------------------------------------------ wire helpers
def _varint(buf, pos):
    shift, out = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) — the checksum of the table blocks and of every tensor."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    tab, c = _CRC_TABLE, crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    """leveldb / TensorFlow store crc32c rotated right by 15 bits plus a constant."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _proto_fields(buf):
    """Generic protobuf wire parser: yields (field number, wire type, value) — value is an int or a bytes object."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, v


def _snappy_decompress(buf):
    """Minimal raw-snappy decoder (block compression type 1 of the table format)."""
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = struct.unpack_from('<H', buf, pos)[0]; pos += 2
        else:
            ln = (tag >> 2) + 1
            off = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(ln):                             # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: length mismatch')
    return bytes(out)


# ------------------------------------------------------------------------------------------------ table reader
ALLOW_UNCHECKED_BLOCKS = False


def _read_block(data, offset, size):
    raw = data[offset:offset + size]
    ctype = data[offset + size]                         # trailer: type byte + 4 bytes of masked crc32c over contents + type byte
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    # TensorFlow always writes block checksums: a zero trailer is corruption too (ALLOW_UNCHECKED_BLOCKS is for hand-built test tables)
    if not (ALLOW_UNCHECKED_BLOCKS and stored == 0) and stored != masked_crc(data[offset:offset + size + 1]):
        raise ValueError('table block at %d: checksum mismatch — not a table file, or a misread handle' % offset)
    if ctype == 1:
        raw = _snappy_decompress(raw)
    elif ctype != 0:
        raise ValueError('unknown block compression type %d' % ctype)
    return raw


def _block_entries(block):
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared]); pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path):
    """All (key, value) pairs of a leveldb-format table file, in key order."""
    data = open(path, 'rb').read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError('%s is not a tensor-bundle index (bad table magic)' % path)
    footer = data[-48:]
    _, p = _varint(footer, 0); _, p = _varint(footer, p)           # metaindex handle (unused)
    ioff, p = _varint(footer, p); isize, p = _varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize)):
        boff, q = _varint(handle, 0); bsize, q = _varint(handle, q)
        out.extend(_block_entries(_read_block(data, boff, bsize)))
    return out


# ------------------------------------------------------------------------------------------------ bundle reader
class BundleEntry(object):
    def __init__(self, name, dtype, shape, shard, offset, size, sliced, crc=0):
        self.name, self.dtype, self.shape, self.shard, self.offset, self.size, self.sliced = name, dtype, shape, shard, offset, size, sliced
        self.crc = crc


def _parse_entry(name, value):
    dtype, shape, shard, offset, size, sliced, crc = 0, [], 0, 0, 0, False, 0
    for f, wt, v in _proto_fields(value):
        if f == 1: dtype = v
        elif f == 2:
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:                              # dim
                    d = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1: d = v3
                    shape.append(d)
        elif f == 3: shard = v
        elif f == 4: offset = v
        elif f == 5: size = v
        elif f == 6: crc = v
        elif f == 7: sliced = True
    return BundleEntry(name, dtype, tuple(shape), shard, offset, size, sliced, crc)


def read_bundle(prefix, names=None, verify=False):
    """{variable name: numpy array} of the checkpoint `prefix` (the path WITHOUT .index / .data-...).  verify=True checks every tensor's
    stored crc32c (pure Python: ~1 s per MB) — on a TensorFlow-written file a full pass without a mismatch proves the offsets were read right."""
    table = read_table(prefix + '.index')
    num_shards = 1
    entries = []
    for key, value in table:
        if key == b'':
            for f, _, v in _proto_fields(value):
                if f == 1: num_shards = v
                if f == 2 and v != 0: raise ValueError('big-endian bundle')
            continue
        entries.append(_parse_entry(key.decode('utf-8'), value))
    out, files = {}, {}
    for e in entries:
        if names is not None and e.name not in names:
            continue
        if e.sliced:
            raise ValueError('%s: partitioned (sliced) variables are not supported' % e.name)
        if e.dtype not in DTYPES:
            raise ValueError('%s: unsupported dtype enum %d' % (e.name, e.dtype))
        if e.shard not in files:
            files[e.shard] = np.memmap('%s.data-%05d-of-%05d' % (prefix, e.shard, num_shards), dtype=np.uint8, mode='r')
        dt = np.dtype(DTYPES[e.dtype])
        n = int(np.prod(e.shape)) if e.shape else 1
        if n * dt.itemsize != e.size:
            raise ValueError('%s: size %d does not match shape %s of %s' % (e.name, e.size, e.shape, dt))
        raw = np.asarray(files[e.shard][e.offset:e.offset + e.size])
        if verify and e.crc != 0 and e.crc != masked_crc(raw.tobytes()):
            raise ValueError('%s: tensor checksum mismatch' % e.name)
        out[e.name] = raw.view(dt).reshape(e.shape).copy()
    return out


def list_bundle(prefix):
    return [(e.name, DTYPES.get(e.dtype, e.dtype), e.shape) for e in
            (_parse_entry(k.decode('utf-8'), v) for k, v in read_table(prefix + '.index') if k != b'')]


# ------------------------------------------------------------------------------------------------ writer (tests; same format, one block per `per_block` keys)
def _trailer(block, ctype=0):
    return bytes([ctype]) + struct.pack('<I', masked_crc(block + bytes([ctype])))


def _build_block(items, restart_interval=16):
    buf, restarts, prev, cnt = bytearray(), [], b'', 0
    for key, value in items:
        if cnt % restart_interval == 0:
            restarts.append(len(buf)); shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        prev, cnt = key, cnt + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack('<I', r)
    buf += struct.pack('<I', len(restarts))
    return bytes(buf)


def _proto(fields):
    out = bytearray()
    for f, wt, v in fields:
        out += _put_varint((f << 3) | wt)
        if wt == 0: out += _put_varint(v)
        elif wt == 2: out += _put_varint(len(v)) + v
        elif wt == 5: out += struct.pack('<I', v)
    return bytes(out)


def write_bundle(prefix, arrays, per_block=3, tensor_crc=True):
    """Writes {name: array} as a one-shard bundle (uncompressed blocks, real checksums).  Test infrastructure for the reader."""
    enum = {np.dtype(v): k for k, v in DTYPES.items()}
    data, items = bytearray(), [(b'', _proto([(1, 0, 1), (2, 0, 0), (3, 2, _proto([(1, 0, 1)]))]))]
    for name in sorted(arrays):
        a = np.asarray(arrays[name])
        a = np.ascontiguousarray(a) if a.ndim else a          # (ascontiguousarray would turn a scalar into shape (1,))
        shape = _proto([(2, 2, _proto([(1, 0, int(d))])) for d in a.shape])
        items.append((name.encode('utf-8'), _proto([(1, 0, enum[a.dtype]), (2, 2, shape), (3, 0, 0), (4, 0, len(data)), (5, 0, a.nbytes), (6, 5, masked_crc(a.tobytes()) if tensor_crc else 0)])))
        data += a.tobytes()
    with open('%s.data-00000-of-00001' % prefix, 'wb') as f:
        f.write(bytes(data))
    out, index_items = bytearray(), []
    for i in range(0, len(items), per_block):
        blk = _build_block(items[i:i + per_block], restart_interval=2)
        index_items.append((items[min(i + per_block, len(items)) - 1][0], _put_varint(len(out)) + _put_varint(len(blk))))
        out += blk + _trailer(blk)
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out += meta + _trailer(meta)
    idx = _build_block(index_items, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out += idx + _trailer(idx)
    footer = meta_handle + idx_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out) + footer)


# ------------------------------------------------------------------------------------------------ conversion to this package's snapshots
# TF-1.0 scopes that this package's variable names do not carry (network.py:97-152: bidirectional_dynamic_rnn / MultiRNNCell / LSTMCell)
_STRIP = [r'bidirectional_rnn/', r'lstm_cell/', r'basic_lstm_cell/']


def normalise_name(name):
    """`logits/bidirectional_rnn/fw/lstm_cell/weights` -> `logits/fw/weights` (this package's name for the same variable)."""
    for pat in _STRIP:
        name = re.sub(pat, '', name)
    return name


def convert(prefix, out_path, wanted=None, iteration=None, verify=False, output_dir=None, restore_lr=False):
    """TensorFlow checkpoint `prefix` -> `.npz` snapshot for checkpoint.restore().  `wanted` = the engine's variable names (Engine.specs);
    variables are matched by exact name, then by name with the RNN helper scopes removed.  Adam slots (`<var>/Adam`, `<var>/Adam_1`,
    `beta1_power`, `beta2_power`) become slot1 / slot2 / optimiser scalars; batch-norm moving averages are dropped (the reference never
    uses them: network.py:176-178 runs batch norm with is_training=True everywhere).  Returns (matched, unmatched TF names).
    EXPERIMENTAL (pinned only on bundles of this module's own writer: no TensorFlow-written file exists offline).
    The reference builds its `tf.train.Saver` in `SolverWrapper.__init__` (lib/lstm/train.py:18), BEFORE `lr`, `global_step` and the Adam
    slots exist (train.py:73-85), so the checkpoints the reference itself writes hold NONE of them, and its own restore re-derives the
    iteration from the file name (train.py:100-104).  A checkpoint written by a Saver created later (or by other TF code for the same
    graph) may hold the unnamed `Variable` (float scalar: learning rate) / `Variable_1` or `global_step` (integer scalar: step count) and
    the Adam slots; they are carried over as follows.  Step count: the integer step variable, else the value derived from `beta2_power`
    (exact up to ~87 000 steps) / `beta1_power`, and only then the `_iter_<n>` of the file name.  Learning rate: the driver's configured
    rate is KEPT, as the reference does (it never restores `lr`); `restore_lr=True` opts into taking the checkpoint's `Variable`.
    output_dir: write the snapshot as `<output_dir>/<basename>_iter_<n>.ckpt` instead of `out_path` and register it in that
    directory's `checkpoint` index, so that test_net / train_net --restore find it (checkpoint.latest_checkpoint)."""
    tf_vars = read_bundle(prefix, verify=verify)
    by_norm = {}
    for k in tf_vars:
        by_norm.setdefault(normalise_name(k), k)
    names = list(wanted) if wanted is not None else [k for k in tf_vars if not re.search(r'/Adam(_1)?$|_power$|moving_(mean|variance)$|/Momentum$|/RMSProp(_1)?$', k)]
    arrays, used = {}, set()
    for name in names:
        src = name if name in tf_vars else by_norm.get(name)
        if src is None:
            continue
        arrays['var/' + name] = tf_vars[src].astype(np.float32)
        used.add(src)
        for slot, suffix in (('slot1/', '/Adam'), ('slot2/', '/Adam_1')):
            if src + suffix in tf_vars:
                arrays[slot + name] = tf_vars[src + suffix].astype(np.float32)
                used.add(src + suffix)
    if 'beta1_power' in tf_vars and 'beta2_power' in tf_vars and any(k.startswith('slot1/') for k in arrays):
        # Adam's step count: the reference's global_step variable (train.py:71,85) if it was saved; else from the float32 powers Adam
        # itself multiplied up — beta2^t first (0.9^t underflows float32 after ~830 steps, 0.999^t after ~87 000).  TF keeps beta^t in
        # FLOAT32 and multiplies by float32(0.999) = 0.99900001287 each step, so the logarithm is taken to THAT base: to the base 0.999 the
        # estimate under-counts by 1.3e-5 per step and rounds wrongly from step 38 539 on — inside the reference's default 40 000-iteration
        # run.  To the float32 base a float32 simulation of the product is off by < 0.01 of a step all the way to the underflow
        # (tests/test_tf_bundle.py replays it at 40 000 and 86 000 steps).  The device keeps the powers in double.
        b1t, b2t = float(np.asarray(tf_vars['beta1_power']).reshape(-1)[0]), float(np.asarray(tf_vars['beta2_power']).reshape(-1)[0])
        step_var = next((k for k in ('global_step', 'Variable_1') if k in tf_vars and np.asarray(tf_vars[k]).size == 1
                         and np.issubdtype(np.asarray(tf_vars[k]).dtype, np.integer)), None)
        if step_var is not None:
            step = int(np.asarray(tf_vars[step_var]).reshape(-1)[0])
            used.add(step_var)
        elif 0 < b2t < 1:
            step = int(round(np.log(b2t) / np.log(float(np.float32(0.999)))))
        elif 0 < b1t < 1:
            step = int(round(np.log(b1t) / np.log(float(np.float32(0.9)))))
        elif re.search(r'_iter_(\d+)', os.path.basename(prefix)):        # both powers underflowed: the Saver's file name (train.py:27-36)
            step = max(0, int(re.search(r'_iter_(\d+)', os.path.basename(prefix)).group(1)) - 1)
        else:
            step = 0
        sc = np.zeros(8, np.float64)                    # csrc/optim.hip: [2] lr (set by the driver), [4] beta1^t, [5] beta2^t, [6] step
        sc[4], sc[5], sc[6] = 0.9 ** step, 0.999 ** step, step
        lr_var = tf_vars.get('Variable')                # an unnamed float scalar = the learning-rate variable of train.py:73
        if lr_var is not None and np.asarray(lr_var).size == 1 and np.issubdtype(np.asarray(lr_var).dtype, np.floating):
            if restore_lr:                              # opt-in: checkpoint.restore() sets engine.lr from sc[2] > 0; by default the driver's
                sc[2] = float(np.asarray(lr_var).reshape(-1)[0])          # configured rate stays, which is what the reference does
            used.add('Variable')
        arrays['opt/scalars'] = sc
        arrays['opt/solver'] = np.int64(0)
        used.update(('beta1_power', 'beta2_power'))
    if iteration is None:
        m = re.search(r'_iter_(\d+)', os.path.basename(prefix))
        iteration = int(m.group(1)) if m else 0
    arrays['meta/iteration'] = np.int64(iteration)
    if output_dir is not None:
        base = re.sub(r'(_iter_\d+)?(\.ckpt)?$', '', os.path.basename(prefix)) or 'converted'
        os.makedirs(output_dir, exist_ok=True)
        out_path = os.path.join(output_dir, '%s_iter_%d.ckpt' % (base, iteration))
    tmp = out_path + '.tmp.npz'
    np.savez(tmp, **arrays)
    os.replace(tmp, out_path)
    if output_dir is not None:
        from .checkpoint import register
        register(out_path)
    matched = sorted(k[4:] for k in arrays if k.startswith('var/'))
    return matched, sorted(k for k in tf_vars if k not in used and not re.search(r'moving_(mean|variance)$', k))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description='TensorFlow checkpoint (V2 bundle) -> lstm_ctc_ocr_amd .npz snapshot, without TensorFlow')
    ap.add_argument('prefix', help='checkpoint path without .index / .data-*, e.g. output/lstm_ctc/LSTM_ctc_iter_40000.ckpt')
    ap.add_argument('out', nargs='?', help='snapshot to write (default: only list the variables)')
    ap.add_argument('--network', default='LSTM_train', help='match against this network\'s variable names')
    ap.add_argument('--verify', action='store_true', help='check every tensor\'s stored crc32c (pure Python, ~1 s per MB)')
    ap.add_argument('--restore-lr', action='store_true', help='take the learning rate from the checkpoint (default: keep the driver\'s, '
                    'as the reference does)')
    ap.add_argument('--output-dir', default=None, help='write <output-dir>/<name>_iter_<n>.ckpt and register it in that directory\'s '
                                                       '`checkpoint` index, so that test_net / train_net --restore load it')
    args = ap.parse_args(argv)
    for name, dt, shape in list_bundle(args.prefix):
        print('%-60s %-10s %s' % (name, getattr(dt, '__name__', dt), shape))
    if args.out or args.output_dir:
        from .models import get_network
        wanted = list(get_network(args.network).param_specs)
        matched, rest = convert(args.prefix, args.out, wanted, verify=args.verify, output_dir=args.output_dir, restore_lr=args.restore_lr)
        print('matched %d of %d variables of %s; unmatched in the network: %s; unused in the checkpoint: %s'
              % (len(matched), len(wanted), args.network, sorted(set(wanted) - set(matched)), rest))


if __name__ == '__main__':
    main()
