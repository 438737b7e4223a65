"""CPU: the TensorFlow-free reader of tensor-bundle checkpoints (lstm_ctc_ocr_amd/tf_bundle.py) on bundles written by its own writer —
multi-block index with prefix-compressed keys, scalars, several dtypes — on hand-encoded snappy streams, and the conversion to this
package's .npz snapshots (variable-name normalisation, Adam slots, step counters).  No TensorFlow-written file exists offline: the
format itself stays unpinned (see the module docstring)."""
import struct

import os

import numpy as np
import pytest

from lstm_ctc_ocr_amd import tf_bundle as tb


def _arrays(seed=0):
    rng = np.random.RandomState(seed)
    return {
        'conv1/weights': rng.randn(3, 3, 1, 64).astype(np.float32),
        'conv1/biases': rng.randn(64).astype(np.float32),
        'conv4_1/conv4_1/beta': rng.randn(32).astype(np.float32),
        'conv4_1/conv4_1/gamma': rng.randn(32).astype(np.float32),
        'conv4_1/conv4_1/moving_mean': np.zeros(32, np.float32),
        'logits/bidirectional_rnn/fw/lstm_cell/weights': rng.randn(12, 16).astype(np.float32),
        'logits/bidirectional_rnn/fw/lstm_cell/biases': rng.randn(16).astype(np.float32),
        'logits/bidirectional_rnn/fw/lstm_cell/weights/Adam': rng.randn(12, 16).astype(np.float32),
        'logits/bidirectional_rnn/fw/lstm_cell/weights/Adam_1': rng.rand(12, 16).astype(np.float32),
        'logits/weights': rng.randn(8, 4).astype(np.float32),
        'beta1_power': np.float32(0.9 ** 1234),
        'beta2_power': np.float32(0.999 ** 1234),
        'global_step': np.int64(1234),
        'some/int32': np.arange(7, dtype=np.int32),
    }


@pytest.mark.parametrize("per_block", [1, 3, 100])
def test_round_trip_through_the_table_format(tmp_path, per_block):
    arrays = _arrays()
    prefix = str(tmp_path / 'LSTM_ctc_iter_2000.ckpt')
    tb.write_bundle(prefix, arrays, per_block=per_block)
    got = tb.read_bundle(prefix)
    assert sorted(got) == sorted(arrays)
    for k, v in arrays.items():
        assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape and np.array_equal(got[k], v), k
    listed = {n: (dt, sh) for n, dt, sh in tb.list_bundle(prefix)}
    assert listed['conv1/weights'] == (np.float32, (3, 3, 1, 64)) and listed['beta1_power'][1] == ()
    only = tb.read_bundle(prefix, names={'conv1/biases'})
    assert list(only) == ['conv1/biases']


def test_crc32c_and_corruption_detection(tmp_path):
    assert tb.crc32c(b'123456789') == 0xE3069283                      # the standard CRC-32C check value
    assert tb.crc32c(b'') == 0 and tb.crc32c(b'\x00' * 32) == 0x8A9136AA
    arrays = {'a/w': np.arange(40, dtype=np.float32), 'b/w': np.arange(7, dtype=np.int32)}
    prefix = str(tmp_path / 'k.ckpt')
    tb.write_bundle(prefix, arrays)
    assert np.array_equal(tb.read_bundle(prefix, verify=True)['a/w'], arrays['a/w'])
    data_file = prefix + '.data-00000-of-00001'
    raw = bytearray(open(data_file, 'rb').read()); raw[5] ^= 0x40
    open(data_file, 'wb').write(bytes(raw))
    tb.read_bundle(prefix)                                            # unverified read does not notice
    with pytest.raises(ValueError, match='tensor checksum'):
        tb.read_bundle(prefix, verify=True)
    idx = bytearray(open(prefix + '.index', 'rb').read()); idx[3] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='checksum mismatch'):
        tb.read_table(prefix + '.index')


def test_rejects_files_that_are_not_tables(tmp_path):
    p = tmp_path / 'x.index'
    p.write_bytes(b'\x00' * 100)
    with pytest.raises(ValueError):
        tb.read_table(str(p))


def test_snappy_block_decoder():
    raw = b'abcdefgh' * 5 + b'XYZ'
    # literal 'abcdefgh' (tag (8-1)<<2), copy-2 of 32 bytes from offset 8 (tag ((32-1)<<2)|2 + uint16 offset), literal 'XYZ'
    stream = tb._put_varint(len(raw)) + bytes([(8 - 1) << 2]) + b'abcdefgh' + bytes([((32 - 1) << 2) | 2]) + struct.pack('<H', 8) \
        + bytes([(3 - 1) << 2]) + b'XYZ'
    assert tb._snappy_decompress(stream) == raw
    # copy-1: length 4..11, 11-bit offset
    raw2 = b'0123' + b'0123'
    stream2 = tb._put_varint(8) + bytes([(4 - 1) << 2]) + b'0123' + bytes([((4 - 4) << 2) | 1 | (0 << 5)]) + bytes([4])
    assert tb._snappy_decompress(stream2) == raw2
    with pytest.raises(ValueError):
        tb._snappy_decompress(tb._put_varint(9) + bytes([(4 - 1) << 2]) + b'0123')          # length mismatch


def test_compressed_blocks_are_read(tmp_path, monkeypatch):
    """An index whose data block is stored snappy-compressed (type byte 1): literal-only stream of the same block bytes.  The
    hand-built table carries zero trailers: only the explicit test switch lets the reader accept them — on a real file a zeroed
    trailer is corruption."""
    monkeypatch.setattr(tb, 'ALLOW_UNCHECKED_BLOCKS', True)
    arrays = {'a/weights': np.arange(12, dtype=np.float32).reshape(3, 4)}
    prefix = str(tmp_path / 'c.ckpt')
    tb.write_bundle(prefix, arrays, per_block=100)
    data = bytearray(open(prefix + '.index', 'rb').read())
    footer = bytes(data[-48:])
    moff, p = tb._varint(footer, 0); msize, p = tb._varint(footer, p)
    blk = bytes(data[:moff - 5])                                   # the single data block (without its trailer)
    assert len(blk) < 60 * 256
    lit = bytearray(tb._put_varint(len(blk)))
    for i in range(0, len(blk), 60):                               # literals of <= 60 bytes: tag = (len - 1) << 2
        piece = blk[i:i + 60]
        lit += bytes([(len(piece) - 1) << 2]) + piece
    comp = bytes(lit)
    meta = tb._build_block([])
    out = bytearray(comp + b'\x01' + b'\x00' * 4)
    meta_handle = tb._put_varint(len(out)) + tb._put_varint(len(meta))
    out += meta + b'\x00' * 5
    idx = tb._build_block([(b'a/weights', tb._put_varint(0) + tb._put_varint(len(comp)))], restart_interval=1)
    idx_handle = tb._put_varint(len(out)) + tb._put_varint(len(idx))
    out += idx + b'\x00' * 5
    foot = meta_handle + idx_handle
    foot += b'\x00' * (40 - len(foot)) + struct.pack('<Q', tb.TABLE_MAGIC)
    open(prefix + '.index', 'wb').write(bytes(out) + foot)
    got = tb.read_bundle(prefix)
    assert np.array_equal(got['a/weights'], arrays['a/weights'])
    monkeypatch.setattr(tb, 'ALLOW_UNCHECKED_BLOCKS', False)
    with pytest.raises(ValueError):
        tb.read_bundle(prefix)


def test_conversion_to_a_snapshot(tmp_path):
    arrays = _arrays(1)
    prefix = str(tmp_path / 'LSTM_ctc_iter_2000.ckpt')
    tb.write_bundle(prefix, arrays)
    wanted = ['conv1/weights', 'conv1/biases', 'conv4_1/conv4_1/beta', 'conv4_1/conv4_1/gamma', 'logits/fw/weights', 'logits/fw/biases',
              'logits/weights', 'logits/biases']
    out = str(tmp_path / 'snap.npz')
    matched, rest = tb.convert(prefix, out, wanted)
    assert matched == sorted(set(wanted) - {'logits/biases'})                        # not in the checkpoint: reported by its absence
    assert 'global_step' not in rest and 'some/int32' in rest and not any('moving_' in r for r in rest)
    snap = np.load(out)
    assert np.array_equal(snap['var/logits/fw/weights'], arrays['logits/bidirectional_rnn/fw/lstm_cell/weights'])
    assert np.array_equal(snap['slot1/logits/fw/weights'], arrays['logits/bidirectional_rnn/fw/lstm_cell/weights/Adam'])
    assert np.array_equal(snap['slot2/logits/fw/weights'], arrays['logits/bidirectional_rnn/fw/lstm_cell/weights/Adam_1'])
    sc = snap['opt/scalars']
    assert sc[6] == 1234 and sc[4] == 0.9 ** 1234 and sc[5] == 0.999 ** 1234 and int(snap['meta/iteration']) == 2000
    # without a step variable the count comes from what Adam itself multiplied up — beta2^t (beta1^t = 0.9^1234 is 0 in float32) — whatever
    # the file is called (the `_iter_<n>` of the name used to take precedence over the exact value) ...
    del arrays['global_step']
    tb.write_bundle(prefix, arrays)
    tb.convert(prefix, out, wanted)
    assert np.load(out)['opt/scalars'][6] == 1234
    plain = str(tmp_path / 'weights_only_name.ckpt')
    tb.write_bundle(plain, arrays)
    tb.convert(plain, out, wanted)
    assert np.load(out)['opt/scalars'][6] == 1234
    # ... exactly, at the lengths the reference trains for: TF multiplies a FLOAT32 beta2_power by float32(0.999) every step, so the
    # logarithm has to be taken to that base (to the base 0.999 the count is one short from step 38 539 on: ADVICE r4)
    p2, at = np.float32(1.0), {}
    for t in range(1, 86001):
        p2 = np.float32(p2 * np.float32(0.999))
        if t in (38539, 40000, 86000):
            at[t] = p2
    for t, v in at.items():
        late = dict(arrays, beta1_power=np.float32(0.0), beta2_power=v)
        tb.write_bundle(plain, late)
        tb.convert(plain, out, wanted)
        assert np.load(out)['opt/scalars'][6] == t, t
    # ... and from the Saver's file name only when both powers have underflowed (`_iter_2000` = 1999 completed steps: the reference's loop
    # starts at 1 and names a snapshot iter + 1, train.py:27-36,111)
    flushed = dict(arrays, beta1_power=np.float32(0.0), beta2_power=np.float32(0.0))
    tb.write_bundle(prefix, flushed)
    tb.convert(prefix, out, wanted)
    assert np.load(out)['opt/scalars'][6] == 1999
    assert tb.normalise_name('logits/bidirectional_rnn/bw/lstm_cell/biases') == 'logits/bw/biases'
    # a Saver created AFTER the optimiser (not the reference's own: train.py:18 builds it before lr / global_step / the Adam slots exist)
    # also holds the unnamed learning-rate and step variables (train.py:73,78) -> `Variable` (float) / `Variable_1` (int): the step comes
    # from the integer, not from float32 beta powers; the learning rate stays the DRIVER's (the reference never restores it) unless asked for
    arrays['Variable'], arrays['Variable_1'] = np.float32(1e-5), np.int32(54321)
    arrays['beta2_power'] = np.float32(0.0)                                      # denormal / flushed after ~87k steps: must not be used
    tb.write_bundle(prefix, arrays)
    matched, rest = tb.convert(prefix, out, wanted)
    sc = np.load(out)['opt/scalars']
    assert sc[6] == 54321 and sc[2] == 0.0 and 'Variable' not in rest and 'Variable_1' not in rest
    tb.convert(prefix, out, wanted, restore_lr=True)
    assert abs(np.load(out)['opt/scalars'][2] - 1e-5) < 1e-12
    # --output-dir: written under the Saver's naming and registered in the directory's `checkpoint` index -> latest_checkpoint finds it
    from lstm_ctc_ocr_amd import checkpoint
    outdir = str(tmp_path / 'out')
    tb.convert(prefix, None, wanted, output_dir=outdir)
    latest = checkpoint.latest_checkpoint(outdir)
    assert latest is not None and os.path.basename(latest) == 'LSTM_ctc_iter_2000.ckpt' and int(np.load(latest)['meta/iteration']) == 2000


@pytest.mark.gpu
def test_converted_checkpoint_restores_into_an_engine(dev, tmp_path):
    """Engine -> TF-named bundle (helper scopes of bidirectional_dynamic_rnn / LSTMCell put back) -> convert -> restore: parameters and
    Adam slots bit-equal, step count kept, learning rate taken from the driver's configuration (a TF checkpoint stores none)."""
    import torch
    from lstm_ctc_ocr_amd import checkpoint
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=11)
    eng.setup_optimizer('Adam', 3e-4)
    tf_name = lambda n: n.replace('logits/fw/', 'logits/bidirectional_rnn/fw/lstm_cell/').replace('logits/bw/', 'logits/bidirectional_rnn/bw/lstm_cell/')
    arrays = {}
    rng = np.random.RandomState(0)
    slots = {}
    for name, a in eng.state_arrays().items():
        arrays[tf_name(name)] = a
        slots[name] = (rng.randn(*a.shape).astype(np.float32), rng.rand(*a.shape).astype(np.float32))
        arrays[tf_name(name) + '/Adam'], arrays[tf_name(name) + '/Adam_1'] = slots[name]
    arrays['beta1_power'], arrays['beta2_power'], arrays['global_step'] = np.float32(0.9 ** 77), np.float32(0.999 ** 77), np.int64(77)
    prefix = str(tmp_path / 'LSTM_ctc_iter_78.ckpt')
    tb.write_bundle(prefix, arrays, per_block=4, tensor_crc=False)      # (the pure-Python crc32c would take minutes on 86 MB)
    out = str(tmp_path / 'LSTM_ctc_iter_78.npz')
    matched, rest = tb.convert(prefix, out, list(eng.specs))
    assert matched == sorted(eng.specs) and rest == []
    eng2 = Engine(get_network('LSTM_train'), device='cuda:0', seed=12)
    eng2.setup_optimizer('Adam', 3e-4)
    checkpoint.restore(eng2, out)
    assert torch.equal(eng2.params, eng.params) and eng2.iteration == 78 and eng2.lr == 3e-4
    sc = eng2.scalars.cpu().numpy()
    assert sc[6] == 77 and sc[2] == 3e-4
    for name in ('conv2/weights', 'logits/bw/weights'):
        o, n = eng2.offsets[name], slots[name][0].size
        assert np.array_equal(eng2.state1[o:o + n].cpu().numpy(), slots[name][0].reshape(-1))
        assert np.array_equal(eng2.state2[o:o + n].cpu().numpy(), slots[name][1].reshape(-1))
