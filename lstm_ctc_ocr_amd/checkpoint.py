"""Checkpoint I/O with the reference's Saver protocol (lib/lstm/train.py:23-37, 96-106) on a portable container.

The reference writes TF-1.0 `.ckpt` bundles through tf.train.Saver(max_to_keep=100); TensorFlow is not available
here, so a snapshot is a single `.npz` holding every variable under its TF name and layout
(conv*/weights [kh,kw,ci,co], conv4_x/conv4_x/{beta,gamma}, logits/{fw,bw}/{weights[768,1024],biases},
logits/{weights,biases}) plus the optimiser slots and step counters, and a text file `checkpoint` naming the
latest one — the same naming (`lstm_ctc[_infix]_iter_<n>.ckpt`), resume rule (iteration parsed from the file
name) and retention as the reference.  A dict of arrays converted from a real TF checkpoint loads through
Engine.load_arrays unchanged; `python -m lstm_ctc_ocr_amd.tf_bundle <prefix> <out.npz>` converts a TensorFlow V2 checkpoint
(weights, Adam slots, step count) into such a snapshot without TensorFlow (format unpinned offline: see that module).
"""
import os

import numpy as np

MAX_TO_KEEP = 100


def _index_path(output_dir):
    return os.path.join(output_dir, 'checkpoint')


def save(engine, path):
    arrays = {'var/' + k: v for k, v in engine.state_arrays().items()}
    if engine.opt_ready:            # optimiser slots per variable, TF Saver style ("<var>/Adam", "<var>/Adam_1"): independent of
        s1 = engine.state1.cpu().numpy()                # the engine's flat parameter layout
        s2 = engine.state2.cpu().numpy() if engine.state2 is not None else None
        for name, spec in engine.specs.items():
            o, n = engine.offsets[name], int(np.prod(spec.shape))
            arrays['slot1/' + name] = s1[o:o + n].reshape(spec.shape)
            if s2 is not None:
                arrays['slot2/' + name] = s2[o:o + n].reshape(spec.shape)
        arrays['opt/scalars'] = engine.scalars.cpu().numpy()
        arrays['opt/solver'] = np.int64(engine.solver)
    arrays['meta/iteration'] = np.int64(engine.iteration)
    tmp = path + '.tmp.npz'
    np.savez(tmp, **arrays)
    os.replace(tmp, path)
    return register(path)


def register(path):
    """Append a snapshot to its directory's `checkpoint` index (newest last) and apply the Saver's retention."""
    out_dir = os.path.dirname(path)
    idx = _index_path(out_dir)
    kept = []
    if os.path.exists(idx):
        kept = [l.strip() for l in open(idx) if l.strip()]
    base = os.path.basename(path)
    kept = [k for k in kept if k != base] + [base]
    while len(kept) > MAX_TO_KEEP:
        old = kept.pop(0)
        try:
            os.remove(os.path.join(out_dir, old))
        except OSError:
            pass
    with open(idx, 'w') as f:
        f.write("\n".join(kept) + "\n")
    return path


def latest_checkpoint(output_dir):
    idx = _index_path(output_dir)
    if not os.path.exists(idx):
        return None
    kept = [l.strip() for l in open(idx) if l.strip()]
    return os.path.join(output_dir, kept[-1]) if kept else None


def restore(engine, path, with_optimizer=True):
    """Load a snapshot.  with_optimizer=False (inference, test.py) restores the variables only: no optimiser slots are
    allocated for a network that will never take a step."""
    import torch
    data = np.load(path)
    engine.load_arrays({k[4:]: data[k] for k in data.files if k.startswith('var/')})
    if with_optimizer and 'opt/scalars' in data.files:
        if not engine.opt_ready:
            engine.setup_optimizer()
        s1 = np.zeros(engine.n_total, np.float32)
        s2 = np.zeros(engine.n_total, np.float32)
        for name, spec in engine.specs.items():
            o, n = engine.offsets[name], int(np.prod(spec.shape))
            if 'slot1/' + name in data.files:
                s1[o:o + n] = data['slot1/' + name].reshape(-1)
            if 'slot2/' + name in data.files:
                s2[o:o + n] = data['slot2/' + name].reshape(-1)
        engine.state1.copy_(torch.from_numpy(s1))
        if engine.state2 is not None:
            engine.state2.copy_(torch.from_numpy(s2))
        sc = torch.from_numpy(data['opt/scalars'])          # the first 8 doubles are state, the rest per-step scratch
        n = min(sc.numel(), engine.scalars.numel(), 8)      # state only: bins and arrival tickets are per-step scratch that a finished step leaves zero
        engine.scalars[:n].copy_(sc[:n])
        if float(data['opt/scalars'][2]) > 0:
            engine.lr = float(data['opt/scalars'][2])       # host mirror of the device learning rate (console line, scale_lr)
        else:                                               # a snapshot converted from a TensorFlow checkpoint (tf_bundle.convert) carries no
            from . import ops                               # learning rate: keep the one the driver configured
            ops.optim_set_lr(engine.scalars, engine.lr)
    engine.iteration = int(data['meta/iteration']) if 'meta/iteration' in data.files else 0
    return engine
