"""Training driver — same entry point, console output, snapshot / validation policy as
/root/reference/lib/lstm/train.py (SolverWrapper :10-162, train_net :165-174), driving the MI355X engine instead of a
TF session.  One process per GPU: under torch.distributed (RCCL) every rank runs this loop on its own data stream and
the engine all-reduces the flat gradient buffer; rank 0 prints and snapshots.
"""
import os

import numpy as np

from . import checkpoint
from . import dist as ocr_dist
from .config import cfg
from .utils.gen import get_batch, stream_seed
from .utils.timer import Timer
from .utils.training import accuracy_calculation


LOSS_SNAPSHOT_BAR = 0.015          # a loss below the best seen so far (starting here) triggers a snapshot + validation (train.py:109,139)


class SolverWrapper(object):
    """The reference's training-loop object (lib/lstm/train.py:10-162) over an Engine instead of a tf.Session.

    Kept from the reference because users and tests see it: constructor arguments, snapshot(sess, iter) and its file naming,
    restoreLabel / mergeLabel, train_model(sess, max_iters, restore), every console line, and the iteration-numbering quirks
    (the loop starts at 1; a resumed run starts at the number in the snapshot's name; a loss-triggered snapshot is always
    written as ..._iter_2.ckpt — SURVEY Q6/Q7).  The loop body is organised as three policies: schedule, snapshot, validation."""

    def __init__(self, sess, network, imgdb, pre_train, output_dir, logdir):
        self.engine, self.net, self.imgdb, self.pre_train = sess, network, imgdb, pre_train
        self.output_dir, self.logdir = output_dir, logdir
        self.loss_log = open(os.path.join(logdir, 'loss.tsv'), 'a') if logdir and os.path.isdir(logdir) else None
        self._val_batch = None
        print('done')

    # ---------------------------------------------------------------------------------------- snapshots
    def snapshot_path(self, iter):
        infix = '_' + cfg.TRAIN.SNAPSHOT_INFIX if cfg.TRAIN.SNAPSHOT_INFIX != '' else ''
        return os.path.join(self.output_dir, '%s_ctc%s_iter_%d.ckpt' % (cfg.TRAIN.SNAPSHOT_PREFIX, infix, iter + 1))

    def snapshot(self, sess, iter):
        os.makedirs(self.output_dir, exist_ok=True)
        filename = self.snapshot_path(iter)
        checkpoint.save(sess, filename)
        print('Wrote snapshot to: {:s}'.format(filename))

    def _resume(self, eng):
        """Load the newest snapshot of output_dir; the iteration to continue from is the number in its file name."""
        ckpt = None
        try:
            ckpt = checkpoint.latest_checkpoint(self.output_dir)
            print('Restoring from {}...'.format(ckpt), end=' ')
            checkpoint.restore(eng, ckpt)
            start = int(os.path.splitext(os.path.basename(ckpt))[0].split('_')[-1])
            eng.iteration = start
            print('done')
            return start
        except Exception:
            raise Exception('Check your pretrained {}'.format(ckpt))

    # ---------------------------------------------------------------------------------------- labels
    def restoreLabel(self, label_vec, label_len):
        """Flat label vector -> one list per sample."""
        out, pos = [], 0
        for n in label_len:
            out.append(label_vec[pos:pos + n])
            pos += n
        return out

    def mergeLabel(self, labels, ignore=0):
        """Per-sample label lists (right-padded with `ignore`) -> flat vector."""
        flat = []
        for seq in labels:
            seq = list(seq)
            while seq and seq[-1] == ignore:
                seq.pop()
            flat.extend(seq)
        return np.array(flat)

    # ---------------------------------------------------------------------------------------- validation
    def _validate(self, eng, val_gen):
        """Sequence accuracy on ONE validation batch, drawn once and reused (train.py:146-162)."""
        if self._val_batch is None:
            images, labels, label_lens, steps = next(val_gen)
            self._val_batch = (np.array(images), np.array(steps), self.restoreLabel(labels, label_lens))
        images, steps, truth = self._val_batch
        return accuracy_calculation(truth, eng.decode(images, steps), ignore_value=0)

    # ---------------------------------------------------------------------------------------- the loop
    def train_model(self, sess, max_iters, restore=False, train_gen=None, val_gen=None):
        eng = sess
        chief = getattr(eng, 'rank', 0) == 0            # data parallel: every rank trains, rank 0 prints / snapshots / validates
        own_gen = train_gen is None
        if train_gen is None:
            # OCR_PIPELINE: 'synth' (default for the reference's single-channel captchas) = the batches are composed on the GPU (utils/synth.py: no
            # PIL workers, the loop runs at 0.96-0.98x of the device-resident rate); 'ring' = PIL worker processes -> shared-memory ring -> pinned
            # asynchronous H2D (utils/pipeline.py: 0.42-0.53x on 16 cores); 'legacy' = the reference's transport, 12 processes -> pickled float batches
            mode = os.environ.get('OCR_PIPELINE') or ('synth' if cfg.NCHANNELS == 1 else 'ring')
            if mode == 'legacy':
                train_gen = get_batch(num_workers=12, batch_size=cfg.TRAIN.BATCH_SIZE, vis=False)
            elif mode == 'synth':
                from .utils.synth import DeviceSynthStream
                train_gen = DeviceSynthStream(eng.device, cfg.TRAIN.BATCH_SIZE)
            elif mode == 'ring':
                from .utils.pipeline import DeviceBatchStream
                train_gen = DeviceBatchStream(eng.device, cfg.TRAIN.BATCH_SIZE)
            else:
                raise ValueError("OCR_PIPELINE=%r: expected 'synth', 'ring' or 'legacy'" % mode)
        val_gen = val_gen or get_batch(num_workers=1, seed=stream_seed(stream=1), batch_size=cfg.VAL.BATCH_SIZE, vis=False)
        world = getattr(eng, 'world', 1)
        self.net.build_loss()
        eng.setup_optimizer(cfg.TRAIN.SOLVER, cfg.TRAIN.LEARNING_RATE)
        first = self._resume(eng) if restore else 1

        timer, best = Timer(), LOSS_SNAPSHOT_BAR
        # The loss of iteration k is read while iteration k + 1 is already running (OCR_LOSS_LAG=1, default): the report is queued
        # behind step k (one kernel + a 32-byte copy + an event) and waited for one iteration later, so the host never drains the
        # queue between two steps.  Every line the reference prints is printed, for the same iteration number, one iteration
        # later; a loss-triggered snapshot / validation therefore sees the weights one update further on.  OCR_LOSS_LAG=0 waits at
        # once (sess.run semantics, train.py:130).
        lag = os.environ.get('OCR_LOSS_LAG', '1') != '0'
        state = {'best': best}

        def on_loss(it, loss, elapsed):
            if world > 1 and it % cfg.TRAIN.DISPLAY == 0:        # what is printed is the GLOBAL-batch loss (one tiny all-reduce, only
                loss = ocr_dist.mean_scalar(loss, eng.device, eng.group)     # on the iterations that print: SURVEY 8e)
            if not chief:
                return False
            if self.loss_log is not None:
                self.loss_log.write('%d\t%.7f\n' % (it, loss))
            if it % cfg.TRAIN.DISPLAY == 0:
                print('iter: %d / %d, total loss: %.7f, lr: %.7f' % (it, max_iters, loss, eng.lr), end=' ')
                print('speed: {:.3f}s / iter'.format(elapsed))
            if loss < state['best']:
                print('loss: ', loss, end=' ')
                self.snapshot(eng, 1)
                state['best'] = loss
                print('accuracy: {:.5f}'.format(self._validate(eng, val_gen)))
                return True
            return False

        pending = None
        for iter in range(first, max_iters):
            timer.tic()
            if iter != 0 and iter % cfg.TRAIN.STEPSIZE == 0:            # step decay of the learning rate
                eng.scale_lr(cfg.TRAIN.GAMMA)
            images, labels, label_lens, steps = next(train_gen)
            if not hasattr(images, 'is_cuda'):                   # host lists of the legacy transport
                images, labels, label_lens, steps = np.array(images), np.array(labels), np.array(label_lens), np.array(steps)
            eng.train_step(images, labels, label_lens, steps, fetch_loss=False)
            handle = eng.report_async()
            fresh = False
            if lag:
                if pending is not None:
                    fresh = on_loss(pending[0], eng.report_wait(pending[1]), pending[2])
                pending = (iter, handle, timer.toc(average=False))
            else:
                fresh = on_loss(iter, eng.report_wait(handle), timer.toc(average=False))
            if not chief:
                continue
            if (iter + 1) % cfg.TRAIN.SNAPSHOT_ITERS == 0 and not fresh:
                self.snapshot(eng, iter)
            if (iter + 1) % cfg.VAL.VAL_STEP == 0 and not fresh:
                print('accuracy: {:.5f}'.format(self._validate(eng, val_gen)))
        if pending is not None:
            on_loss(pending[0], eng.report_wait(pending[1]), pending[2])
        if self.loss_log is not None:
            self.loss_log.flush()
        if own_gen and hasattr(train_gen, 'close'):          # worker processes / feeder thread of the stream this call started
            train_gen.close()


def make_engine(network, use_graphs=True):
    """One engine per process; under torch.distributed.run the process group is RCCL over xGMI."""
    import torch
    from .engine import Engine
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    eng = Engine(network, device='cuda:%d' % local_rank, seed=cfg.RNG_SEED, use_graphs=use_graphs)
    eng.rank = int(os.environ.get('RANK', '0'))
    return eng


def train_net(network, imgdb, pre_train, output_dir, log_dir, max_iters=40000, restore=False):
    eng = make_engine(network)
    sw = SolverWrapper(eng, network, imgdb, pre_train, output_dir, logdir=log_dir)
    print('Solving...')
    sw.train_model(eng, max_iters, restore=restore)
    print('done solving')
