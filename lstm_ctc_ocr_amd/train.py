"""Training driver — same entry point, console output, snapshot / validation policy as
/root/reference/lib/lstm/train.py (SolverWrapper :10-162, train_net :165-174), driving the MI355X engine instead of a
TF session.  One process per GPU: under torch.distributed (RCCL) every rank runs this loop on its own data stream and
the engine all-reduces the flat gradient buffer; rank 0 prints and snapshots.
"""
import os

import numpy as np

from . import checkpoint
from .config import cfg
from .utils.gen import get_batch
from .utils.timer import Timer
from .utils.training import accuracy_calculation


class SolverWrapper(object):
    def __init__(self, sess, network, imgdb, pre_train, output_dir, logdir):
        """`sess` is the Engine (the role tf.Session plays in the reference)."""
        self.net = network
        self.imgdb = imgdb
        self.pre_train = pre_train
        self.output_dir = output_dir
        self.logdir = logdir
        self.engine = sess
        self.loss_log = open(os.path.join(logdir, 'loss.tsv'), 'a') if logdir and os.path.isdir(logdir) else None
        print('done')

    def snapshot(self, sess, iter):
        if not os.path.exists(self.output_dir):
            os.makedirs(self.output_dir)
        infix = ('_' + cfg.TRAIN.SNAPSHOT_INFIX if cfg.TRAIN.SNAPSHOT_INFIX != '' else '')
        filename = (cfg.TRAIN.SNAPSHOT_PREFIX + '_ctc' + infix + '_iter_{:d}'.format(iter + 1) + '.ckpt')
        filename = os.path.join(self.output_dir, filename)
        checkpoint.save(sess, filename)
        print('Wrote snapshot to: {:s}'.format(filename))

    def restoreLabel(self, label_vec, label_len):
        labels = []
        for l_len in label_len:
            labels.append(label_vec[:l_len])
            label_vec = label_vec[l_len:]
        return labels

    def mergeLabel(self, labels, ignore=0):
        label_lst = []
        for l in labels:
            while l[-1] == ignore:
                l = l[:-1]
            label_lst.extend(l)
        return np.array(label_lst)

    def train_model(self, sess, max_iters, restore=False, train_gen=None, val_gen=None):
        eng = sess
        rank = getattr(eng, 'rank', 0)
        if train_gen is None:
            train_gen = get_batch(num_workers=12, batch_size=cfg.TRAIN.BATCH_SIZE, vis=False)
        if val_gen is None:
            val_gen = get_batch(num_workers=1, batch_size=cfg.VAL.BATCH_SIZE, vis=False)

        loss_node, dense_decoded = self.net.build_loss()
        eng.setup_optimizer(cfg.TRAIN.SOLVER, cfg.TRAIN.LEARNING_RATE)
        restore_iter = 1

        if restore:                                     # resuming a trainer (train.py:96-106)
            ckpt = None
            try:
                ckpt = checkpoint.latest_checkpoint(self.output_dir)
                print('Restoring from {}...'.format(ckpt), end=' ')
                checkpoint.restore(eng, ckpt)
                stem = os.path.splitext(os.path.basename(ckpt))[0]
                restore_iter = int(stem.split('_')[-1])
                eng.iteration = restore_iter
                print('done')
            except Exception:
                raise Exception('Check your pretrained {}'.format(ckpt))

        timer = Timer()
        loss_min = 0.015
        first_val = True
        for iter in range(restore_iter, max_iters):
            timer.tic()
            if iter != 0 and iter % cfg.TRAIN.STEPSIZE == 0:            # step LR decay (train.py:114-115)
                eng.scale_lr(cfg.TRAIN.GAMMA)

            img_Batch, label_Batch, label_len_Batch, time_step_Batch = next(train_gen)
            ctc_loss = eng.train_step(np.array(img_Batch), np.array(label_Batch), np.array(label_len_Batch),
                                      np.array(time_step_Batch))
            if self.loss_log is not None and rank == 0:
                self.loss_log.write('%d\t%.7f\n' % (iter, ctc_loss))
            _diff_time = timer.toc(average=False)

            if rank != 0:
                continue
            if iter % cfg.TRAIN.DISPLAY == 0:
                print('iter: %d / %d, total loss: %.7f, lr: %.7f' % (iter, max_iters, ctc_loss, eng.lr), end=' ')
                print('speed: {:.3f}s / iter'.format(_diff_time))
            if (iter + 1) % cfg.TRAIN.SNAPSHOT_ITERS == 0 or ctc_loss < loss_min:
                if ctc_loss < loss_min:
                    print('loss: ', ctc_loss, end=' ')
                    self.snapshot(eng, 1)
                    loss_min = ctc_loss
                else:
                    self.snapshot(eng, iter)
            if (iter + 1) % cfg.VAL.VAL_STEP == 0 or loss_min == ctc_loss:
                if first_val:
                    val_img_Batch, val_label_Batch, val_label_len_Batch, val_time_step_Batch = next(val_gen)
                    org = self.restoreLabel(val_label_Batch, val_label_len_Batch)
                    first_val = False
                res = eng.decode(np.array(val_img_Batch), np.array(val_time_step_Batch))
                acc = accuracy_calculation(org, res, ignore_value=0)
                print('accuracy: {:.5f}'.format(acc))
        if self.loss_log is not None:
            self.loss_log.flush()


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
