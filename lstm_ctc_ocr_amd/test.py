"""Inference driver — /root/reference/lib/lstm/test.py (SolverWrapper.test_model :27-88, test_net :91-100): walks a
directory of `<idx>_<label>.png` captchas, pads each to a multiple of POOL_SCALE, decodes with batch 1 and prints the
per-file result and the exact-match accuracy in the reference's format.

Deviations: images are read with PIL (cv2 is not available); `time_step_len` is
W // POOL_SCALE + OFFSET_TIME_STEP — the reference feeds W // POOL_SCALE, one more than the graph produces
(SURVEY Q3), which TF's sequence ops reject; decode is the device implementation of the reference's
ctc_beam_search_decoder (beam 100, blank C-1, merge_repeated) followed by the same zero stripping.
"""
import math
import os

import numpy as np
from PIL import Image

from . import checkpoint
from .config import cfg, get_encode_decode_dict
from .utils.timer import Timer


class SolverWrapper(object):
    def __init__(self, sess, network, imgdb, output_dir, logdir, pretrained_model=None):
        self.net = network
        self.imgdb = imgdb
        self.output_dir = output_dir
        self.pretrained_model = pretrained_model
        self.engine = sess
        print('done')

    def test_model(self, sess, testDir=None, restore=True):
        eng = sess
        if restore:
            ckpt = None
            try:
                ckpt = checkpoint.latest_checkpoint(self.output_dir)
                print('Restoring from {}...'.format(ckpt), end=' ')
                checkpoint.restore(eng, ckpt, with_optimizer=False)
                print('done')
            except Exception:
                raise Exception('Check your pretrained {}'.format(ckpt))

        timer = Timer()
        encode_maps, decode_maps = get_encode_decode_dict()
        total = correct = 0
        for file in sorted(os.listdir(testDir)):
            timer.tic()
            total += 1
            img = Image.open(os.path.join(testDir, file))
            img = np.array(img.convert('L') if cfg.NCHANNELS == 1 else img.convert('RGB'))
            print(file, end=' ')
            if img.shape[0] != cfg.NUM_FEATURES:        # the reference reshapes blindly (test.py:69) and raises for raw 160x60
                from .utils.gen import _resize      # captchas; apply the training generator's resize (gen.py:47-52) instead
                img = _resize(img, int(cfg.NUM_FEATURES / img.shape[0] * img.shape[1]), cfg.NUM_FEATURES)
            w = img.shape[1]
            width = int(math.ceil(w / cfg.POOL_SCALE) * cfg.POOL_SCALE)
            pad = [(0, 0), (0, width - w)] + [(0, 0)] * (img.ndim - 2)
            img = np.pad(img, pad, mode='constant', constant_values=0).astype(np.float32) / 255.
            img = np.reshape(img.swapaxes(0, 1), [1, width, cfg.NUM_FEATURES])
            res = eng.decode(img, np.array([width // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP], np.int32))[0]
            org = file.split('.')[0].split('_')[1]
            res = ''.join(decode_maps[i] for i in res if i != 0)
            if org == res:
                correct += 1
            _diff_time = timer.toc(average=False)
            print('cost time: {:.3f},\n    res: {}'.format(_diff_time, res))
        print('total acc:{}/{}={:.4f}'.format(correct, total, correct / max(total, 1)))
        return correct, total


def test_net(network, imgdb, testDir, output_dir, log_dir, pretrained_model=None, restore=True):
    from .train import make_engine
    eng = make_engine(network)
    sw = SolverWrapper(eng, network, imgdb, output_dir, logdir=log_dir, pretrained_model=pretrained_model)
    print('Solving...')
    sw.test_model(eng, testDir=testDir, restore=restore)
    print('done solving')
