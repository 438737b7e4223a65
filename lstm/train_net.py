#!/usr/bin/env python
"""CLI: train the CRNN on the MI355X engine.  Flags follow the reference's lstm/train_net.py:17-48
(--gpu --iters --cfg --pre_train --rand --network --set --restore); the reference's own script also runs unmodified
when dropped next to this repo's lib/ (see INTEGRATION.md).
Multi-GPU: python -m torch.distributed.run --nproc-per-node N lstm/train_net.py --network=LSTM_train --cfg=lstm/lstm.yml"""
import argparse
import os
import pprint
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import numpy as np  # noqa: E402

from lib.lstm.config import cfg, cfg_from_file, cfg_from_list, get_log_dir, get_output_dir  # noqa: E402
from lib.lstm.train import train_net  # noqa: E402
from lib.networks.factory import get_network  # noqa: E402
from easydict import EasyDict as edict  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train a lstm network')
    p.add_argument('--gpu', dest='gpu_id', help='GPU device id to use [0]', default=0, type=int)
    p.add_argument('--iters', dest='max_iters', help='number of iterations to train', default=1000000, type=int)
    p.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    p.add_argument('--pre_train', dest='pre_train', help='pre trained model', default=None, type=str)
    p.add_argument('--rand', dest='randomize', help='randomize (do not use a fixed seed)', action='store_true')
    p.add_argument('--network', dest='network_name', help='name of the network', default=None, type=str)
    p.add_argument('--set', dest='set_cfgs', help='set config keys', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--restore', dest='restore', help='restore or not', default=0, type=int)
    if argv is None and len(sys.argv) == 1:
        p.print_help()
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs)
    # one process per GPU: the launcher's LOCAL_RANK selects the device; --gpu / cfg.GPU_ID are kept for CLI parity
    print('Using config:')
    pprint.pprint(cfg)
    if not args.randomize:
        np.random.seed(cfg.RNG_SEED)
    tail = args.network_name.split('_')[-1]
    imgdb = edict({'path': './data/train_4_6.tfrecords', 'name': 'lstm_' + tail, 'val_path': './data/val.tfrecords'})
    output_dir = get_output_dir(imgdb, None)
    log_dir = get_log_dir(imgdb)
    print('Output will be saved to `{:s}`'.format(output_dir))
    print('Logs will be saved to `{:s}`'.format(log_dir))
    print('/gpu:{:d}'.format(args.gpu_id))
    network = get_network(args.network_name)
    print('Use network `{:s}` in training'.format(args.network_name))
    train_net(network, imgdb, pre_train=args.pre_train, output_dir=output_dir, log_dir=log_dir,
              max_iters=args.max_iters, restore=bool(int(args.restore)))


if __name__ == '__main__':
    main()
