#!/usr/bin/env python
"""CLI: evaluate a trained CRNN on ./data/val/ — flags of the reference's lstm/test_net.py:19-38
(--gpu --network --cfg --restore)."""
import argparse
import os
import pprint
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from lib.lstm.config import cfg, cfg_from_file, get_log_dir, get_output_dir  # noqa: E402
from lib.lstm.test import test_net  # noqa: E402
from lib.networks.factory import get_network  # noqa: E402
from easydict import EasyDict as edict  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Test a lstm network')
    p.add_argument('--gpu', dest='gpu_id', help='GPU device id to use [0]', default=0, type=int)
    p.add_argument('--network', dest='network_name', help='name of the network', default=None, type=str)
    p.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    p.add_argument('--restore', dest='restore', help='restore or not', default=1, type=int)
    p.add_argument('--dir', dest='test_dir', help='directory of <idx>_<label>.png files', default='./data/val/', type=str)
    if argv is None and len(sys.argv) == 1:
        p.print_help()
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    print('Using config:')
    pprint.pprint(cfg)
    tail = args.network_name.split('_')[-1]
    imgdb = edict({'path': './data/train.tfrecords', 'name': 'lstm_' + tail, 'val_path': './data/val.tfrecords'})
    output_dir = get_output_dir(imgdb, None)
    log_dir = get_log_dir(imgdb)
    print('Output will be saved to `{:s}`'.format(output_dir))
    print('Logs will be saved to `{:s}`'.format(log_dir))
    print('/gpu:{:d}'.format(args.gpu_id))
    network = get_network(args.network_name)
    print('Use network `{:s}` in training'.format(args.network_name))
    test_net(network, imgdb, testDir=args.test_dir, output_dir=output_dir, log_dir=log_dir, restore=bool(int(args.restore)))


if __name__ == '__main__':
    main()
