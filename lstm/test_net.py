#!/usr/bin/env python
"""Evaluate a trained CRNN on a directory of <idx>_<label>.png files (default ./data/val/):
    python lstm/test_net.py --network=LSTM_test --cfg=lstm/lstm.yml
Flags of the reference's script of this name (--gpu --network --cfg --restore) plus --dir.
"""
import _cli


def parse_args(argv=None):
    return _cli.parse(_cli.build_parser('Test a lstm network', ('--gpu', '--network', '--cfg', '--restore', '--dir'), restore_default=1), argv)


def main(argv=None):
    args = parse_args(argv)
    network, imgdb, output_dir, log_dir = _cli.open_session(args, {'path': './data/train.tfrecords', 'val_path': './data/val.tfrecords'})
    from lib.lstm.test import test_net
    test_net(network, imgdb, testDir=args.test_dir, output_dir=output_dir, log_dir=log_dir, restore=bool(int(args.restore)))


if __name__ == '__main__':
    main()
