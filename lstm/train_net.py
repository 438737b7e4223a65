#!/usr/bin/env python
"""Train the CRNN on the MI355X engine:   python lstm/train_net.py --network=LSTM_train --cfg=lstm/lstm.yml [--restore=1]
Same flags as the reference's script of this name (--gpu --iters --cfg --pre_train --rand --network --set --restore); the
reference's own file also runs unmodified against this repository's lib/ (INTEGRATION.md).
Several GPUs: python -m torch.distributed.run --nproc-per-node N lstm/train_net.py --network=LSTM_train --cfg=lstm/lstm.yml
"""
import _cli


def parse_args(argv=None):
    flags = ('--gpu', '--iters', '--cfg', '--pre_train', '--network', '--restore')
    return _cli.parse(_cli.build_parser('Train a lstm network', flags, restore_default=0, with_set=True, with_rand=True), argv)


def main(argv=None):
    args = parse_args(argv)
    network, imgdb, output_dir, log_dir = _cli.open_session(
        args, {'path': './data/train_4_6.tfrecords', 'val_path': './data/val.tfrecords'})
    import numpy as np
    from lib.lstm.config import cfg
    from lib.lstm.train import train_net
    if not args.randomize:
        np.random.seed(cfg.RNG_SEED)
    train_net(network, imgdb, pre_train=args.pre_train, output_dir=output_dir, log_dir=log_dir, max_iters=args.max_iters,
              restore=bool(int(args.restore)))


if __name__ == '__main__':
    main()
