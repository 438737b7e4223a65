"""Shared plumbing of the two command-line drivers (lstm/train_net.py, lstm/test_net.py).

The flag names, destinations and defaults are those of the reference's scripts (lstm/train_net.py:17-48, lstm/test_net.py:19-38)
so its train.sh / test.sh command lines work unchanged; everything else — one flag table, one session-setup routine for both
drivers — is this repository's own.
"""
import argparse
import os
import pprint
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

#        flag          dest           type   default   help
FLAGS = {
    '--gpu':       ('gpu_id',       int,   0,        'GPU device id to use [0] (kept for compatibility: the launcher\'s LOCAL_RANK selects the device)'),
    '--iters':     ('max_iters',    int,   1000000,  'number of iterations to train'),
    '--cfg':       ('cfg_file',     str,   None,     'optional config file'),
    '--pre_train': ('pre_train',    str,   None,     'pre trained model'),
    '--network':   ('network_name', str,   None,     'name of the network'),
    '--restore':   ('restore',      int,   None,     'restore or not'),
    '--dir':       ('test_dir',     str,   './data/val/', 'directory of <idx>_<label>.png files'),
}


def build_parser(description, flags, restore_default, with_set=False, with_rand=False):
    parser = argparse.ArgumentParser(description=description)
    for flag in flags:
        dest, kind, default, text = FLAGS[flag]
        parser.add_argument(flag, dest=dest, type=kind, help=text, default=restore_default if flag == '--restore' else default)
    if with_rand:
        parser.add_argument('--rand', dest='randomize', action='store_true', help='randomize (do not use a fixed seed)')
    if with_set:
        parser.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER, help='set config keys')
    return parser


def parse(parser, argv):
    if argv is None and len(sys.argv) == 1:
        parser.print_help()
    return parser.parse_args(argv)


def open_session(args, records):
    """Overlay the configuration, derive the image-database record and the output / log directories, echo them the way the
    reference's drivers do, and build the network.  Returns (network, imgdb, output_dir, log_dir)."""
    from easydict import EasyDict
    from lib.lstm.config import cfg, cfg_from_file, cfg_from_list, get_log_dir, get_output_dir
    from lib.networks.factory import get_network

    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if getattr(args, 'set_cfgs', None) is not None:
        cfg_from_list(args.set_cfgs)
    print('Using config:')
    pprint.pprint(cfg)
    imgdb = EasyDict(dict(records, name='lstm_' + args.network_name.split('_')[-1]))
    output_dir, log_dir = get_output_dir(imgdb, None), get_log_dir(imgdb)
    print('Output will be saved to `{:s}`'.format(output_dir))
    print('Logs will be saved to `{:s}`'.format(log_dir))
    print('/gpu:{:d}'.format(args.gpu_id))
    network = get_network(args.network_name)
    print('Use network `{:s}` in training'.format(args.network_name))
    return network, imgdb, output_dir, log_dir
