#!/usr/bin/env bash
# Train the captcha CRNN on one MI355X (extra flags are passed through, e.g. ./train.sh --iters 40000 --restore=1).
# Several GPUs of one node:  python -m torch.distributed.run --nproc-per-node N lstm/train_net.py --network=LSTM_train --cfg=lstm/lstm.yml
set -e
cd "$(dirname "$0")"
exec python ./lstm/train_net.py --network=LSTM_train --cfg=./lstm/lstm.yml --restore=0 "$@"
