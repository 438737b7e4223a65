#!/usr/bin/env bash
# Evaluate the latest snapshot on ./data/val/<idx>_<label>.png (write some with: python -m lstm_ctc_ocr_amd.utils.genImg 500 ./data/val/).
set -e
cd "$(dirname "$0")"
exec python ./lstm/test_net.py --network=LSTM_test --cfg=./lstm/lstm.yml --restore=1 "$@"
