python ./lstm/train_net.py --network=LSTM_train --cfg=./lstm/lstm.yml --restore=0
