python ./lstm/test_net.py --network=LSTM_test --cfg=./lstm/lstm.yml --restore=1
