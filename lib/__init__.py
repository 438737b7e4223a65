"""Import-path shim: the reference's `lib.*` module tree re-exported from lstm_ctc_ocr_amd, so that
lstm/train_net.py and lstm/test_net.py of ilovin/lstm_ctc_ocr run unmodified against the MI355X engine."""
