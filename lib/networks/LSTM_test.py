from lstm_ctc_ocr_amd.models import LSTM_test  # noqa: F401
