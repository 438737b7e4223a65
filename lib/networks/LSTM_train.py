from lstm_ctc_ocr_amd.models import LSTM_train  # noqa: F401
