from lstm_ctc_ocr_amd.utils.training import accuracy_calculation  # noqa: F401
