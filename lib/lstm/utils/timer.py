from lstm_ctc_ocr_amd.utils.timer import Timer  # noqa: F401
