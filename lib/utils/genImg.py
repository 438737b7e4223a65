from lstm_ctc_ocr_amd.utils.genImg import run  # noqa: F401
