from lstm_ctc_ocr_amd.edict import EasyDict  # noqa: F401  (stand-in when the PyPI package is absent)
