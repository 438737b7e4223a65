from lstm_ctc_ocr_amd.network import Network, layer, DEFAULT_PADDING  # noqa: F401
