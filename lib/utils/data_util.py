from lstm_ctc_ocr_amd.utils.data_util import GeneratorEnqueuer  # noqa: F401
