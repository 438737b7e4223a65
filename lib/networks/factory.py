from lstm_ctc_ocr_amd.models import get_network, list_networks  # noqa: F401
