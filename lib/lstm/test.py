from lstm_ctc_ocr_amd.test import SolverWrapper, test_net  # noqa: F401
