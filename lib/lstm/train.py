from lstm_ctc_ocr_amd.train import SolverWrapper, train_net  # noqa: F401
