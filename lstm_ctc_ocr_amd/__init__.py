"""MI355X-native CRNN-OCR hot path (capabilities of ilovin/lstm_ctc_ocr): PyTorch-ROCm is used for device memory,
streams and torch.distributed only; every operator is a hand-written gfx950 HIP kernel behind the C ABI declared in
include/ocr_hip.h (built into lstm_ctc_ocr_amd/libocrhip.so)."""
__version__ = "0.1.0"
