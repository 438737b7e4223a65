"""The C-ABI boundary without a GPU: libocrhip.so loads, exports every symbol include/ocr_hip.h declares, and its
argument validation / status-code convention (warp-ctc's ctcStatus_t numbering) holds for host-only paths."""
import ctypes

from lstm_ctc_ocr_amd import _native as nat


def test_library_exports_every_declared_symbol():
    lib = nat.lib()
    declared = nat.declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
        assert name in nat._SIGS, "ctypes signature missing for %s" % name
    assert lib.ocr_abi_version() == 1


def test_status_codes_and_host_side_validation():
    lib = nat.lib()
    assert nat.status_string(0) == 'no error' and nat.status_string(2) == 'invalid value'
    sz = ctypes.c_size_t(0)
    assert lib.ocr_ctc_workspace_size(10, 63, 64, ctypes.byref(sz)) == 0
    assert sz.value >= 64 * 63 * 22 * 4
    assert lib.ocr_ctc_workspace_size(10, 0, 64, ctypes.byref(sz)) == 2            # invalid value
    assert lib.ocr_ctc_workspace_size(10, 63, 64, None) == 2
    # NULL operands are rejected before any launch is attempted
    assert lib.ocr_ctc_loss(None, None, None, None, None, 64, 64, 63, 10, 0, None, None, None) == 2
    assert lib.ocr_gemm_nt_bf16(None, 0, None, 0, None, 0, 8, 8, 8, None, None, 0, 0, 1, 0, 0, 0, 0, None) == 2
    assert lib.ocr_lstm_seq_supported(64, 256) == 1 and lib.ocr_lstm_seq_supported(64, 128) == 0
    assert lib.ocr_lstm_seq_supported(64 * 9, 256) == 0                            # would not be one workgroup per CU


def test_hot_path_has_no_cpu_fallback():
    import pytest
    import torch
    from lstm_ctc_ocr_amd import ops
    with pytest.raises(nat.NativeError):
        ops.maxpool_fwd(torch.zeros(1, 2, 2, 8, dtype=torch.bfloat16), 2, 2)       # CPU tensor -> loud failure
    if not torch.cuda.is_available():
        from lstm_ctc_ocr_amd.engine import Engine
        from lstm_ctc_ocr_amd.models import get_network
        with pytest.raises(nat.NativeError):
            Engine(get_network('LSTM_train'))
