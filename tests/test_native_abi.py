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


def test_library_is_built_from_the_sources_in_the_tree():
    """ocr_build_id() (csrc/Makefile: sha256 over the product sources) against the same hash recomputed from the tree: a stale .so —
    e.g. one left over from before an edit, which would travel to the GPU box as it is — fails here, on the CPU."""
    assert nat.build_id() == nat.source_build_id(), 'libocrhip.so is stale: run `python -c "import __graft_entry__ as g; g.build()"`'


def test_kernel_choice_reports_invalid_sizes_out_of_band():
    import pytest
    from lstm_ctc_ocr_amd import ops
    assert nat.lib().ocr_conv3x3_kernel_choice(0, 64, 4, 256, 512, 0, 0, 0) < 0          # 0..7 are kernel families
    with pytest.raises(nat.NativeError):
        ops.conv3x3_kernel_choice(64, 64, 0, 256, 512)


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
    f = ctypes.c_float
    assert lib.ocr_optim_scalar_count() == 76
    assert lib.ocr_optim_step_guarded(None, None, None, None, 16, 0, 16, f(0.0), f(10.0), 0, f(0.9), f(0.999), f(1e-8), None, None, 0, None) == 2
    assert lib.ocr_optim_step_guarded2(None, None, None, None, 16, 0, 16, f(0.0), f(10.0), 0, f(0.9), f(0.999), f(1e-8), None, None, 0, None, None) == 2
    assert lib.ocr_guard_flag(None, 1, None, None) == 2 and lib.ocr_guard_flag(None, 0, None, None) == 2
    assert lib.ocr_lstm_seq_test_skew(-1, -1) == 2 and lib.ocr_lstm_seq_test_skew(0, -1) == 0
    assert lib.ocr_occupy_cus(0, 256, 0, ctypes.c_float(10.0), None) == 2 and lib.ocr_occupy_cus(8, 2048, 0, ctypes.c_float(10.0), None) == 2
    assert lib.ocr_occupy_cus(8, 256, 161 * 1024, ctypes.c_float(10.0), None) == 2 and lib.ocr_occupy_cus(8, 256, 0, ctypes.c_float(-1.0), None) == 2
    assert lib.ocr_lstm_seq_supported(64, 256) == 1 and lib.ocr_lstm_seq_supported(64, 128) == 0
    assert lib.ocr_lstm_seq_supported(64 * 9, 256) == 0                            # would not be one workgroup per CU
    assert lib.ocr_lstm_seq_supported(64, 512) == 1 and lib.ocr_lstm_seq_supported(128, 512) == 1      # configs[4]: 512 units per direction
    assert lib.ocr_lstm_seq_supported(129, 512) == 0
    assert lib.ocr_lstm_seq_sync_words(64, 256) > 2 * 8 * 64 and lib.ocr_lstm_seq_sync_words(0, 256) == 0


def test_warpctc_abi_is_exported_with_its_own_prototypes():
    """include/warpctc_abi.h: the FFI the reference binds (network.py:6,653-654 -> libwarpctc compute_ctc_loss)."""
    import re
    import os
    import numpy as np
    from lstm_ctc_ocr_amd import warpctc
    lib = warpctc._lib()
    hdr = open(os.path.join(os.path.dirname(nat.HEADER_PATH), 'warpctc_abi.h')).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name in ('compute_ctc_loss', 'get_workspace_size', 'ctcGetStatusString', 'get_warpctc_version'):
        assert re.search(r"\b%s\s*\(" % name, hdr) and hasattr(lib, name), name
    assert lib.get_warpctc_version() == 2
    assert lib.ctcGetStatusString(0) == b'no error' and lib.ctcGetStatusString(3) == b'execution failed'
    assert ctypes.sizeof(warpctc.ctcOptions) == 24                      # {int; pad; union{unsigned, void*}; int; pad} on LP64
    ll = np.array([2, 1], np.int32); il = np.array([5, 3], np.int32)
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    opt = warpctc.ctcOptions(); opt.loc = warpctc.CTC_GPU; opt.blank_label = 0
    sz = ctypes.c_size_t(0)
    assert lib.get_workspace_size(ip(ll), ip(il), 5, 2, opt, ctypes.byref(sz)) == 0 and sz.value > 0
    assert lib.get_workspace_size(ip(ll), ip(il), 5, 0, opt, ctypes.byref(sz)) == 2                 # invalid value
    assert lib.get_workspace_size(None, ip(il), 5, 2, opt, ctypes.byref(sz)) == 2
    opt.loc = warpctc.CTC_CPU                   # there is no host implementation: loud status, never a silent CPU computation
    assert lib.get_workspace_size(ip(ll), ip(il), 5, 2, opt, ctypes.byref(sz)) == 3
    costs = np.zeros(2, np.float32)
    assert lib.compute_ctc_loss(None, None, ip(ll), ip(ll), ip(il), 5, 2, costs.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                None, opt) == 2


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


def test_job_table_records_match_the_c_structs():
    """The engine hands three kinds of job tables to one-launch kernels (re-packs, fills, weight-gradient slab reductions); their numpy
    record layouts must be the C structs' (csrc/nn_ops.hip PackJob, csrc/wgrad9.hip W9ReduceJob; the .hip side
    static_asserts the sizes), and host-side argument checks of the new entry points reject NULL tables before any launch."""
    from lstm_ctc_ocr_amd.engine import Engine
    off = lambda dt: {n: dt.fields[n][1] for n in dt.names}
    assert Engine.PACK_DTYPE.itemsize == 64
    assert Engine.W9_JOB_DTYPE.itemsize == 64
    assert off(Engine.W9_JOB_DTYPE) == dict(dw=0, part=8, dbias=16, cs_part=24, n4=32, slab4=40, S=48, rows=52, Cout=56, block_start=60)
    lib = nat.lib()
    assert lib.ocr_wgrad9_reduce_jobs(None, 1, 1, None) == 2
    job = (ctypes.c_ubyte * 64)()
    nb, deferred = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.ocr_conv3x3_wgrad_defer_bf16(None, None, None, None, 8, 8, 8, 64, 64, None, 0, ctypes.cast(job, ctypes.c_void_p),
                                            ctypes.byref(nb), ctypes.byref(deferred), None) == 2


def test_persistent_lstm_entry_points_check_their_arguments_on_the_host():
    """The persistent LSTM launches (plain and with flags) and their knobs refuse bad arguments before any launch; a kernel-family setter accepts
    only the two families that exist."""
    lib = nat.lib()
    assert lib.ocr_lstm_fwd_seq2(None, None, None, None, None, None, 64, 63, 256, 1.0, None, 1, None) == 2
    assert lib.ocr_lstm_bwd_seq2(None, 1024, 0, None, None, None, None, None, 64, 63, 256, None, 1, None) == 2
    assert lib.ocr_lstm_fwd_seq(None, None, None, None, None, None, 64, 63, 256, 1.0, None, None) == 2
    assert lib.ocr_lstm_fwd_seq_x(None, None, None, 512, None, None, None, None, None, 64, 63, 256, 1.0, None, 0, None) == 2
    assert lib.ocr_lstm_fwd_seq_x_supported(64, 256, 512) == 1 and lib.ocr_lstm_fwd_seq_x_supported(64, 256, 1024) == 1 and lib.ocr_lstm_fwd_seq_x_supported(64, 256, 768) == 0
    assert lib.ocr_set_lstm_ksplit(2) == 2 and lib.ocr_set_lstm_ksplit(4) == 0
    assert lib.ocr_lstm_seq_supported(64, 256) == 1 and lib.ocr_lstm_seq_supported(64, 100) == 0
    # the training form of conv1 + pool: the all-ones region must be 16-byte granules
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.ocr_conv1_pool_fwd_train(p, p, p, p, 1, 8, 8, 64, None, None, 0, p, 6, None) == 2
