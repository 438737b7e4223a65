"""ctypes binding of libocrhip.so — the only way the Python host reaches the GPU kernels.

There is deliberately NO fallback: if the shared library is missing or a call returns a non-zero status the
caller gets an exception.  Pointers are raw device addresses (``tensor.data_ptr()``), the stream is the raw
``hipStream_t`` of torch's current stream, exactly what a C/C++ caller of include/ocr_hip.h would pass.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCR_NATIVE_LIB: load another build of the same library — the `make EXPERIMENTS=1` flavour with the rejected variants, for the A/B tools
LIB_PATH = os.environ.get("OCR_NATIVE_LIB") or os.path.join(_HERE, "libocrhip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ocr_hip.h")

EPI_BIAS, EPI_RELU, EPI_OUT_F32, EPI_MASK, EPI_ROWSWAP, EPI_ACCUM = 1, 2, 4, 16, 32, 64

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double

_SIGS = {
    "ocr_abi_version": ([], _I),
    "ocr_build_id": ([], ctypes.c_char_p),
    "ocr_status_string": ([_I], ctypes.c_char_p),
    "ocr_ctc_workspace_size": ([_I, _I, _I, ctypes.POINTER(ctypes.c_size_t)], _I),
    "ocr_ctc_loss": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    "ocr_set_ctc_engine": ([_I], _I),
    "ocr_ctc_debug": ([_P], _I),
    "ocr_ctc_train_supported": ([_I, _I, _I], _I),
    "ocr_ctc_loss_train": ([_P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P], _I),
    "ocr_ctc_greedy_decode": ([_P, _P, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    "ocr_ctc_beam_workspace_size": ([_I, _I, _I, _I, ctypes.POINTER(ctypes.c_size_t)], _I),
    "ocr_ctc_beam_decode": ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, ctypes.c_size_t, _P], _I),
    "ocr_gemm_nt_bf16": ([_P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P], _I),
    "ocr_set_gemm_engine": ([_I], _I),
    "ocr_gemm_tn_batched_bf16": ([_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _F, _I, _P, _L, _P], _I),
    "ocr_gemm_tn_jobs_supported": ([_P, _I], _I),
    "ocr_gemm_tn_jobs_bf16": ([_P, _I, _P], _I),
    "ocr_lstm_xh": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "ocr_set_wgrad_engine": ([_I], _I),
    "ocr_conv3x3_bf16": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P], _I),
    "ocr_conv_halo_clock_debug": ([_P], _I),
    "ocr_occupy_cus": ([_I, _I, _I, _F, _P], _I),
    "ocr_conv_ws_debug": ([_P], _I),
    "ocr_conv3x3_accum_supported": ([_I, _I, _I, _I, _I], _I),
    "ocr_conv3x3_kernel_choice": ([_I, _I, _I, _I, _I, _I, _I, _I], _I),
    "ocr_conv3x3_pool_supported": ([_I, _I, _I, _I, _I, _I, _I], _I),
    "ocr_conv3x3_relu_pool_bf16": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P], _I),
    "ocr_gemm_tn_bf16": ([_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _L, _F, _I, _P, _P], _I),
    "ocr_conv3x3_wgrad_bf16": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "ocr_conv3x3_wgrad_workspace_size": ([_I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_size_t)], _I),
    "ocr_conv3x3_wgrad_ws_bf16": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _P], _I),
    "ocr_conv3x3_wgrad_defer_bf16": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _P, ctypes.POINTER(_I),
                                      ctypes.POINTER(_I), _P], _I),
    "ocr_wgrad9_reduce_jobs": ([_P, _I, _I, _P], _I),
    "ocr_conv1_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "ocr_conv1_wgrad": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "ocr_eltwise_bf16": ([_I, _P, _P, _P, _L, _P], _I),
    "ocr_conv1_pool_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "ocr_conv1_pool_fwd_train": ([_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _L, _P, _L, _P], _I),
    "ocr_conv1_pool_bwd_codes": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P], _I),
    "ocr_conv1_pool_bwd_slab_rows": ([_I, _I, _I], _I),
    "ocr_conv1_pool_bwd_slab": ([_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P], _I),
    "ocr_conv1_pool_bwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "ocr_maxpool_fwd": ([_P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "ocr_maxpool_bwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "ocr_bn_workspace_bytes": ([_L, _I], ctypes.c_size_t),
    "ocr_bn_train_fwd": ([_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P, _P, _P], _I),
    "ocr_bn_train_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P], _I),
    "ocr_bn_train_fwd2": ([_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P, _P, _I, _P, _P], _I),
    "ocr_bn_train_bwd2": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _I, _P], _I),
    "ocr_conv3x3_bnbwd_rows": ([_I, _I, _I, _I, _I], _I),
    "ocr_conv3x3_dgrad_bnbwd_bf16": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "ocr_conv3x3_stats_rows": ([_I, _I, _I, _I, _I, _I], _I),
    "ocr_conv3x3_bf16_stats": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P], _I),
    "ocr_bn_infer_fwd": ([_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P], _I),
    "ocr_bn_infer_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P], _I),
    "ocr_dropout_bf16": ([_P, _P, _L, ctypes.c_uint, _P, _F, _P], _I),
    "ocr_avgpool_bf16": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "ocr_subsample_bf16": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "ocr_softmax_f32": ([_P, _P, _L, _I, _P], _I),
    "ocr_colsum_bf16": ([_P, _P, _L, _I, _L, _P], _I),
    "ocr_pack_transpose": ([_P, _P, _I, _I, _L, _I, _P], _I),
    "ocr_pack_conv_dgrad": ([_P, _P, _I, _I, _P], _I),
    "ocr_pack_jobs": ([_P, _I, _I, _P], _I),
    "ocr_cast_f32_bf16": ([_P, _P, _L, _P], _I),
    "ocr_u8_to_unit_f32": ([_P, _P, _L, _P], _I),
    "ocr_step_report": ([_P, _I, _P, _P, _I, _P, _P], _I),
    "ocr_bind_batch": ([_P, _I, _P, _L, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P], _I),
    "ocr_captcha_synth": ([_P, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P], _I),
    "ocr_cast2d_f32_bf16": ([_P, _L, _P, _L, _I, _I, _P], _I),
    "ocr_tnc_to_ntc_bf16": ([_P, _P, _I, _I, _I, _F, _P], _I),
    "ocr_conv5_col2im": ([_P, _P, _I, _I, _I, _P], _I),
    "ocr_lstm_fwd_step": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P], _I),
    "ocr_lstm_bwd_step": ([_P, _L, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "ocr_lstm_seq_supported": ([_I, _I], _I),
    "ocr_lstm_seq_debug": ([_P], _I),
    "ocr_lstm_seq_test_skew": ([_I, _I], _I),
    "ocr_lstm_seq_sync_words": ([_I, _I], _L),
    "ocr_lstm_fwd_seq": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P], _I),
    "ocr_lstm_bwd_seq": ([_P, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P], _I),
    "ocr_lstm_fwd_seq2": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _P], _I),
    "ocr_lstm_fwd_seq_x_supported": ([_I, _I, _I], _I),
    "ocr_lstm_fwd_seq_x": ([_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _P], _I),
    "ocr_lstm_bwd_seq2": ([_P, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P], _I),
    "ocr_lstm_hprev": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "ocr_lstm_pack_bias": ([_P, _P, _P, _I, _I, _P], _I),
    "ocr_optim_scalar_count": ([], _I),
    "ocr_optim_init": ([_P, _D, _P], _I),
    "ocr_optim_set_lr": ([_P, _D, _I, _P], _I),
    "ocr_optim_step": ([_P, _P, _P, _P, _L, _L, _L, _F, _F, _I, _F, _F, _F, _P, _P], _I),
    "ocr_optim_step_guarded": ([_P, _P, _P, _P, _L, _L, _L, _F, _F, _I, _F, _F, _F, _P, _P, _I, _P], _I),
    "ocr_optim_step_guarded2": ([_P, _P, _P, _P, _L, _L, _L, _F, _F, _I, _F, _F, _F, _P, _P, _I, _P, _P], _I),
    "ocr_guard_flag": ([_P, _I, _P, _P], _I),
    "ocr_wgrad9_debug": ([_P], _I),
    "ocr_probe_tr16": ([_P, _P, _P], _I),
    "ocr_set_lstm_proto": ([_I], _I),
    "ocr_set_lstm_ksplit": ([_I], _I),
    "ocr_probe_xcc": ([_P, _I, _I, _P], _I),
    "ocr_mfma_busy_probe": ([_P, _I, _I, _I, _P, _P], _I),
}


class NativeError(RuntimeError):
    pass


def declared_symbols(header=HEADER_PATH):
    """Entry points declared by include/ocr_hip.h (used by the no-GPU export test)."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ocr_[a-z0-9_]+)\s*\(", text)))


def source_build_id(csrc=os.path.join(_HERE, "csrc"), experiments=False):
    """The id `make` compiles into the library (csrc/Makefile: sha256 over the sorted sources, common.h and the Makefile, 16 hex digits;
    the experiments flavour adds experiments/conv_*.hip and the suffix '-exp'), recomputed from the source tree: equal to build_id()
    unless the .so is stale."""
    import glob
    import hashlib
    rel = [os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip"))]
    if experiments:
        rel += ["experiments/" + os.path.basename(f) for f in glob.glob(os.path.join(csrc, "experiments", "conv_*.hip"))]
    h = hashlib.sha256()
    for f in sorted(rel) + ["common.h", "Makefile"]:          # make's $(sort ...) orders the same byte-wise way
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16] + ("-exp" if experiments else "")


def build_id():
    """Source hash the loaded library was built from (+ '-exp' for the experiments flavour)."""
    return lib().ocr_build_id().decode()


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libocrhip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU/eager fallback for the hot path)" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        for env, setter in (("OCR_GEMM_ENGINE", "ocr_set_gemm_engine"), ("OCR_WGRAD_ENGINE", "ocr_set_wgrad_engine")):
            if os.environ.get(env):                       # A/B knobs for experiments (see include/ocr_hip.h)
                getattr(_lib, setter)(int(os.environ[env]))
    return _lib


def status_string(code):
    return lib().ocr_status_string(code).decode()


def call(name, *args):
    """Invoke an int-status entry point; raise on a non-zero status."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise NativeError("%s failed: status %d (%s)" % (name, rc, status_string(rc)))


def ptr(t):
    """Device (or host) address of a torch tensor / None -> NULL."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
