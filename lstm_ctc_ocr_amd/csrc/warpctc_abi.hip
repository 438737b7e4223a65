// warp-ctc's C ABI (include/warpctc_abi.h) over the gfx950 CTC kernels of ctc.hip.
// Replaces libwarpctc behind `warpctc_tensorflow.ctc` (reference lib/networks/network.py:6,653-654): host label / length
// arrays and host costs as in warp-ctc's GPU path; they are staged through the caller's workspace, so the library still
// allocates nothing and keeps no state.
#include "common.h"
#include "../../include/warpctc_abi.h"
#include <algorithm>

extern "C" int ocr_ctc_workspace_size(int max_label_len, int max_time, int minibatch, size_t* bytes);
extern "C" int ocr_ctc_loss(const float* activations, float* gradients, const int* flat_labels, const int* label_lengths,
                            const int* input_lengths, int alphabet_size, int minibatch, int max_time, int max_label_len,
                            int blank_label, float* costs, void* workspace, void* stream);

namespace {
struct Extents { int max_t, max_l; long total_l; bool ok; };
Extents extents(const int* label_lengths, const int* input_lengths, int minibatch) {
    Extents e = {0, 0, 0, true};
    for (int i = 0; i < minibatch; ++i) {
        if (label_lengths[i] < 0 || input_lengths[i] <= 0) e.ok = false;
        e.max_t = std::max(e.max_t, input_lengths[i]);
        e.max_l = std::max(e.max_l, label_lengths[i]);
        e.total_l += label_lengths[i];
    }
    return e;
}
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// staging area in front of the kernels' own workspace: labels, label lengths, input lengths, costs
size_t staging_bytes(long total_l, int minibatch) {
    return align256((size_t)std::max(total_l, 1L) * 4) + 3 * align256((size_t)minibatch * 4);
}
}  // namespace

extern "C" int get_warpctc_version(void) { return 2; }

extern "C" const char* ctcGetStatusString(ctcStatus_t status) {
    switch (status) {
        case CTC_STATUS_SUCCESS: return "no error";
        case CTC_STATUS_MEMOPS_FAILED: return "cuda memcpy or memset failed";
        case CTC_STATUS_INVALID_VALUE: return "invalid value";
        case CTC_STATUS_EXECUTION_FAILED: return "execution failed";
        default: return "unknown error";
    }
}

extern "C" ctcStatus_t get_workspace_size(const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                                          int minibatch, struct ctcOptions info, size_t* size_bytes) {
    if (!label_lengths || !input_lengths || !size_bytes || alphabet_size <= 0 || minibatch <= 0) return CTC_STATUS_INVALID_VALUE;
    if (info.loc != CTC_GPU) return CTC_STATUS_EXECUTION_FAILED;          // no host implementation in this library
    Extents e = extents(label_lengths, input_lengths, minibatch);
    if (!e.ok) return CTC_STATUS_INVALID_VALUE;
    size_t inner = 0;
    if (ocr_ctc_workspace_size(e.max_l, e.max_t, minibatch, &inner) != OCR_OK) return CTC_STATUS_INVALID_VALUE;
    *size_bytes = staging_bytes(e.total_l, minibatch) + inner;
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t compute_ctc_loss(const float* const activations, float* gradients, const int* const flat_labels,
                                        const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                                        int minibatch, float* costs, void* workspace, struct ctcOptions options) {
    if (!activations || !flat_labels || !label_lengths || !input_lengths || !costs || !workspace || alphabet_size <= 0 ||
        minibatch <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (options.loc != CTC_GPU) return CTC_STATUS_EXECUTION_FAILED;
    if (options.blank_label < 0 || options.blank_label >= alphabet_size) return CTC_STATUS_INVALID_VALUE;
    Extents e = extents(label_lengths, input_lengths, minibatch);
    if (!e.ok) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)options.stream;
    unsigned char* ws = (unsigned char*)workspace;
    int* d_labels = (int*)ws;               ws += align256((size_t)std::max(e.total_l, 1L) * 4);
    int* d_llen = (int*)ws;                 ws += align256((size_t)minibatch * 4);
    int* d_ilen = (int*)ws;                 ws += align256((size_t)minibatch * 4);
    float* d_costs = (float*)ws;            ws += align256((size_t)minibatch * 4);
    if (e.total_l > 0 && hipMemcpyAsync(d_labels, flat_labels, (size_t)e.total_l * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    if (hipMemcpyAsync(d_llen, label_lengths, (size_t)minibatch * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d_ilen, input_lengths, (size_t)minibatch * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    int rc = ocr_ctc_loss(activations, gradients, d_labels, d_llen, d_ilen, alphabet_size, minibatch, e.max_t, e.max_l,
                          options.blank_label, d_costs, ws, stream);
    if (rc != OCR_OK) return (ctcStatus_t)rc;
    if (hipMemcpyAsync(costs, d_costs, (size_t)minibatch * 4, hipMemcpyDeviceToHost, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
    return CTC_STATUS_SUCCESS;
}
