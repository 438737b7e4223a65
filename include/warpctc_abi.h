/* warp-ctc's own C ABI, exported by libocrhip.so with warp-ctc's exact prototypes — the FFI the reference really binds:
 * `import warpctc_tensorflow` -> libwarpctc `compute_ctc_loss` (/root/reference/lib/networks/network.py:6,653-654;
 * README.md:20 pins warp-ctc only as "master").  The declarations restate baidu-research/warp-ctc `include/ctc.h`
 * (not vendored in the reference): a caller compiled or ctypes-bound against warp-ctc links against this library unchanged.
 *
 * GPU semantics of warp-ctc are kept:
 *   activations / gradients / workspace : DEVICE memory, activations f32 [maxT, minibatch, alphabet_size] with
 *                                         maxT = max(input_lengths), unnormalised (softmax is applied inside);
 *   flat_labels / label_lengths / input_lengths / costs : HOST memory (warp-ctc: "always in CPU memory");
 *   gradients == NULL -> score only;  options.loc must be CTC_GPU (this library has no CPU path: CTC_CPU returns
 *   CTC_STATUS_EXECUTION_FAILED, loudly, instead of computing on the host);  options.stream is a hipStream_t;
 *   the call returns after the costs have arrived in host memory (warp-ctc synchronises its stream the same way).
 * Infeasible samples (label longer than the input allows) get cost 0 and a zero gradient, like warp-ctc.
 * The training loop itself uses ocr_ctc_loss_train (include/ocr_hip.h), which keeps labels and costs in HBM and never syncs.
 */
#ifndef WARPCTC_ABI_H
#define WARPCTC_ABI_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CTC_STATUS_SUCCESS = 0,
    CTC_STATUS_MEMOPS_FAILED = 1,
    CTC_STATUS_INVALID_VALUE = 2,
    CTC_STATUS_EXECUTION_FAILED = 3,
    CTC_STATUS_UNKNOWN_ERROR = 4
} ctcStatus_t;

typedef enum { CTC_CPU = 0, CTC_GPU = 1 } ctcComputeLocation;

struct ctcOptions {
    ctcComputeLocation loc;     /* must be CTC_GPU */
    union {
        unsigned int num_threads;   /* CTC_CPU only (unsupported) */
        void* stream;               /* hipStream_t (warp-ctc: CUstream) */
    };
    int blank_label;            /* warp-ctc default 0 — what the reference trains with */
};

int get_warpctc_version(void);
const char* ctcGetStatusString(ctcStatus_t status);

ctcStatus_t compute_ctc_loss(const float* const activations, float* gradients, const int* const flat_labels,
                             const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                             int minibatch, float* costs, void* workspace, struct ctcOptions options);

ctcStatus_t get_workspace_size(const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                               int minibatch, struct ctcOptions info, size_t* size_bytes);

#ifdef __cplusplus
}
#endif
#endif
