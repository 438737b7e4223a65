/* libocrhip — C ABI of the MI355X (gfx950) CRNN-OCR hot path.
 *
 * Conventions (the same ones baidu warp-ctc's `compute_ctc_loss` uses, which is the only FFI the reference
 * itself crosses — /root/reference/lib/networks/network.py:6,653-654):
 *   - every pointer is a caller-owned DEVICE pointer (HBM) unless stated otherwise; the library keeps no state,
 *     allocates nothing and never synchronises; all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return value is an int status: 0 ok, 1 memory-op failed, 2 invalid argument, 3 launch failed
 *     (warp-ctc's ctcStatus_t numbering); no exceptions, no torch types;
 *   - "bf16" buffers hold raw bfloat16 bits (uint16_t); activations use the reference layout [N, W, H, C]
 *     (image width W = time is TF's "height" axis, the 32-pixel feature axis H is TF's "width": gen.py:63-64).
 *
 * Each entry point cites the reference line whose computation it replaces.
 */
#ifndef OCR_HIP_H
#define OCR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { OCR_STATUS_OK = 0, OCR_STATUS_MEMOPS = 1, OCR_STATUS_INVALID = 2, OCR_STATUS_EXEC = 3 };

/* epilogue flags of the GEMM / implicit-GEMM convolution engine */
enum {
    OCR_EPI_BIAS = 1, OCR_EPI_RELU = 2, OCR_EPI_OUT_F32 = 4, /* 8 reserved (split-K atomics, set internally) */
    OCR_EPI_MASK = 16, OCR_EPI_ROWSWAP = 32, OCR_EPI_ACCUM = 64
};

const char* ocr_status_string(int status);           /* warp-ctc: ctcGetStatusString */
int ocr_abi_version(void);
/* first 16 hex digits of sha256 over the sources this library was built from (csrc/Makefile; "-exp" appended by the experiments
 * flavour): profiles are pinned to it, and lstm_ctc_ocr_amd._native.source_build_id() recomputes it from the tree to detect a stale .so */
const char* ocr_build_id(void);

/* ---- CTC (replaces warpctc_tensorflow.ctc, network.py:653-654; warp-ctc get_workspace_size/compute_ctc_loss) */
int ocr_ctc_workspace_size(int max_label_len, int max_time, int minibatch, size_t* bytes);
/* activations/gradients: f32 [max_time, minibatch, alphabet_size], unnormalised; gradients may be NULL (score only).
 * flat_labels / label_lengths / input_lengths are DEVICE int32 arrays (warp-ctc's GPU path takes them from the host;
 * keeping them in HBM removes the per-step host sync of train.py:130).  costs: f32 [minibatch]. */
int ocr_ctc_loss(const float* activations, float* gradients, const int* flat_labels, const int* label_lengths,
                 const int* input_lengths, int alphabet_size, int minibatch, int max_time, int max_label_len,
                 int blank_label, float* costs, void* workspace, void* stream);
/* training form: one launch producing costs and scale * gradient as bf16 [minibatch][max_time][alphabet] (what the backward
 * GEMMs read); label offsets computed in-kernel.  ocr_ctc_train_supported() == 0 -> use ocr_ctc_loss + ocr_tnc_to_ntc_bf16 */
/* diagnostic: device int64[5] receiving 100 MHz wall-clock stamps at the phase boundaries of sample 0 of the fast kernel (NULL = off) */
int ocr_ctc_debug(void* dbg);
int ocr_ctc_train_supported(int alphabet_size, int max_time, int max_label_len);
int ocr_ctc_loss_train(const float* activations, void* grad_ntc_bf16, float scale, const int* flat_labels,
                       const int* label_lengths, const int* input_lengths, int alphabet_size, int minibatch, int max_time,
                       int max_label_len, int blank_label, float* costs, void* stream);
int ocr_set_ctc_engine(int fast);   /* 1 (default): 4-wave LDS-resident kernel where it fits; 0: one-wave general kernel */
/* best-path decode (argmax, collapse repeats, drop blank): the greedy counterpart of
 * tf.nn.ctc_beam_search_decoder + sparse_tensor_to_dense(default 0) at network.py:656-657 / test.py:30-31.
 * decoded: int32 [minibatch, max_time] padded with pad_value; decoded_lengths: int32 [minibatch]. */
int ocr_ctc_greedy_decode(const float* activations, const int* input_lengths, int alphabet_size, int minibatch,
                          int max_time, int blank_label, int pad_value, int* decoded, int* decoded_lengths,
                          void* stream);

/* TF-semantics prefix beam search: tf.nn.ctc_beam_search_decoder(inputs, seq_len, beam_width=100, top_paths=1,
 * merge_repeated=True) as called at network.py:656 / test.py:30 — BLANK = alphabet_size-1, output dense + padded.
 * beam_width <= 128; neg_log_prob (may be NULL): f32 [minibatch].  Workspace from ocr_ctc_beam_workspace_size. */
int ocr_ctc_beam_workspace_size(int alphabet_size, int minibatch, int max_time, int beam_width, size_t* bytes);
int ocr_ctc_beam_decode(const float* activations, const int* input_lengths, int alphabet_size, int minibatch,
                        int max_time, int beam_width, int merge_repeated, int pad_value, int* decoded,
                        int* decoded_lengths, float* neg_log_prob, void* workspace, size_t workspace_bytes, void* stream);

/* ---- dense contractions (tf.nn.conv2d network.py:166, tf.matmul :126, LSTMCell matmul :104-107) ------------- */
/* out[m][n] = sum_k P[m][k] * Q[n][k] (+bias[n]) ; bf16 operands, fp32 accumulate, K % 8 == 0, N % 4 == 0.
 * physical P row = m + (m / row_group) * row_skip when row_group > 0 (conv5's overlapping 2x2 VALID windows). */
int ocr_gemm_nt_bf16(const void* P, long ldp, const void* Q, long ldq, void* out, long ldo, int M, int N, int K,
                     const float* bias, const void* mask, long ldmask, int flags, int splits, int row_group,
                     int row_skip, int swap_inner, int swap_outer, void* stream);
/* engine selector for A/B measurements: 1 (default) = tap-reuse convolution kernels (conv_k3 / conv_k2 / conv_halo, chosen per shape:
 * DESIGN section 3) + 256-row LDS-DMA GEMM tiles, 0 = 128x128 register-staged tiles, 2 / 3 = LDS-DMA tiles without the tap-reuse kernels
 * (4 = round 2's ping-pong halo conv: experiments build only) */
int ocr_set_gemm_engine(int use_large_tile);
/* 3x3 SAME stride-1 convolution, x bf16 [Nb,W,H,Cin], wpack bf16 [Cout][3][3][Cin], y [Nb,W,H,Cout]
 * (conv_single network.py:160-191; also its data gradient with flipped/transposed weights).  The result does not depend on which kernel
 * the dispatcher takes beyond fp32 summation order (environment OCR_CONV_K2 / OCR_CONV_K3 / OCR_K2_CFG select for the parity tests). */
int ocr_conv3x3_bf16(const void* x, const void* wpack, void* y, int Nb, int W, int H, int Cin, int Cout,
                     const float* bias, const void* mask, int flags, void* stream);
/* The convolution that also leaves the batch-norm statistics of its output behind (network.py:173-178: conv -> bias -> batch_norm):
 * partials = float [rows][2][Cout] with rows = ocr_conv3x3_stats_rows(...) = Nb*W*H / 256: per 256-pixel tile the per-channel sum and sum
 * of squares of the bf16 values that were stored (what a statistics pass over y would read).  Only where the plane-layout kernels take the
 * shape: the query returns 0 otherwise and the call OCR_STATUS_INVALID (run ocr_conv3x3_bf16 + ocr_bn_train_fwd then).  flags: BIAS / RELU. */
int ocr_conv3x3_stats_rows(int Nb, int W, int H, int Cin, int Cout, int flags);
int ocr_conv3x3_bf16_stats(const void* x, const void* wpack, void* y, int Nb, int W, int H, int Cin, int Cout, const float* bias, int flags,
                           float* partials, void* stream);
/* Which kernel family the two convolution entry points run for a shape — a host-only query (nothing is launched, works without a GPU):
 * 0 generic GEMM engines, 1 conv_halo, 2 / 3 conv_k2 tile A (256 x 128) / D (256 x 64), 4 / 5 conv_k3 A / D, 6 / 7 conv_k3w (tiles that
 * cross image boundaries) A / D, 8 conv_ws (weights in registers, persistent over the pixel tiles: Cin = 64 / 128 at H = 16 / 8); a NEGATIVE value (-OCR_STATUS_INVALID) for non-positive sizes.  flags as for ocr_conv3x3_bf16;
 * (kw, kh) = (0, 0) or the window of the fused max-pool. */
int ocr_conv3x3_kernel_choice(int Nb, int W, int H, int Cin, int Cout, int flags, int kw, int kh);
/* non-zero: ocr_conv3x3_bf16 accepts OCR_EPI_ACCUM for this shape (y (bf16) += result: the data gradient of a tensor with several
 * consumers is added to what was already delivered, no scratch tensor + add pass) */
int ocr_conv3x3_accum_supported(int Nb, int W, int H, int Cin, int Cout);
/* diagnostic: workgroup 0 of the convolution kernels (conv_halo, conv_k3 / conv_k3w) stamps {shader clock counter, 100 MHz wall clock} at entry ([0], [1]) and exit ([2], [3])
 * into dbg (device int64[8]; NULL = off) and adds its lifetime to [4] (shader clocks), [5] (wall ticks), [6] (launches) */
int ocr_conv_halo_clock_debug(void* dbg);
/* diagnostic: every workgroup of the weight-stationary convolution kernel (conv_ws) stamps the 100 MHz wall clock per phase and tile into
 * dbg (device int64[workgroups * 64]; NULL = off) — tools/ws_phases.py */
int ocr_conv_ws_debug(void* dbg);
/* conv3x3 + bias + ReLU AND the max-pool behind it from one epilogue (LSTM_train.py:26-33): y [Nb,W,H,Cout] and pooled
 * [Nb, W/kw, H/kh, Cout]; (kw, kh) = (1, 2) (feature axis) or (2, 2).  ocr_conv3x3_pool_supported() != 0 tells whether the shape is
 * covered; otherwise run ocr_conv3x3_bf16 + ocr_maxpool_fwd (ocr_conv3x3_relu_pool_bf16 then returns 2). */
int ocr_conv3x3_pool_supported(int Nb, int W, int H, int Cin, int Cout, int kw, int kh);
int ocr_conv3x3_relu_pool_bf16(const void* x, const void* wpack, void* y, void* pooled, int Nb, int W, int H, int Cin, int Cout,
                               const float* bias, int kw, int kh, void* stream);
/* out[I][ldo] (f32) += scale * A^T B, A bf16 [Mk][lda], B bf16 [Mk][ldb]  (weight gradients of matmul layers);
 * colsum (may be NULL): colsum[j] += scale * sum_m B[m][j] — the bias gradient, produced by the same pass */
int ocr_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* out, long ldo, int Mk, int I, int J,
                     int row_group, int row_skip, long a_row_off, float scale, int splits, float* colsum, void* stream);
/* nbatch products of one shape in one launch; problem b: A + b*strideA, B + b*strideB, out + b*strideOut, colsum + b*strideColsum */
int ocr_gemm_tn_batched_bf16(const void* A, long lda, long strideA, const void* B, long ldb, long strideB, float* out, long ldo,
                             long strideOut, int Mk, int I, int J, int nbatch, float scale, int splits, float* colsum,
                             long strideColsum, void* stream);
/* Round 4: several plain weight-gradient products in ONE launch of the ping-pong kernel (csrc/gemm_tn3.hip): one workgroup per 128 x 128
 * output tile over the whole contraction - no split over m, no atomics, bit-reproducible.  For every job and b < nbatch:
 *     out_b[I][J] += scale * A_b^T B_b,   colsum_b[J] += scale * column sums of B_b   (colsum may be NULL)
 * with A_b = A + b*strideA ([Mk rows][lda], rows in groups of row_group with row_skip unused rows behind each group; 0 = plain), B_b = B +
 * b*strideB, element strides.  jobs is a HOST array of 1 or 2 descriptors.  Needs I % 128 == 0, J % 128 == 0, Mk >= 256, 16-byte aligned
 * rows: ocr_gemm_tn_jobs_supported tells beforehand (host-only); OCR_STATUS_INVALID otherwise (use ocr_gemm_tn_bf16 / _batched_bf16). */
typedef struct ocr_tn_job {
    const void* A; long lda; long strideA;
    const void* B; long ldb; long strideB;
    float* out; long ldo; long strideOut;
    float* colsum; long strideColsum;
    int Mk, I, J, nbatch;
    int row_group, row_skip;
    float scale; int reserved;
} ocr_tn_job;
int ocr_gemm_tn_jobs_supported(const ocr_tn_job* jobs, int njobs);
int ocr_gemm_tn_jobs_bf16(const ocr_tn_job* jobs, int njobs, void* stream);
/* A/B knob: 2 (default) = nine-tap slab kernel for conv weight gradients when a workspace is given (else as 1);
 * 1 = LDS-DMA per-tap tiles + fp32 atomics where I,J % 128 == 0; 0 = register-staged kernel */
int ocr_set_wgrad_engine(int engine);
/* dw f32 [3][3][Cin][Cout] (TF HWIO) += conv weight gradient; dbias (may be NULL) += sum over pixels of dy */
int ocr_conv3x3_wgrad_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin,
                           int Cout, int splits, void* stream);

/* the same through a caller-owned workspace of ocr_conv3x3_wgrad_workspace_size() bytes (0: shape not covered, the call then
 * behaves like ocr_conv3x3_wgrad_bf16): all nine taps from one staged tile, per-split partial slabs summed in a fixed order by a
 * second kernel — no atomics, bit-reproducible.  The workspace is scratch: neither zeroed by the caller nor kept. */
int ocr_conv3x3_wgrad_workspace_size(int Nb, int W, int H, int Cin, int Cout, size_t* bytes);
int ocr_conv3x3_wgrad_ws_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin,
                              int Cout, int splits, void* workspace, size_t workspace_bytes, void* stream);
/* Deferred form: a backward pass may keep the partial slabs of all its layers (one workspace per layer) and reduce them in ONE launch at
 * its end — the per-layer reduce kernels are short and each is a dependent-kernel boundary.  Where the slab kernel covers the shape only it
 * runs here: `job` (64 bytes of HOST memory) receives the pending reduction, *job_blocks its block count, *deferred = 1, and the
 * workspace must stay untouched until ocr_wgrad9_reduce_jobs has run; otherwise the whole weight gradient is computed at once
 * (*deferred = 0).  ocr_wgrad9_reduce_jobs takes a DEVICE table of such jobs whose int at byte offset 60 (`block_start`) the caller
 * has set to the running sum of the preceding jobs' block counts; results are bit-identical to the per-layer form. */
int ocr_conv3x3_wgrad_defer_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin, int Cout,
                                 void* workspace, size_t workspace_bytes, void* job, int* job_blocks, int* deferred, void* stream);
int ocr_wgrad9_reduce_jobs(const void* jobs, int njobs, int total_blocks, void* stream);

/* ---- conv1 (Cin = 1), pooling, batch-norm, reductions, packing (network.py:160-191, 343-350, 176-178) -------- */
int ocr_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int Nb, int W, int H, int Cout,
                  int relu, void* stream);
int ocr_conv1_wgrad(const float* x, const void* dz, float* dw, float* db, int Nb, int W, int H, int Cout,
                    void* stream);
/* element-wise bf16 (n % 8 == 0): op 0 out = a + b (Network.add), 1 out = relu(a) (Network.relu), 2 out = b > 0 ? a : 0,
 * 3 out = relu(a + b) (add + relu of a residual block in one pass), 4 out += b > 0 ? a : 0 (ReLU backward accumulated) */
int ocr_eltwise_bf16(int op, const void* a, const void* b, void* out, long n, void* stream);
/* conv1 + ReLU + 2x2 max-pool in one pass (LSTM_train.py:24-25); the backward recomputes the window instead of reading a
 * stored 67 MB activation: p / dp are the POOLED map [Nb, W/2, H/2, Cout] */
int ocr_conv1_pool_fwd(const float* x, const float* w, const float* bias, void* p, int Nb, int W, int H, int Cout, void* stream);
/* training form.  codes (may be NULL; Cout == 64): uint32 [Nb * W/2 * H/2][8], 4 bits per pooled output = position of the window's first
 * maximum (bf16-rounded values, TF scan order) | ReLU bit << 2, consumed by ocr_conv1_pool_bwd_codes (bit-identical gradients without
 * recomputing the windows).  zero (may be NULL): fp32 [zero_n] cleared by the same launch (zero_n % 4 == 0, 16-byte aligned): the step's
 * flat gradient buffer.  ones (may be NULL): ones_n 32-bit words (% 4 == 0, 16-byte aligned) set to 0xFFFFFFFF by the same launch: the
 * hand-off blocks of the step's persistent LSTM launches (OCR_LSTM_PREPARED below). */
int ocr_conv1_pool_fwd_train(const float* x, const float* w, const float* bias, void* p, int Nb, int W, int H, int Cout,
                             void* codes, float* zero, long zero_n, void* ones, long ones_n, void* stream);
int ocr_conv1_pool_bwd_codes(const float* x, const float* w, const float* bias, const void* dp, float* dw, float* db,
                             int Nb, int W, int H, int Cout, const void* codes, void* stream);
/* slab form: no atomics - block b leaves {dW [9][64] | db [64]} (fp32) in slab[b][640], b < ocr_conv1_pool_bwd_slab_rows(Nb, W, H); the rows
 * are added by ocr_wgrad9_reduce_jobs (jobs with n4 = 144 / 16, slab4 = 160, S = rows): bit-reproducible.  codes may be NULL (recompute). */
int ocr_conv1_pool_bwd_slab_rows(int Nb, int W, int H);
int ocr_conv1_pool_bwd_slab(const float* x, const float* w, const float* bias, const void* dp, int Nb, int W, int H, int Cout,
                            const void* codes, float* slab, void* stream);
int ocr_conv1_pool_bwd(const float* x, const float* w, const float* bias, const void* dp, float* dw, float* db, int Nb,
                       int W, int H, int Cout, void* stream);
int ocr_maxpool_fwd(const void* x, void* y, int Nb, int W, int H, int C, int kw, int kh, void* stream);
int ocr_maxpool_bwd(const void* x, const void* dy, void* dx, int Nb, int W, int H, int C, int kw, int kh,
                    int relu_mask, void* stream);
/* training-mode batch norm over rows of x[M][C] (network.py:176-178): batch statistics, biased variance.  `workspace` is
 * ocr_bn_workspace_bytes(M, C) bytes of caller-owned scratch (per-block partial sums; neither zeroed nor kept) */
size_t ocr_bn_workspace_bytes(long M, int C);
/* residual (bf16 [M][C], may be NULL): y = [relu](bf16(bn(x)) + residual) — the tail of a residual block (bn, Network.add,
 * Network.relu) in the apply pass, bit-identical to the three separate passes */
int ocr_bn_train_fwd(const void* x, void* y, const float* gamma, const float* beta, float* save_mean,
                     float* save_rstd, long M, int C, float eps, int relu, void* workspace, const void* residual, void* stream);
int ocr_bn_train_bwd(const void* x, const void* y, const void* dy, void* dx, const float* gamma,
                     const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, long M, int C,
                     int relu, void* workspace, void* stream);
/* Round 4 forms.  partial_rows > 0: the workspace already holds that many rows [rows][2][C] of per-tile sums / sums of squares written by
 * the producing convolution's epilogue (ocr_conv3x3_bf16_stats) - no statistics pass over x.  pooled (may be NULL): the 1 x 2 max-pool over
 * row pairs (2q, 2q + 1) that follows the layer (LSTM_train.py:33) is written by the apply pass too ([M / 2][C]; M even, no residual);
 * bit-identical to ocr_maxpool_fwd on y.  pooled_dy != 0: `dy` is the gradient of that pool ([M / 2][C]); both backward passes route it to
 * the first maximum of each row pair themselves - bit-identical to ocr_maxpool_bwd followed by ocr_bn_train_bwd. */
int ocr_bn_train_fwd2(const void* x, void* y, const float* gamma, const float* beta, float* save_mean, float* save_rstd, long M, int C,
                      float eps, int relu, void* workspace, const void* residual, int partial_rows, void* pooled, void* stream);
int ocr_bn_train_bwd2(const void* x, const void* y, const void* dy, void* dx, const float* gamma, const float* save_mean,
                      const float* save_rstd, float* dgamma, float* dbeta, long M, int C, int relu, void* workspace, int pooled_dy,
                      int partial_rows, void* stream);
/* partial_rows > 0 (not with pooled_dy): dy is ALREADY ReLU-masked and the workspace holds that many rows [rows][2][C] of (sum dz, sum dz *
 * xhat) from the data-gradient kernel that wrote dy — ocr_conv3x3_dgrad_bnbwd_bf16: dx = (mask_y > 0) ? conv3x3(dy, wdgrad) : 0 plus those
 * per-256-pixel-tile sums with xhat = (z - mean) * rstd, rows = ocr_conv3x3_bnbwd_rows(...) (0: shape not covered, use the separate passes).
 * The batch-norm backward of the producing layer then needs no statistics pass over (z, y, dy) and no second read of y. */
int ocr_conv3x3_bnbwd_rows(int Nb, int W, int H, int Cin, int Cout);
int ocr_conv3x3_dgrad_bnbwd_bf16(const void* dy, const void* wdgrad, void* dx, int Nb, int W, int H, int Cin, int Cout, const void* mask_y,
                                 const void* z, const float* mean, const float* rstd, float* partials, void* stream);
int ocr_colsum_bf16(const void* a, float* out, long M, int C, long lda, void* stream);
int ocr_pack_transpose(const float* in, void* out, int R, int Cc, long ldin, int lstm_units, void* stream);
int ocr_pack_conv_dgrad(const float* w, void* out, int Cin, int Cout, void* stream);
/* all weight re-packs of a step in ONE launch.  jobs: device array of 64-byte records
 * {int type(0 transpose+perm,1 conv-dgrad flip,2 strided cast,3 flat cast,4 LSTM bias [4*lstm_units] fp32 -> packed gate order, fp32 dst);
 *  int R, Cc, lstm_units; long ldin, ldout;
 *  const float* src; bf16* dst; long n; int block_start, nblocks;}  with block_start ascending */
int ocr_pack_jobs(const void* jobs, int njobs, int total_blocks, void* stream);
int ocr_cast_f32_bf16(const float* in, void* out, long n, void* stream);
/* uint8 pixels -> fp32 in [0, 1] (= u8 / 255, correctly rounded: identical to the host's `astype(float32) / 255.`, gen.py:59-65); n % 4 == 0 */
int ocr_u8_to_unit_f32(const void* in, float* out, long n, void* stream);
/* What train.py:130,139 fetches after sess.run, gathered on the device into out[4] (doubles): mean per-sample CTC cost, sum w^2 of
 * the regularised parameters (scalars[1]) and the global gradient norm (scalars[7]) of the optimiser block (scalars: NULL, or the WHOLE block of
 * ocr_optim_scalar_count() doubles — entry [72] is read too),
 * and a bit mask of the error words that read 1 — the persistent LSTM kernels' time-out mark; 0 and 0xFFFFFFFF (a caller-prepared block,
 * OCR_LSTM_PREPARED) both mean "nothing happened" (word_addrs: device array of nwords <= 32 device addresses of int error words); 2^40 is added
 * when the update of the step the report follows was dropped on the device (scalars[72], ocr_optim_step_guarded*). */
int ocr_step_report(const float* costs, int n, const double* scalars, const void* word_addrs, int nwords, double* out, void* stream);
/* One launch binding a DEVICE-resident batch to the engine's fixed input buffers (the feed_dict of train.py:126-130 once the
 * batch is already in HBM): pixels -> x (uint8 / 255 when pixels_are_u8, else an fp32 copy; n_pixels % 4 == 0) and the
 * int32 vectors seq_len, flat labels, labels_len (counts may be 0 for inference). */
int ocr_bind_batch(const void* pixels, int pixels_are_u8, float* x, long n_pixels, const int* seq_len, int* seq_len_dst, int n_seq,
                   const int* labels, int* labels_dst, int n_labels, const int* labels_len, int* labels_len_dst, int n_labels_len,
                   void* stream);
int ocr_cast2d_f32_bf16(const float* in, long ldin, void* out, long ldout, int rows, int cols, void* stream);
int ocr_tnc_to_ntc_bf16(const float* in, void* out, int T, int N, int C, float scale, void* stream);
int ocr_conv5_col2im(const void* col, void* dx, int Nb, int W, int HC, void* stream);

/* ---- layers of the reference DSL outside the shipped graphs (lstm_ctc_ocr_amd/csrc/dsl_ops.hip) --------------- */
/* stand-alone batch_normalization with the stored (moving) statistics — is_training=False, network.py:466-473 */
int ocr_bn_infer_fwd(const void* x, void* y, const float* gamma, const float* beta, const float* mean, const float* var,
                     long M, int C, float eps, int relu, void* stream);
int ocr_bn_infer_bwd(const void* x, const void* y, const void* dy, void* dx, const float* gamma, const float* mean,
                     const float* var, float* dgamma, float* dbeta, long M, int C, float eps, int relu, void* stream);
/* tf.nn.dropout (network.py:626-628): out = keep(i) ? in / keep_prob : 0; keep(i) = hash(seed, *step_counter, i) < keep_prob —
 * the same call with the same arguments on the gradient is the backward pass (nothing stored); step_counter: device double or NULL */
int ocr_dropout_bf16(const void* in, void* out, long n, unsigned seed, const void* step_counter, float keep_prob, void* stream);
/* tf.nn.avg_pool, window == stride in {1,2}^2 (network.py:352-359); backward: src = dy (pooled), dst = dx */
int ocr_avgpool_bf16(const void* src, void* dst, int Nb, int W, int H, int C, int kw, int kh, int backward, void* stream);
/* strided pick y[n,wo,ho,:] = x[n, wo*sw+ow, ho*sh+oh, :] (a strided convolution = stride-1 convolution + pick) / its transpose */
int ocr_subsample_bf16(const void* src, void* dst, int Nb, int W, int H, int C, int Wo, int Ho, int sw, int sh, int ow, int oh,
                       int backward, void* stream);
int ocr_softmax_f32(const float* in, float* out, long rows, int C, void* stream);     /* network.py:441-447, last axis */

/* ---- LSTM (bi_lstm network.py:97-129, lstm network.py:130-152; TF-1.0 LSTMCell gate order i,j,f,o, forget_bias 1.0) -----
 * ndir = 2: forward | backward direction (bidirectional_dynamic_rnn, backward reversed within each length);  ndir = 1: forward only
 * (dynamic_rnn).  Row strides of the per-step tensors follow it: xproj / dz [R][ndir][4U], hout / dhout [R][ndir*U],
 * gates [ndir][R][4U], cell [ndir][R][U]. */
int ocr_lstm_fwd_step(const float* xproj, const void* whT_packed, const int* seq_len, void* hout, float* gates,
                      float* cell, int Nb, int T, int U, int step, float forget_bias, int ndir, void* stream);
int ocr_lstm_bwd_step(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                      const float* gates, const float* cell, void* dz, float* dc_state, int Nb, int T, int U,
                      int step, int ndir, void* stream);
/* whole-sequence (persistent) variants: one launch for all T steps.  The U/16 workgroups of a (direction, batch tile) group
 * exchange h_t / dz_t either through a small ring that stays in one XCD's L2 (protocol 4, default), through the output tensor
 * itself (protocol 2; the call first fills it with the bf16 pattern 0xFFFF = "not written yet") or behind counters (protocol 0) —
 * see ocr_set_lstm_proto.  `sync`: ocr_lstm_seq_sync_words(Nb, U) int32 words of scratch (group counters | ring | tail; what the
 * protocol needs is initialised by the call; LAST word = spin-timeout error flag, non-zero => results invalid).
 * ocr_lstm_seq_supported() tells whether the shape is covered (two directions, U == 256 or 512 — the latter with the contraction
 * axis split over two waves — and a grid of at most one workgroup per CU); otherwise use the step entry points. */
int ocr_lstm_seq_supported(int Nb, int U);
long ocr_lstm_seq_sync_words(int Nb, int U);
int ocr_lstm_seq_debug(void* dbg /* device int64[4*T] phase stamps of workgroup 0, NULL = off */);
/* test hook: in the persistent launches that follow, unit block 0 of EVERY (direction, batch tile) group sleeps units x 64 shader clocks at the
 * top of iteration `at` (at < 0: of every iteration; units 0 = off, the default; 0 .. 4096) — the skew between the workgroups of a group that HBM
 * contention produces once in ~10^4 launches, made deterministic: the ring hand-off that stands in for the dependent op sequence of
 * network.py:104-109 must be right for any relative speed of its workgroups (tests/test_gpu_stress.py, tools/lstm_ring_model.py).  Baked into a
 * captured graph at capture time. */
int ocr_lstm_seq_test_skew(int units, int at);
int ocr_lstm_fwd_seq(const float* xproj, const void* whT_packed, const int* seq_len, void* hout, float* gates,
                     float* cell, int Nb, int T, int U, float forget_bias, void* sync, void* stream);
int ocr_lstm_bwd_seq(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                     const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync, void* stream);
/* The same launches with flags.  OCR_LSTM_PREPARED: the caller has set EVERY word of `sync` to 0xFFFFFFFF earlier on this stream (the
 * training engine does that for all of a step's launches inside a kernel it runs anyway, ocr_conv1_pool_fwd_train) and the call's own
 * fill launch is skipped — under the ring protocol (4) only; under the counter protocol the call prepares the block itself as before.
 * The error word (last word of `sync`) of a caller-prepared block reads 0xFFFFFFFF when nothing happened and 1 after a time-out. */
#define OCR_LSTM_PREPARED 1
int ocr_lstm_fwd_seq2(const float* xproj, const void* whT_packed, const int* seq_len, void* hout, float* gates,
                      float* cell, int Nb, int T, int U, float forget_bias, void* sync, int flags, void* stream);
int ocr_lstm_bwd_seq2(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                      const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync, int flags, void* stream);
/* The forward recurrence WITH the input projection inside (no projection GEMM, no fp32 projection tensor): x bf16 [Nb * T][D] (the layer's
 * input rows, batch-major), wxT_packed bf16 [2][4U][D] and bias_packed fp32 [2][4U] exactly as the projection GEMM takes them
 * (ocr_pack_transpose with lstm_units, ocr_lstm_pack_bias).  Covered: ring protocol, 16-row tiles, four-wave workgroups, D = 512 / 1024,
 * U = 256 / 512 — ocr_lstm_fwd_seq_x_supported; otherwise OCR_ERR_INVALID (run the GEMM + ocr_lstm_fwd_seq2).  flags as above. */
int ocr_lstm_fwd_seq_x_supported(int Nb, int U, int D);
int ocr_lstm_fwd_seq_x(const void* x, const void* wxT_packed, const float* bias_packed, int D, const void* whT_packed,
                       const int* seq_len, void* hout, float* gates, float* cell, int Nb, int T, int U, float forget_bias,
                       void* sync, int flags, void* stream);
/* hand-off protocol of the persistent kernels: 4 (default) = data-as-flag through a ring inside one XCD's L2, 2 = data-as-flag
 * through the output tensor inside one XCD's L2 — both need workgroups with equal (id & 7) on one XCD, see ocr_probe_xcc;
 * 0 = counters (sc1): placement independent */
int ocr_set_lstm_proto(int proto);
/* waves per workgroup of the persistent kernels' 16-row tiles under protocol 4: 4 (default: the contraction axis, the polls and the gate
 * math of a step split over four waves) or 1 (the one-wave kernels of rounds 2-3).  Same tensors, same hand-off ring; results equal up to
 * fp32 summation order.  OCR_ERR_INVALID for other values.  The environment variable OCR_LSTM_KSPLIT (1 / 4) wins over this setter. */
int ocr_set_lstm_ksplit(int waves);
int ocr_lstm_hprev(const void* hout, const int* seq_len, void* hprev, int Nb, int T, int U, int ndir, void* stream);
/* xh [ndir][Nb*T][D+U] = [x | h_{t-1} in direction order]: operand of the LSTMCell-matrix weight gradient (one GEMM per direction) */
int ocr_lstm_xh(const void* x, const void* hout, const int* seq_len, void* xh, int Nb, int T, int D, int U, int ndir, void* stream);
int ocr_lstm_pack_bias(const float* b_fw, const float* b_bw /* NULL when ndir == 1 */, float* out, int U, int ndir, void* stream);

/* ---- optimiser (train.py:73-85: clip_by_global_norm 10.0 + Adam / Momentum / RMSProp; L2 of network.py:630-637) */
int ocr_optim_scalar_count(void);   /* doubles in the caller-owned `scalars` block: 8 of state + per-step partial-sum bins + 2 of the guard */
int ocr_optim_init(void* scalars /* ocr_optim_scalar_count() doubles */, double lr, void* stream);
int ocr_optim_set_lr(void* scalars, double lr, int multiply, void* stream);
int ocr_optim_step(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                   float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                   void* scalars, void* stream);
/* the same step behind a guard: guard_addrs = device array of nguard (<= 64) device addresses of int words that read 1 when a kernel of this step
 * reported invalid results (the persistent LSTM launches' error words: a bounded inter-workgroup wait expired).  The update is then dropped on the
 * device — moments, parameters and the bias-correction powers stay; scalars[72] = 1 for that step, scalars[73] counts the dropped steps. */
int ocr_optim_step_guarded(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                           float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                           void* scalars, const void* guard_addrs, int nguard, void* stream);
/* ... and / or behind a drop flag (data parallel, train.py:79-83 applied by every replica or by none): drop_flag = a device float, > 0 => drop
 * (NULL: none).  ocr_guard_flag writes 1.0f / 0.0f (any of the nguard words reads 1 / none does) into flag_out; a rank puts that word at the end of
 * its late gradient bucket, the SUM all-reduce turns it into the number of ranks whose step timed out, and every rank drops the SAME step.
 * scalars[74] counts the expired waits the guard has seen (error words that read 1 + ranks that raised the flag). */
int ocr_optim_step_guarded2(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                            float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                            void* scalars, const void* guard_addrs, int nguard, const float* drop_flag, void* stream);
int ocr_guard_flag(const void* guard_addrs, int nguard, float* flag_out, void* stream);

/* ---- GPU-side captcha synthesis (the data side of the path: /root/reference/lib/lstm/utils/gen.py:31-37 generateImg — captcha.ImageCaptcha on 12
 * worker processes — and :41-67 groupBatch: resize to height 32, [W, 32] rows, right-padded with 0) ---------------------------------------------
 * One workgroup per image composes resident glyph masks in LDS with Pillow's own arithmetic (affine bilinear rotation, DIV255 paste, bicubic
 * resize, noise dots + arc, 3 x 3 SMOOTH, bilinear resize) and writes out[n_images][W][out_h] uint8 — what ocr_bind_batch takes as pixels.
 * params: int32 [n_images][words_per_image] records (lstm_ctc_ocr_amd/utils/synth.draw_params: every random draw + the integer geometry;
 * max_glyphs glyph slots per record); atlas: concatenated 8-bit glyph masks (records hold byte offsets into it); stamp: n_stamp (dx, dy) int32
 * pairs = the footprint of one noise dot; canvas_cap / width_cap >= every record's canvas_w / width (they size the LDS image:
 * 60 * (canvas_cap + width_cap) + 34816 bytes <= 160 KB, else OCR_ERR_INVALID).  The records live in device memory: that they respect the two
 * caps, the atlas' extent and max_glyphs is the caller's contract and is NOT checked by the launch (utils/synth.draw_params produces records and
 * caps together). */
int ocr_captcha_synth(const int* params, int n_images, int words_per_image, int max_glyphs, const void* atlas, const int* stamp, int n_stamp,
                      void* out, int W, int out_h, int canvas_cap, int width_cap, void* stream);

/* diagnostics: s_memtime stamps of workgroup 0 (NULL = off); device int64 [8 waves][64 steps][8] / [8][80][8] */
int ocr_wgrad9_debug(void* dbg);

/* ---- device probes used by the test-suite (not part of the hot path) ------------------------------------------ */
/* ds_read_b64_tr_b16 lane-semantics probe: LDS holds shorts 0..8191 (value = element index); lane l reads at
 * byte offset addr[l]; out[l*4+j] = element j returned to lane l. */
int ocr_probe_tr16(const int* addr /* 64 */, int* out /* 64*4 */, void* stream);
/* out[id] = XCC_ID of workgroup id of a 1-D grid, out[nblocks + id] = its HW_ID register (evidence for id & 7 == XCD) */
int ocr_probe_xcc(int* out /* 2*nblocks */, int nblocks, int threads, void* stream);
/* one-GPU stand-in for the CUs a concurrent RCCL collective holds (torch.distributed all_reduce over xGMI: lib/lstm/train.py has no
 * counterpart — the reference is single-device): nblocks workgroups of `threads` threads with lds_bytes of LDS each stay
 * resident for `us` microseconds without issuing work.  Used by the OCR_FAKE_COMM_CUS / OCR_FAKE_COMM_US emulation (lstm_ctc_ocr_amd/dist.py). */
int ocr_occupy_cus(int nblocks, int threads, int lds_bytes, float us, void* stream);
/* counter calibration: every wave issues iters x 8 back-to-back v_mfma_f32_16x16x32_bf16 (a saturated matrix pipe with an exactly known
 * instruction count); clk[4] = {shader clock, 100 MHz clock} of workgroup 0 at loop entry and exit, or NULL (tools/mfma_busy_probe.py) */
int ocr_mfma_busy_probe(float* out, int nblocks, int threads /* multiple of 64, <= 512 */, int iters, long long* clk, void* stream);

#ifdef __cplusplus
}
#endif
#endif
