// Fused optimiser step for gfx950.
//
// Reference: lib/lstm/train.py:73-85 — tf.gradients -> tf.clip_by_global_norm(g, 10.0) -> Adam /
// RMSProp / Momentum apply_gradients, with the L2 regulariser of lib/networks/network.py:630-637,660-662
// (wd * ||w||^2 / 2 on conv + FC weights) folded in as g += wd * w.
//
// All parameters live in ONE flat fp32 buffer in which the L2-regularised tensors form one contiguous range, so the whole
// step is two HBM-bound grid-stride kernels over that buffer (+ a 1-thread "tick" that advances the
// Adam bias correction on the device, which keeps the step replayable from a hipGraph with no host
// scalars baked in).  The same flat gradient buffer is what the data-parallel all-reduce sees.
#include "common.h"
#include <math.h>

// device scalar block layout (doubles): [0] sum g^2, [1] sum w^2 over regularised range,
// [2] lr, [3] lr_t (bias-corrected), [4] beta1^t, [5] beta2^t, [6] step count, [7] last global norm
#define SC_NORM2 0
#define SC_REG2 1
#define SC_LR 2
#define SC_LRT 3
#define SC_B1T 4
#define SC_B2T 5
#define SC_STEP 6
#define SC_GNORM 7
// [8 .. 8+32) partial sums of g^2, [40 .. 40+32) partial sums of w^2: the norm pass adds each block's total into bin
// (block & 31) — 32 addresses in different cache lines take the atomics in parallel, where ONE address serialised them
// (~12 ns each: 384 blocks x 2 atomics were a 9 us tail on a 12 us pass) — and the update kernels add the bins up.
#define SC_BINS 32
#define SC_NORM_BINS 8
#define SC_REG_BINS 40
// [72] 1 while the running step's update is dropped, [73] number of dropped updates so far (ocr_optim_step_guarded)
// [74] number of expired hand-off waits the guard has seen so far (error words that read 1; with a drop flag: ranks that raised it)
#define SC_SKIP 72
#define SC_SKIPPED 73
#define SC_TIMEOUTS 74
#define SC_TOTAL 76
// Round 4, measured and reverted (profiles/r04c_kernel_stats.md): folding the tick into the norm pass — last-arriving block by two-level
// arrival tickets, __threadfence() between a block's bin atomics and its ticket — made optim_prep_kernel 62.8 us instead of 16.9 (+ 4.9 for
// the tick launch it saved): a device-scope release fence on this chip writes the XCD's L2 back (the pass has just stored 22 MB of
// gradients), once per block.  The tick stays a launch of its own.

// guard (round 5): device addresses of int words that read 1 when a kernel of this step reported that its results are invalid — the
// persistent LSTM launches' error words (a bounded inter-workgroup wait expired: lstm_seq.hip).  The step is then DROPPED on the device:
// no moment, no parameter and no bias-correction update (the host learns of it from the step report, one step later, and logs it) —
// before, the garbage gradient was applied and the run died on the report.
// drop_flag (round 6, data parallel): a float that is > 0 when ANY rank's step reported such a time-out — every rank writes 1.0 / 0.0 into a
// word that rides at the end of the late gradient bucket (ocr_guard_flag), the SUM all-reduce makes it the number of ranks that raised it, and
// every rank then drops the SAME step: the replicas stay bit-identical (before: the guard was single-GPU only; with several ranks the garbage
// gradient was all-reduced and applied everywhere, and the job died on the report one step later).
__global__ void optim_tick_kernel(double* sc, double beta1, double beta2, const long long* __restrict__ guard, int nguard,
                                  const float* __restrict__ drop_flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int nbad = 0;
        for (int i = 0; i < nguard; ++i) nbad += *(const int*)guard[i] == 1;
        if (drop_flag) { const float f = *drop_flag; if (f > 0.f) nbad += (int)(f + 0.5f); }
        const bool bad = nbad > 0;
        sc[SC_SKIP] = bad ? 1.0 : 0.0;
        if (bad) { sc[SC_SKIPPED] += 1.0; sc[SC_TIMEOUTS] += (double)nbad; }
        else {
            double b1t = sc[SC_B1T] * beta1, b2t = sc[SC_B2T] * beta2;
            sc[SC_B1T] = b1t; sc[SC_B2T] = b2t;
            sc[SC_STEP] += 1.0;
            sc[SC_LRT] = sc[SC_LR] * sqrt(1.0 - b2t) / (1.0 - b1t);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * SC_BINS) sc[SC_NORM_BINS + threadIdx.x] = 0.0;
}
// out[0] = 1.0f if any of the nguard int words reads 1, else 0.0f (written EVERY step: the word needs no clearing)
__global__ void guard_flag_kernel(const long long* __restrict__ guard, int nguard, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        bool bad = false;
        for (int i = 0; i < nguard; ++i) bad = bad || *(const int*)guard[i] == 1;
        out[0] = bad ? 1.0f : 0.0f;
    }
}
// totals of the two bin arrays, computed by every block of an update kernel (64 cached loads); block 0 also publishes them
__device__ __forceinline__ float optim_gnorm(double* sc) {
    __shared__ double tot[2];
    if (threadIdx.x < 64) {
        double a = threadIdx.x < SC_BINS ? sc[SC_NORM_BINS + threadIdx.x] : 0.0;
        double b = threadIdx.x < SC_BINS ? sc[SC_REG_BINS + threadIdx.x] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
        if (threadIdx.x == 0) { tot[0] = a; tot[1] = b; }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[SC_NORM2] = tot[0]; sc[SC_REG2] = tot[1]; sc[SC_GNORM] = sqrt(tot[0]); }
    return (float)sqrt(tot[0]);
}

// pass 1: sum (g + wd * w)^2 (wd on the regularised range [reg0, reg1) only) and sum w^2 over that range
__global__ __launch_bounds__(256) void optim_prep_kernel(const float* __restrict__ p, const float* __restrict__ g, long n,
                                                         long reg0, long reg1, float wd, double* sc) {
    float s2 = 0.f, r2 = 0.f;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    // (round 4: g + wd * w is not stored any more — the update kernel forms it again from g and w, which it reads anyway: 22 MB of stores less
    //  per step; the flat gradient buffer keeps the loss term alone.  Four independent 16-byte load pairs per iteration: the pass is a pure
    //  read stream now and was latency-bound with one pair in flight per thread.)
    for (; i + 3 * stride < n; i += 4 * stride) {   // n, reg0, reg1 % 4 == 0
        f32x4 gv[4], pv[4];
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long ik = i + k * stride;
            in[k] = ik >= reg0 && ik < reg1;
            gv[k] = *(const f32x4*)(g + ik);
            pv[k] = in[k] ? *(const f32x4*)(p + ik) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (in[k]) {
                gv[k] = gv[k] + pv[k] * wd;
                r2 += pv[k].x * pv[k].x + pv[k].y * pv[k].y + pv[k].z * pv[k].z + pv[k].w * pv[k].w;
            }
            s2 += gv[k].x * gv[k].x + gv[k].y * gv[k].y + gv[k].z * gv[k].z + gv[k].w * gv[k].w;
        }
    }
    for (; i < n; i += stride) {
        f32x4 gv = *(const f32x4*)(g + i);
        if (i >= reg0 && i < reg1) {
            f32x4 pv = *(const f32x4*)(p + i);
            gv = gv + pv * wd;
            r2 += pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w;
        }
        s2 += gv.x * gv.x + gv.y * gv.y + gv.z * gv.z + gv.w * gv.w;
    }
    s2 = wave_sum(s2); r2 = wave_sum(r2);
    __shared__ float red[2][4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s2; red[1][wave] = r2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int bin = blockIdx.x & (SC_BINS - 1);
        atomicAdd(&sc[SC_NORM_BINS + bin], (double)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        atomicAdd(&sc[SC_REG_BINS + bin], (double)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
}

// pass 2, Adam (TF form): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void adam_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long n,
                                                          float beta1, float beta2, float eps, float clip,
                                                          double* sc, long reg0, long reg1, float wd) {
    const float gnorm = optim_gnorm(sc);
    if (sc[SC_SKIP] != 0.0) return;                          // dropped step (uniform over the grid: written by the tick launch)
    const float scale = (clip > 0.f) ? clip / fmaxf(gnorm, clip) : 1.f;
    const float lrt = (float)sc[SC_LRT];
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        f32x4 mv = *(const f32x4*)(m + i), vv = *(const f32x4*)(v + i), pv = *(const f32x4*)(p + i);
        f32x4 gv = *(const f32x4*)(g + i);
        if (i >= reg0 && i < reg1) gv = gv + pv * wd;          // the L2 term, exactly as the norm pass formed it
        gv = gv * scale;
        mv = mv * beta1 + gv * (1.f - beta1);
        vv = vv * beta2 + gv * gv * (1.f - beta2);
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] -= lrt * mv[r] / (sqrtf(vv[r]) + eps);
        *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv; *(f32x4*)(p + i) = pv;
    }
}
// Momentum (TF MomentumOptimizer): acc = mom*acc + g; w -= lr*acc
__global__ __launch_bounds__(256) void momentum_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, long n, float mom, float clip,
                                                              double* sc, long reg0, long reg1, float wd) {
    const float gnorm = optim_gnorm(sc);
    if (sc[SC_SKIP] != 0.0) return;                          // dropped step (uniform over the grid: written by the tick launch)
    const float scale = (clip > 0.f) ? clip / fmaxf(gnorm, clip) : 1.f;
    const float lr = (float)sc[SC_LR];
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        f32x4 pv = *(const f32x4*)(p + i);
        f32x4 gv = *(const f32x4*)(g + i);
        if (i >= reg0 && i < reg1) gv = gv + pv * wd;
        gv = gv * scale;
        f32x4 mv = *(const f32x4*)(m + i) * mom + gv;
        pv = pv - mv * lr;
        *(f32x4*)(m + i) = mv; *(f32x4*)(p + i) = pv;
    }
}
// RMSProp (TF RMSPropOptimizer defaults: decay .9, momentum 0, eps 1e-10): ms = d ms + (1-d) g^2; w -= lr g / sqrt(ms + eps)
__global__ __launch_bounds__(256) void rmsprop_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ ms, long n, float decay, float eps,
                                                             float clip, double* sc, long reg0, long reg1, float wd) {
    const float gnorm = optim_gnorm(sc);
    if (sc[SC_SKIP] != 0.0) return;                          // dropped step (uniform over the grid: written by the tick launch)
    const float scale = (clip > 0.f) ? clip / fmaxf(gnorm, clip) : 1.f;
    const float lr = (float)sc[SC_LR];
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        f32x4 pv = *(const f32x4*)(p + i);
        f32x4 gv = *(const f32x4*)(g + i);
        if (i >= reg0 && i < reg1) gv = gv + pv * wd;
        gv = gv * scale;
        f32x4 sv = *(const f32x4*)(ms + i) * decay + gv * gv * (1.f - decay);
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] -= lr * gv[r] / sqrtf(sv[r] + eps);
        *(f32x4*)(ms + i) = sv; *(f32x4*)(p + i) = pv;
    }
}
__global__ void optim_init_kernel(double* sc, double lr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        sc[SC_NORM2] = 0; sc[SC_REG2] = 0; sc[SC_LR] = lr; sc[SC_LRT] = lr; sc[SC_B1T] = 1.0; sc[SC_B2T] = 1.0;
        sc[SC_STEP] = 0; sc[SC_GNORM] = 0; sc[SC_SKIP] = 0; sc[SC_SKIPPED] = 0; sc[SC_TIMEOUTS] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * SC_BINS) sc[SC_NORM_BINS + threadIdx.x] = 0.0;
}
__global__ void optim_set_lr_kernel(double* sc, double lr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sc[SC_LR] = lr;
}
__global__ void optim_scale_lr_kernel(double* sc, double gamma) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sc[SC_LR] *= gamma;
}

// ------------------------------------------------------------------------------------------
// C ABI.  `scalars` is a caller-owned device block of ocr_optim_scalar_count() doubles (layout above).
// ------------------------------------------------------------------------------------------
extern "C" int ocr_optim_scalar_count(void) { return SC_TOTAL; }
extern "C" int ocr_optim_init(void* scalars, double lr, void* stream) {
    if (!scalars) return OCR_ERR_INVALID;
    optim_init_kernel<<<1, 64, 0, (hipStream_t)stream>>>((double*)scalars, lr);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_optim_set_lr(void* scalars, double lr, int multiply, void* stream) {
    if (!scalars) return OCR_ERR_INVALID;
    if (multiply) optim_scale_lr_kernel<<<1, 64, 0, (hipStream_t)stream>>>((double*)scalars, lr);
    else optim_set_lr_kernel<<<1, 64, 0, (hipStream_t)stream>>>((double*)scalars, lr);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
// solver: 0 Adam (beta1, beta2, eps), 1 Momentum (beta1 = momentum), 2 RMSProp (beta1 = decay, eps)
// state1/state2: Adam m, v ; Momentum accumulator (state2 unused) ; RMSProp mean-square (state2 unused)
extern "C" int ocr_optim_step_guarded2(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                                       float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                                       void* scalars, const void* guard_addrs, int nguard, const float* drop_flag, void* stream_);
extern "C" int ocr_optim_step(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                              float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                              void* scalars, void* stream_) {
    return ocr_optim_step_guarded2(params, grads, state1, state2, n, reg_begin, reg_end, weight_decay, clip_norm, solver, beta1, beta2, eps, scalars,
                                   nullptr, 0, nullptr, stream_);
}
// ... with a guard: guard_addrs = device array of nguard device addresses of int words; the update is dropped when any of them reads 1
extern "C" int ocr_optim_step_guarded(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                                      float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                                      void* scalars, const void* guard_addrs, int nguard, void* stream_) {
    return ocr_optim_step_guarded2(params, grads, state1, state2, n, reg_begin, reg_end, weight_decay, clip_norm, solver, beta1, beta2, eps, scalars,
                                   guard_addrs, nguard, nullptr, stream_);
}
// ... and / or with a drop flag: a device float, > 0 => drop (data parallel: the all-reduced ocr_guard_flag word)
extern "C" int ocr_optim_step_guarded2(float* params, float* grads, float* state1, float* state2, long n, long reg_begin, long reg_end,
                                       float weight_decay, float clip_norm, int solver, float beta1, float beta2, float eps,
                                       void* scalars, const void* guard_addrs, int nguard, const float* drop_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nguard < 0 || nguard > 64 || (nguard && !guard_addrs)) return OCR_ERR_INVALID;
    if (!params || !grads || !state1 || !scalars || n <= 0 || (n & 3) || (reg_begin & 3) || (reg_end & 3) || reg_begin < 0 ||
        reg_end < reg_begin || reg_end > n)
        return OCR_ERR_INVALID;
    if (solver == 0 && !state2) return OCR_ERR_INVALID;
    double* sc = (double*)scalars;
    optim_tick_kernel<<<1, 64, 0, stream>>>(sc, (double)beta1, (double)beta2, (const long long*)guard_addrs, nguard, drop_flag);
    OCR_CHECK_LAUNCH();
    int blocks = (int)((n / 4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    int pblocks = blocks > 1024 ? 1024 : blocks;       // the atomics at the end go to 32 bins, see SC_BINS
    optim_prep_kernel<<<pblocks, 256, 0, stream>>>(params, grads, n, reg_begin, weight_decay > 0.f ? reg_end : reg_begin, weight_decay, sc);
    OCR_CHECK_LAUNCH();
    const long r0 = reg_begin, r1 = weight_decay > 0.f ? reg_end : reg_begin;
    if (solver == 0) adam_update_kernel<<<blocks, 256, 0, stream>>>(params, grads, state1, state2, n, beta1, beta2, eps, clip_norm, sc, r0, r1, weight_decay);
    else if (solver == 1) momentum_update_kernel<<<blocks, 256, 0, stream>>>(params, grads, state1, n, beta1, clip_norm, sc, r0, r1, weight_decay);
    else if (solver == 2) rmsprop_update_kernel<<<blocks, 256, 0, stream>>>(params, grads, state1, n, beta1, eps, clip_norm, sc, r0, r1, weight_decay);
    else return OCR_ERR_INVALID;
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
// flag_out[0] := 1.0f if any of the nguard int words (device addresses in guard_addrs) reads 1, else 0.0f — the word a data-parallel rank appends to
// its late gradient bucket so that the SUM all-reduce tells every rank whether ANY rank's persistent LSTM launch timed out in this step
extern "C" int ocr_guard_flag(const void* guard_addrs, int nguard, float* flag_out, void* stream) {
    if (nguard < 0 || nguard > 64 || (nguard && !guard_addrs) || !flag_out) return OCR_ERR_INVALID;
    guard_flag_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const long long*)guard_addrs, nguard, flag_out);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
