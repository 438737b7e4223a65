// Layers of the reference DSL that the shipped graphs do not use but deeper configurations (BASELINE configs[4])
// are written with: stand-alone batch_normalization in inference mode, dropout, avg_pool, strided convolution (as a stride-1
// convolution + strided pick), softmax.  All HBM-bound element-wise / small-reduction kernels: 16 bytes per lane, grid-stride.
//   batch_normalization   lib/networks/network.py:466-473  (is_training=False: moving statistics, never updated in the reference)
//   dropout               lib/networks/network.py:626-628  (tf.nn.dropout: keep with probability p, scale kept values by 1/p)
//   avg_pool              lib/networks/network.py:352-359
//   conv (strides)        lib/networks/network.py:193-216  (tf.nn.conv2d strides [1, s_h, s_w, 1])
//   softmax               lib/networks/network.py:441-447
#include "common.h"

static inline int dsl_grid(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
__device__ __forceinline__ void dsl_unpack8(u32x4 v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ u32x4 dsl_pack8(const float* o) {
    return (u32x4){pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
}

// ------------------------------------------------------------------------------------------------ inference-mode batch norm
// y = gamma * (x - mean) * rsqrt(var + eps) + beta [+ ReLU]   with the STORED (moving) statistics
__global__ __launch_bounds__(256) void bn_infer_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ var, long M, int C, float eps, int relu) {
    const int groups = C >> 3;
    const long total = M * groups;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % groups) * 8;
        float v[8], o[8];
        dsl_unpack8(*(const u32x4*)(x + i * 8), v);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float sc = gamma[c0 + c] * rsqrtf(var[c0 + c] + eps);
            o[c] = (v[c] - mean[c0 + c]) * sc + beta[c0 + c];
            if (relu) o[c] = fmaxf(o[c], 0.f);
        }
        *(u32x4*)(y + i * 8) = dsl_pack8(o);
    }
}
// dx = dy' * gamma * rsqrt(var + eps) (dy' = dy where y > 0 if relu), dgamma += sum dy' * xhat, dbeta += sum dy'
// one workgroup = 256 rows x 64-channel slab, per-channel partial sums in LDS, one fp32 atomic per channel and workgroup
__global__ __launch_bounds__(256) void bn_infer_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ dx, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ var, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           long M, int C, float eps, int relu) {
    __shared__ float sg[32][64], sb[32][64];
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;        // 8 channel groups of 8 = 64 channels, 32 row lanes
    const int cslabs = (C + 63) / 64;
    const long rblocks = (M + 255) / 256;
    for (long blk = blockIdx.x; blk < rblocks * cslabs; blk += gridDim.x) {
        const int c0 = (int)(blk % cslabs) * 64 + cg * 8;
        const long r0 = (blk / cslabs) * 256;
        float ag[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ab[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c0 < C) {
            float sc[8], mu[8], rs[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { rs[c] = rsqrtf(var[c0 + c] + eps); sc[c] = gamma[c0 + c] * rs[c]; mu[c] = mean[c0 + c]; }
            for (int k = 0; k < 8; ++k) {
                const long r = r0 + rl + 32 * k;
                if (r >= M) break;
                float xv[8], gv[8], yv[8], o[8];
                dsl_unpack8(*(const u32x4*)(x + r * C + c0), xv);
                dsl_unpack8(*(const u32x4*)(dy + r * C + c0), gv);
                if (relu) dsl_unpack8(*(const u32x4*)(y + r * C + c0), yv);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float g = (relu && !(yv[c] > 0.f)) ? 0.f : gv[c];
                    o[c] = g * sc[c];
                    ag[c] += g * (xv[c] - mu[c]) * rs[c];
                    ab[c] += g;
                }
                *(u32x4*)(dx + r * C + c0) = dsl_pack8(o);
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) { sg[rl][cg * 8 + c] = ag[c]; sb[rl][cg * 8 + c] = ab[c]; }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int c = (int)(blk % cslabs) * 64 + threadIdx.x;
            if (c < C) {
                float tg = 0.f, tb = 0.f;
                for (int k = 0; k < 32; ++k) { tg += sg[k][threadIdx.x]; tb += sb[k][threadIdx.x]; }
                atomicAdd(dgamma + c, tg);
                atomicAdd(dbeta + c, tb);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ dropout
// keep(i) is a pure function of (seed, step counter read from device memory, element index): the backward pass and the CPU
// oracle (oracle/plan_exec.py) regenerate the same mask, nothing is stored.  32-bit mix (murmur3 finaliser), keep iff
// hash < keep_prob * 2^32.
__device__ __forceinline__ uint32_t dsl_mix(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long n8, uint32_t seed,
                                                      const double* __restrict__ step_counter, float keep_prob) {
    const uint32_t step = step_counter ? (uint32_t)(long)(*step_counter) : 0u;
    const uint32_t thr = keep_prob >= 1.f ? 0xffffffffu : (uint32_t)((double)keep_prob * 4294967296.0);
    const float inv = 1.0f / keep_prob;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8], o[8];
        dsl_unpack8(*(const u32x4*)(in + i * 8), v);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t h = dsl_mix(dsl_mix((uint32_t)(i * 8 + c) ^ seed) + step * 0x9e3779b9u);
            o[c] = (keep_prob >= 1.f || h < thr) ? v[c] * inv : 0.f;
        }
        *(u32x4*)(out + i * 8) = dsl_pack8(o);
    }
}

// ------------------------------------------------------------------------------------------------ average pool (window == stride)
template <int KW, int KH, bool BWD>
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int Nb, int W, int H, int C) {
    const int Wo = W / KW, Ho = H / KH, groups = C >> 3;
    const long total = (long)Nb * Wo * Ho * groups;
    const float inv = 1.0f / (KW * KH);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int gq = (int)(idx % groups);
        const long op = idx / groups;
        const int ho = (int)(op % Ho);
        const long q = op / Ho;
        const int wo = (int)(q % Wo);
        const long n = q / Wo;
        const long big = (((n * W + (long)wo * KW) * H) + (long)ho * KH) * C + gq * 8;      // window origin in the un-pooled map
        if (!BWD) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
#pragma unroll
            for (int a = 0; a < KW; ++a)
#pragma unroll
                for (int b = 0; b < KH; ++b) {
                    dsl_unpack8(*(const u32x4*)(src + big + ((long)a * H + b) * C), v);
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] += v[c];
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] *= inv;
            *(u32x4*)(dst + op * C + gq * 8) = dsl_pack8(acc);
        } else {                                              // src = dy (pooled), dst = dx (un-pooled): every window cell gets dy / (KW KH)
            float v[8];
            dsl_unpack8(*(const u32x4*)(src + op * C + gq * 8), v);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] *= inv;
            const u32x4 pk = dsl_pack8(v);
#pragma unroll
            for (int a = 0; a < KW; ++a)
#pragma unroll
                for (int b = 0; b < KH; ++b) *(u32x4*)(dst + big + ((long)a * H + b) * C) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------ strided pick / its transpose
// fwd: y[n, wo, ho, :] = x[n, wo*sw + ow, ho*sh + oh, :]      bwd (SCATTER): dx = 0 everywhere except the picked positions
template <bool BWD>
__global__ __launch_bounds__(256) void subsample_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int Nb, int W, int H, int C,
                                                        int Wo, int Ho, int sw, int sh, int ow, int oh) {
    const int groups = C >> 3;
    if (!BWD) {
        const long total = (long)Nb * Wo * Ho * groups;
        for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
            const int gq = (int)(idx % groups);
            const long op = idx / groups;
            const int ho = (int)(op % Ho);
            const long q = op / Ho;
            const int wo = (int)(q % Wo);
            const long n = q / Wo;
            *(u32x4*)(dst + op * C + gq * 8) = *(const u32x4*)(src + (((n * W + (long)wo * sw + ow) * H) + (long)ho * sh + oh) * C + gq * 8);
        }
    } else {
        const long total = (long)Nb * W * H * groups;
        for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
            const int gq = (int)(idx % groups);
            const long ip = idx / groups;
            const int h = (int)(ip % H);
            const long q = ip / H;
            const int w = (int)(q % W);
            const long n = q / W;
            u32x4 v = {0, 0, 0, 0};
            const int dw = w - ow, dh = h - oh;
            if (dw >= 0 && dh >= 0 && dw % sw == 0 && dh % sh == 0 && dw / sw < Wo && dh / sh < Ho)
                v = *(const u32x4*)(src + (((n * Wo + dw / sw) * Ho) + dh / sh) * C + gq * 8);
            *(u32x4*)(dst + ip * C + gq * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ softmax over the last axis (fp32 rows)
__global__ __launch_bounds__(64) void softmax_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int C) {
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* p = in + r * C;
        float m = -INFINITY;
        for (int c = threadIdx.x; c < C; c += 64) m = fmaxf(m, p[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 64) s += __expf(p[c] - m);
        s = wave_sum(s);
        const float inv = 1.0f / s;
        for (int c = threadIdx.x; c < C; c += 64) out[r * C + c] = __expf(p[c] - m) * inv;
    }
}

// ================================================================================================ C ABI
extern "C" int ocr_bn_infer_fwd(const void* x, void* y, const float* gamma, const float* beta, const float* mean, const float* var,
                                long M, int C, float eps, int relu, void* stream) {
    if (!x || !y || !gamma || !beta || !mean || !var || M <= 0 || C <= 0 || (C & 7)) return OCR_ERR_INVALID;
    bn_infer_fwd_kernel<<<dsl_grid(M * (C >> 3)), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, gamma, beta, mean, var, M, C, eps, relu);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_bn_infer_bwd(const void* x, const void* y, const void* dy, void* dx, const float* gamma, const float* mean,
                                const float* var, float* dgamma, float* dbeta, long M, int C, float eps, int relu, void* stream) {
    if (!x || !dy || !dx || !gamma || !mean || !var || !dgamma || !dbeta || (relu && !y) || M <= 0 || C <= 0 || (C & 7)) return OCR_ERR_INVALID;
    const long blocks = ((M + 255) / 256) * ((C + 63) / 64);
    bn_infer_bwd_kernel<<<(int)(blocks > 2048 ? 2048 : blocks), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, gamma, mean, var, dgamma, dbeta, M, C, eps, relu);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
// out = keep(i) ? in / keep_prob : 0 — forward on activations, backward on gradients (same mask).  step_counter: device
// double incremented once per training step by the optimiser (scalars[6], ocr_optim_step) or NULL (= step 0).
extern "C" int ocr_dropout_bf16(const void* in, void* out, long n, unsigned seed, const void* step_counter, float keep_prob, void* stream) {
    if (!in || !out || n <= 0 || (n & 7) || !(keep_prob > 0.f) || keep_prob > 1.f) return OCR_ERR_INVALID;
    dropout_kernel<<<dsl_grid(n / 8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, n / 8, seed, (const double*)step_counter, keep_prob);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_avgpool_bf16(const void* src, void* dst, int Nb, int W, int H, int C, int kw, int kh, int backward, void* stream) {
    if (!src || !dst || (C & 7) || kw < 1 || kw > 2 || kh < 1 || kh > 2 || W % kw || H % kh) return OCR_ERR_INVALID;
    const long total = (long)Nb * (W / kw) * (H / kh) * (C >> 3);
    const bf16_t* s = (const bf16_t*)src; bf16_t* d = (bf16_t*)dst; hipStream_t st = (hipStream_t)stream;
    const int gr = dsl_grid(total, 8192);
#define AVG(KW_, KH_) do { if (backward) avgpool_kernel<KW_, KH_, true><<<gr, 256, 0, st>>>(s, d, Nb, W, H, C); \
                           else avgpool_kernel<KW_, KH_, false><<<gr, 256, 0, st>>>(s, d, Nb, W, H, C); } while (0)
    if (kw == 2 && kh == 2) AVG(2, 2); else if (kw == 1 && kh == 2) AVG(1, 2); else if (kw == 2 && kh == 1) AVG(2, 1); else AVG(1, 1);
#undef AVG
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
// forward: dst [Nb, Wo, Ho, C] = src [Nb, W, H, C] at (wo*sw + ow, ho*sh + oh); backward: dst [Nb, W, H, C] = scatter of src [Nb, Wo, Ho, C]
extern "C" int ocr_subsample_bf16(const void* src, void* dst, int Nb, int W, int H, int C, int Wo, int Ho, int sw, int sh, int ow, int oh,
                                  int backward, void* stream) {
    if (!src || !dst || (C & 7) || sw < 1 || sh < 1 || ow < 0 || oh < 0 || Wo < 1 || Ho < 1 || (long)(Wo - 1) * sw + ow >= W ||
        (long)(Ho - 1) * sh + oh >= H)
        return OCR_ERR_INVALID;
    const bf16_t* s = (const bf16_t*)src; bf16_t* d = (bf16_t*)dst; hipStream_t st = (hipStream_t)stream;
    if (backward) subsample_kernel<true><<<dsl_grid((long)Nb * W * H * (C >> 3), 8192), 256, 0, st>>>(s, d, Nb, W, H, C, Wo, Ho, sw, sh, ow, oh);
    else subsample_kernel<false><<<dsl_grid((long)Nb * Wo * Ho * (C >> 3), 8192), 256, 0, st>>>(s, d, Nb, W, H, C, Wo, Ho, sw, sh, ow, oh);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_softmax_f32(const float* in, float* out, long rows, int C, void* stream) {
    if (!in || !out || rows <= 0 || C <= 0) return OCR_ERR_INVALID;
    softmax_rows_kernel<<<(int)(rows > 8192 ? 8192 : rows), 64, 0, (hipStream_t)stream>>>(in, out, rows, C);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
