// bf16 MFMA tile engine for gfx950: C[m][n] = sum_k P[m][k] * Q[n][k]  (both operands K-contiguous)
//
// One engine serves every dense contraction on the CRNN hot path:
//   * a2  conv_single 3x3 SAME  fwd / dgrad as an IMPLICIT GEMM: P rows are output pixels of the
//         reference layout [N, W, H, C] (lib/networks/network.py:160-191), k = (tap, ci); the
//         loader shifts the pixel by the tap and zero-fills the SAME-padding halo.
//   * a2  conv5 2x2 VALID as a plain GEMM over overlapping rows (row-group skip, see below).
//   * a6  the hoisted LSTM input projection, the FC logits (network.py:118-128) and their
//         data-gradient GEMMs.
//
// MFMA orientation: v_mfma_f32_16x16x32_bf16 with A := Q tile (rows -> output channel n) and
// B := P tile (cols -> output row m).  Each lane then owns 4 CONSECUTIVE n of one m, i.e. one
// 8-byte bf16x4 (or 16-byte f32x4) store into the n-contiguous output — no LDS transpose in
// the epilogue.
//
// Tile: 256 threads = 4 waves as 2(m) x 2(n); block tile (32*FM) x (32*FN), BK = 32 per step,
// LDS rows padded to 80 B (5 x 16-B slots: gcd(5,16)=1 spreads a 16-row fragment read over all
// 16 slots of the 256-B bank row).  Global -> register -> LDS staging, next tile's loads issued
// before the current tile's MFMAs (T14-style split).
#include "common.h"

enum {
    EPI_BIAS = 1,        // + bias[n]
    EPI_RELU = 2,        // max(0, .)
    EPI_OUT_F32 = 4,     // fp32 output (else bf16)
    EPI_ATOMIC = 8,      // atomicAdd into fp32 output (split-K)
    EPI_MASK = 16,       // zero where mask[m][n] <= 0 (ReLU backward fused into dgrad)
    EPI_ROWSWAP = 32,    // out row = (m % inner) * outer + m / inner   ([N,T] -> [T,N])
    EPI_ACCUM = 64,      // out += value (fp32, non-atomic; single split only)
};

struct GemmArgs {
    const bf16_t* P;   // m rows
    const bf16_t* Q;   // n rows
    long ldp, ldq;
    int M, N, K;       // K % 8 == 0
    int k_per_split;   // multiple of 32; gridDim.z splits
    // plain-mode row groups: physical row = m + (m / grp) * skip   (conv5's overlapping windows)
    int grp, skip;
    // conv-mode geometry (P = activations [Nb, cW, cH, cC], K = 9*cC, cC % 32 == 0)
    int cW, cH, cC;
    // epilogue
    void* out;
    long ldo;
    const float* bias;
    const bf16_t* mask;
    long ldmask;
    int flags;
    int swap_inner, swap_outer;
};

#define LDS_ROW 40   // bf16 elements per LDS row: 32 payload + 8 pad  (80 B)

template <int MODE /*0 plain, 1 conv3x3*/, int FM, int FN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
    constexpr int BM = 32 * FM, BN = 32 * FN;
    constexpr int NCP = BM * 4 / 256;   // 16-B chunks per thread for the P tile
    constexpr int NCQ = BN * 4 / 256;
    __shared__ __attribute__((aligned(16))) bf16_t Ps[BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) bf16_t Qs[BN * LDS_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    // ---- per-thread staging coordinates (fixed for the whole K loop)
    const bf16_t* prow[NCP];
    bool pvalid[NCP];
    int pw[NCP], ph[NCP];
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        int c = tid + i * 256;
        int r = c >> 2;
        int m = m0 + r;
        pvalid[i] = m < g.M;
        int mm = pvalid[i] ? m : 0;
        if (MODE == 0) {
            long phys = (long)mm + (g.grp > 0 ? (long)(mm / g.grp) * g.skip : 0);
            prow[i] = g.P + phys * g.ldp + (c & 3) * 8;
            pw[i] = ph[i] = 0;
        } else {
            int h = mm % g.cH;
            int q = mm / g.cH;
            int w = q % g.cW;
            pw[i] = w; ph[i] = h;
            prow[i] = g.P + (long)mm * g.cC + (c & 3) * 8;   // pixel (n,w,h), channel chunk
        }
    }
    const bf16_t* qrow[NCQ];
    bool qvalid[NCQ];
#pragma unroll
    for (int i = 0; i < NCQ; ++i) {
        int c = tid + i * 256;
        int r = c >> 2;
        int n = n0 + r;
        qvalid[i] = n < g.N;
        qrow[i] = g.Q + (long)(qvalid[i] ? n : 0) * g.ldq + (c & 3) * 8;
    }

    u32x4 pst[NCP], qst[NCQ];
    auto load_tiles = [&](int k0) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NCP; ++i) {
                int kc = k0 + ((tid + i * 256) & 3) * 8;
                u32x4 v = {0, 0, 0, 0};
                if (pvalid[i] && kc < kend) v = *(const u32x4*)(prow[i] + k0);
                pst[i] = v;
            }
        } else {
            int tap = k0 / g.cC;            // uniform: cC % 32 == 0 keeps a K tile inside one tap
            int ci0 = k0 - tap * g.cC;
            int dw = tap / 3 - 1, dh = tap % 3 - 1;
            long shift = ((long)dw * g.cH + dh) * g.cC + ci0;
#pragma unroll
            for (int i = 0; i < NCP; ++i) {
                int ww = pw[i] + dw, hh = ph[i] + dh;
                bool ok = pvalid[i] && (unsigned)ww < (unsigned)g.cW && (unsigned)hh < (unsigned)g.cH;
                u32x4 v = {0, 0, 0, 0};
                if (ok) v = *(const u32x4*)(prow[i] + shift);
                pst[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < NCQ; ++i) {
            int kc = k0 + ((tid + i * 256) & 3) * 8;
            u32x4 v = {0, 0, 0, 0};
            if (qvalid[i] && kc < kend) v = *(const u32x4*)(qrow[i] + k0);
            qst[i] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NCP; ++i) {
            int c = tid + i * 256;
            *(u32x4*)(&Ps[(c >> 2) * LDS_ROW + (c & 3) * 8]) = pst[i];
        }
#pragma unroll
        for (int i = 0; i < NCQ; ++i) {
            int c = tid + i * 256;
            *(u32x4*)(&Qs[(c >> 2) * LDS_ROW + (c & 3) * 8]) = qst[i];
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    if (kbeg < kend) {
        load_tiles(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            __syncthreads();            // previous tile's fragment reads are done
            store_tiles();
            __syncthreads();
            if (k0 + 32 < kend) load_tiles(k0 + 32);   // in flight during the MFMAs below
            bf16x8 af[FN], bfr[FM];
#pragma unroll
            for (int a = 0; a < FN; ++a)
                af[a] = *(const bf16x8*)(&Qs[(wn * 16 * FN + a * 16 + frow) * LDS_ROW + fk]);
#pragma unroll
            for (int b = 0; b < FM; ++b)
                bfr[b] = *(const bf16x8*)(&Ps[(wm * 16 * FM + b * 16 + frow) * LDS_ROW + fk]);
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns n = nb + (lane>>4)*4 .. +3 at m = mb + (lane&15)
    const int flags = g.flags;
    const bool add_bias = (flags & EPI_BIAS) && (!(flags & EPI_ATOMIC) || blockIdx.z == 0);
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        int n = n0 + wn * 16 * FN + a * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (add_bias) bv = *(const f32x4*)(g.bias + n);
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            int m = m0 + wm * 16 * FM + b * 16 + (lane & 15);
            if (m >= g.M) continue;
            f32x4 v = acc[a][b] + bv;
            if (flags & EPI_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (flags & EPI_MASK) {
                u32x2 mk = *(const u32x2*)(g.mask + (long)m * g.ldmask + n);
                if (!(bf_lo(mk.x) > 0.f)) v.x = 0.f;
                if (!(bf_hi(mk.x) > 0.f)) v.y = 0.f;
                if (!(bf_lo(mk.y) > 0.f)) v.z = 0.f;
                if (!(bf_hi(mk.y) > 0.f)) v.w = 0.f;
            }
            long orow = m;
            if (flags & EPI_ROWSWAP) orow = (long)(m % g.swap_inner) * g.swap_outer + m / g.swap_inner;
            if (flags & EPI_OUT_F32) {
                float* o = (float*)g.out + orow * g.ldo + n;
                if (flags & EPI_ATOMIC) {
                    atomicAdd(o + 0, v.x); atomicAdd(o + 1, v.y); atomicAdd(o + 2, v.z); atomicAdd(o + 3, v.w);
                } else if (flags & EPI_ACCUM) {
                    f32x4 old = *(f32x4*)o;
                    *(f32x4*)o = old + v;
                } else {
                    *(f32x4*)o = v;
                }
            } else {
                u32x2 pk;
                pk.x = pack_bf2(v.x, v.y);
                pk.y = pack_bf2(v.z, v.w);
                *(u32x2*)((bf16_t*)g.out + orow * g.ldo + n) = pk;
            }
        }
    }
}

template <int MODE, int FM, int FN>
static int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 32 * FM, BN = 32 * FN;
    int splits = ceil_div(g.K, g.k_per_split);
    dim3 grid(ceil_div(g.M, BM), ceil_div(g.N, BN), splits);
    gemm_nt_kernel<MODE, FM, FN><<<grid, 256, 0, stream>>>(g);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

static int dispatch_gemm(GemmArgs& g, int mode, int splits, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K & 7) || (g.N & 3)) return OCR_ERR_INVALID;
    if (!g.P || !g.Q || !g.out) return OCR_ERR_INVALID;
    if (splits < 1) splits = 1;
    int kps = ceil_div(ceil_div(g.K, splits), 32) * 32;
    g.k_per_split = kps;
    if (splits > 1) {   // a split request always means "accumulate into the fp32 output with atomics"
        if (!(g.flags & EPI_OUT_F32) || (g.flags & (EPI_RELU | EPI_MASK | EPI_ACCUM))) return OCR_ERR_INVALID;
        g.flags |= EPI_ATOMIC;
    }
    // big tiles when they still fill the chip, small tiles otherwise
    long big_blocks = (long)ceil_div(g.M, 128) * ceil_div(g.N, 128) * ceil_div(g.K, kps);
    bool big = big_blocks >= 192 && g.N >= 128;
    if (mode == 1) {
        if (g.cC % 32 || g.K != 9 * g.cC) return OCR_ERR_INVALID;
        return big ? launch_gemm<1, 4, 4>(g, stream) : launch_gemm<1, 2, 2>(g, stream);
    }
    return big ? launch_gemm<0, 4, 4>(g, stream) : launch_gemm<0, 2, 2>(g, stream);
}

// large-tile LDS-DMA engine (igemm.hip); returns -1 when it does not cover the shape
int ig_try_dispatch(const void* P, long ldp, const void* Q, long ldq, void* out, long ldo, int M, int N, int K,
                    const float* bias, const void* mask, long ldmask, int flags, int mode, int grp, int skip, int cW,
                    int cH, int cC, int swap_inner, int swap_outer, hipStream_t stream);
int halo_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                      const void* mask, int flags, hipStream_t stream, void* pool = nullptr, int pool_kind = 0);
#ifdef OCR_EXPERIMENTS
int pp_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream);
#endif
static int g_use_igemm = 1, g_use_halo = 1, g_use_pp = 0;
extern int g_ig_stages;
// A/B knob. 0 = 128x128 register-staged tiles only; 1 (default) = lock-step tap-reuse halo kernel (conv_halo.hip) for 3x3
// convs + LDS-DMA 256-row tiles (three-stage) for plain GEMMs; 4 = ping-pong halo kernel (conv_pp.hip: measured 14 % SLOWER
// than the lock-step kernel in round 2, kept for A/B), then as 1;
// 2 = LDS-DMA tiles two-stage, no halo; 3 = LDS-DMA tiles three-stage, no halo   (4 only in builds with make EXPERIMENTS=1)
extern "C" int ocr_set_gemm_engine(int mode) {
    g_use_igemm = mode != 0; g_use_halo = mode == 1 || mode == 4; g_use_pp = mode == 4; g_ig_stages = (mode == 2) ? 2 : 3;
    return OCR_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int ocr_gemm_nt_bf16(const void* P, long ldp, const void* Q, long ldq, void* out, long ldo,
                                int M, int N, int K, const float* bias, const void* mask, long ldmask,
                                int flags, int splits, int row_group, int row_skip, int swap_inner,
                                int swap_outer, void* stream) {
    GemmArgs g = {};
    g.P = (const bf16_t*)P; g.Q = (const bf16_t*)Q; g.ldp = ldp; g.ldq = ldq;
    g.M = M; g.N = N; g.K = K; g.grp = row_group; g.skip = row_skip;
    g.out = out; g.ldo = ldo; g.bias = bias; g.mask = (const bf16_t*)mask; g.ldmask = ldmask;
    g.flags = flags & ~EPI_ATOMIC; g.swap_inner = swap_inner; g.swap_outer = swap_outer;
    if ((flags & EPI_BIAS) && !bias) return OCR_ERR_INVALID;
    if ((flags & EPI_MASK) && !mask) return OCR_ERR_INVALID;
    if ((flags & EPI_ROWSWAP) && (swap_inner <= 0 || swap_outer <= 0)) return OCR_ERR_INVALID;
    if (g_use_igemm && splits <= 1 && P && Q && out && M > 0 && N > 0 && K > 0) {
        int rc = ig_try_dispatch(P, ldp, Q, ldq, out, ldo, M, N, K, bias, mask, ldmask, g.flags, 0, row_group, row_skip, 0, 0, 0,
                                 swap_inner, swap_outer, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    return dispatch_gemm(g, 0, splits, (hipStream_t)stream);
}

bool halo_covers(long M, int W, int H, int Cin, int Cout);
// flags may carry OCR_EPI_ACCUM (y += result, bf16) only where this returns non-zero (the halo kernel's epilogue)
extern "C" int ocr_conv3x3_accum_supported(int Nb, int W, int H, int Cin, int Cout) {
    return g_use_halo && !g_use_pp && halo_covers((long)Nb * W * H, W, H, Cin, Cout);
}
// Which kernel family ocr_conv3x3_bf16 / ocr_conv3x3_relu_pool_bf16 would run for this shape (host-only: nothing is launched, no GPU is
// needed): 0 the generic GEMM engines (igemm / gemm), 1 conv_halo, 2 / 3 conv_k2 tile A (256 x 128) / D (256 x 64), 4 / 5 conv_k3 A / D,
// 6 / 7 conv_k3w (general width) A / D, 8 conv_ws (weight-stationary persistent); -OCR_STATUS_INVALID (negative) for non-positive sizes.  flags as for ocr_conv3x3_bf16; (kw, kh) = (0, 0) or the window of a fused max-pool.
extern "C" int ocr_conv3x3_kernel_choice(int Nb, int W, int H, int Cin, int Cout, int flags, int kw, int kh) {
    if (Nb <= 0 || W <= 0 || H <= 0 || Cin <= 0 || Cout <= 0) return -OCR_ERR_INVALID;      // negative: 0..7 are kernel families
    const int pool_kind = (kw == 1 && kh == 2) ? 1 : ((kw == 2 && kh == 2) ? 2 : 0);
    if ((kw || kh) && !pool_kind) return 0;
    const int fl = flags & ~(EPI_ATOMIC | EPI_ROWSWAP);
    if (!g_use_halo) return 0;
    const int rc = halo_try_dispatch(nullptr, nullptr, nullptr, Nb * W * H, W, H, Cin, Cout, (fl & EPI_BIAS) ? (const float*)16 : nullptr,
                                     (fl & EPI_MASK) ? (const void*)16 : nullptr, fl, nullptr, pool_kind ? (void*)16 : nullptr, pool_kind);
    return rc >= 1 ? rc : 0;
}

// 3x3 SAME stride-1 convolution over the reference layout [Nb, W, H, C] as an implicit GEMM.
// wpack is [Cout][3][3][Cin] bf16 (K-contiguous rows).  Used for the forward (wpack = packed
// weights) and for dgrad (x := dY, wpack := flipped/transposed weights, Cin/Cout swapped).
extern "C" int ocr_conv3x3_bf16(const void* x, const void* wpack, void* y, int Nb, int W, int H,
                                int Cin, int Cout, const float* bias, const void* mask, int flags,
                                void* stream) {
    GemmArgs g = {};
    g.P = (const bf16_t*)x; g.Q = (const bf16_t*)wpack; g.ldp = Cin; g.ldq = 9L * Cin;
    g.M = Nb * W * H; g.N = Cout; g.K = 9 * Cin; g.cW = W; g.cH = H; g.cC = Cin;
    g.out = y; g.ldo = Cout; g.bias = bias; g.mask = (const bf16_t*)mask; g.ldmask = Cout;
    g.flags = flags & ~(EPI_ATOMIC | EPI_ROWSWAP);
    if ((flags & EPI_BIAS) && !bias) return OCR_ERR_INVALID;
    if ((flags & EPI_MASK) && !mask) return OCR_ERR_INVALID;
    if ((flags & EPI_ACCUM) && !ocr_conv3x3_accum_supported(Nb, W, H, Cin, Cout)) return OCR_ERR_INVALID;      // bf16 accumulate: halo epilogue only
#ifdef OCR_EXPERIMENTS
    if (g_use_pp && x && wpack && y && g.M > 0 && Cout > 0 && Cin > 0) {
        int rc = pp_try_dispatch(x, wpack, y, g.M, W, H, Cin, Cout, bias, mask, g.flags, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
#endif
    if (g_use_halo && x && wpack && y && g.M > 0 && Cout > 0 && Cin > 0) {
        int rc = halo_try_dispatch(x, wpack, y, g.M, W, H, Cin, Cout, bias, mask, g.flags, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    if (g_use_igemm && x && wpack && y && g.M > 0 && Cout > 0 && Cin > 0) {
        int rc = ig_try_dispatch(x, Cin, wpack, 9L * Cin, y, Cout, g.M, Cout, 9 * Cin, bias, mask, Cout, g.flags, 1, 0, 0, W, H, Cin,
                                 0, 0, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    return dispatch_gemm(g, 1, 1, (hipStream_t)stream);
}
// conv3x3 (+ bias, optional ReLU) that also leaves the batch-norm statistics of its output behind (network.py:173-178: conv -> bias ->
// batch_norm): partials = float [rows][2][Cout], rows = ocr_conv3x3_stats_rows(...) = Nb*W*H / 256 — per 256-pixel tile the per-channel sum
// and sum of squares of the bf16 values that were stored.  ocr_bn_train_fwd2(partial_rows = rows) finishes from them without reading the
// tensor for a statistics pass.  Only where the plane-layout kernels (conv_k3 / conv_k3w) take the shape: the query returns 0 otherwise
// and the call OCR_STATUS_INVALID (the caller then runs ocr_conv3x3_bf16 + the three-pass batch norm).
extern "C" int ocr_conv3x3_stats_rows(int Nb, int W, int H, int Cin, int Cout, int flags) {
    if (Nb <= 0 || W <= 0 || H <= 0 || Cin <= 0 || Cout <= 0 || !g_use_halo || g_use_pp) return 0;
    const long M = (long)Nb * W * H;
    const int fl = flags & ~(EPI_ATOMIC | EPI_ROWSWAP);
    if (M > 0x7fffffffL || (M & 255) || (fl & (EPI_MASK | EPI_ACCUM))) return 0;
    const int rc = halo_try_dispatch(nullptr, nullptr, nullptr, (int)M, W, H, Cin, Cout, (fl & EPI_BIAS) ? (const float*)16 : nullptr, nullptr, fl,
                                     nullptr, (void*)16, 3);
    return rc >= 4 ? (int)(M / 256) : 0;
}
extern "C" int ocr_conv3x3_bf16_stats(const void* x, const void* wpack, void* y, int Nb, int W, int H, int Cin, int Cout, const float* bias,
                                      int flags, float* partials, void* stream) {
    if (!x || !wpack || !y || !partials || ((flags & EPI_BIAS) && !bias) || !ocr_conv3x3_stats_rows(Nb, W, H, Cin, Cout, flags)) return OCR_ERR_INVALID;
    const int rc = halo_try_dispatch(x, wpack, y, Nb * W * H, W, H, Cin, Cout, bias, nullptr, flags & ~(EPI_ATOMIC | EPI_ROWSWAP), (hipStream_t)stream,
                                     partials, 3);
    return rc >= 0 ? rc : OCR_ERR_INVALID;
}
// The data gradient of a convolution whose INPUT came from a batch-norm + ReLU layer (network.py:176-182 -> the next conv_single): dx = (y > 0) ?
// conv3x3(dy, flipped weights) : 0 as ocr_conv3x3_bf16 with OCR_EPI_MASK writes it, and in the same epilogue the two per-channel sums the
// batch-norm backward pass of that producer needs — partials [rows][2][Cout] = (sum dx, sum dx * xhat) per 256-pixel tile, xhat = (z - mean) *
// rstd of the producer's pre-normalisation output z.  ocr_bn_train_bwd2(partial_rows = rows, premasked) finishes from them: no statistics
// pass over (z, y, dy), no second read of y.  rows = ocr_conv3x3_bnbwd_rows(...) (0: not covered - use the separate passes).
struct K3BnBwdHost { float* partials; const void* z; const float* mean; const float* rstd; };      // = conv_k3.hip: K3BnBwd
extern "C" int ocr_conv3x3_bnbwd_rows(int Nb, int W, int H, int Cin, int Cout) {
    if (Nb <= 0 || W <= 0 || H <= 0 || Cin <= 0 || Cout <= 0 || !g_use_halo || g_use_pp) return 0;
    const long M = (long)Nb * W * H;
    if (M > 0x7fffffffL || (M & 255)) return 0;
    const int rc = halo_try_dispatch(nullptr, nullptr, nullptr, (int)M, W, H, Cin, Cout, nullptr, (const void*)16, EPI_MASK, nullptr, (void*)16, 4);
    return rc >= 4 ? (int)(M / 256) : 0;
}
extern "C" int ocr_conv3x3_dgrad_bnbwd_bf16(const void* dy, const void* wdgrad, void* dx, int Nb, int W, int H, int Cin, int Cout, const void* mask_y,
                                            const void* z, const float* mean, const float* rstd, float* partials, void* stream) {
    if (!dy || !wdgrad || !dx || !mask_y || !z || !mean || !rstd || !partials || !ocr_conv3x3_bnbwd_rows(Nb, W, H, Cin, Cout)) return OCR_ERR_INVALID;
    K3BnBwdHost e = {partials, z, mean, rstd};
    const int rc = halo_try_dispatch(dy, wdgrad, dx, Nb * W * H, W, H, Cin, Cout, nullptr, mask_y, EPI_MASK, (hipStream_t)stream, &e, 4);
    return rc >= 0 ? rc : OCR_ERR_INVALID;
}
// conv3x3 + bias + ReLU with the max-pool that follows it (LSTM_train.py:26-33: conv2 -> pool2 2 x 2, conv3_2 / conv4_2 -> 1 x 2
// over the feature axis) written by the same epilogue: y (kept for the backward pass) AND pooled.  kw = window along W (time),
// kh = window along H (feature): (1, 2) or (2, 2).  Only the halo kernel has this epilogue: OCR_ERR_INVALID when the shape is
// not covered (ocr_conv3x3_pool_supported tells beforehand) - the caller then runs ocr_conv3x3_bf16 + ocr_maxpool_fwd.
static int conv3x3_pool_kind(int kw, int kh) { return (kw == 1 && kh == 2) ? 1 : ((kw == 2 && kh == 2) ? 2 : 0); }
extern "C" int ocr_conv3x3_pool_supported(int Nb, int W, int H, int Cin, int Cout, int kw, int kh) {
    const long M = (long)Nb * W * H;
    const int kind = conv3x3_pool_kind(kw, kh);
    if (!g_use_halo || g_use_pp || !kind || (Cin & 63) || (Cout & 63) || M < 1024 || M > 0x7fffffffL || H > 30 || (H & 1)) return 0;
    if (kind == 2 && ((H != 4 && H != 8 && H != 16) || (W & 1))) return 0;
    const int NR = 128 + 2 * H + 2, NRp = (NR + 32) / 32 * 32;           // the 4-wave instances (counted vmcnt literals for 5 and 6)
    return NRp / 32 == 5 || NRp / 32 == 6;
}
extern "C" int ocr_conv3x3_relu_pool_bf16(const void* x, const void* wpack, void* y, void* pooled, int Nb, int W, int H, int Cin,
                                          int Cout, const float* bias, int kw, int kh, void* stream) {
    if (!x || !wpack || !y || !pooled || !bias || !ocr_conv3x3_pool_supported(Nb, W, H, Cin, Cout, kw, kh)) return OCR_ERR_INVALID;
    int rc = halo_try_dispatch(x, wpack, y, Nb * W * H, W, H, Cin, Cout, bias, nullptr, EPI_BIAS | EPI_RELU, (hipStream_t)stream, pooled,
                               conv3x3_pool_kind(kw, kh));
    return rc >= 0 ? rc : OCR_ERR_INVALID;
}
