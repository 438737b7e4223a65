// Weight-gradient contraction for gfx950: D[i][j] += sum_m A[m][i] * B[m][j]
// (the contraction index m is the SLOW index of both operands: "TN" form).
//
// Serves every weight gradient of the reference graph (tf.gradients at lib/lstm/train.py:81):
//   * conv_single 3x3 SAME (network.py:160-191):  dW[kh,kw,ci,co] = sum_pixels X[shifted][ci] * dY[pix][co]
//     (implicit: the A row for pixel m and tap (kh,kw) is the pixel shifted by the tap, zero in the halo)
//   * conv5 2x2 VALID, LSTM Wx / Wh, FC:  plain  dW[k_in][n_out] = X^T dY  (row-group skip for conv5)
// The result is accumulated with fp32 atomics straight into the TF-layout gradient buffer
// ([kh,kw,ci,co] resp. [in,out]), so split-M partial sums and the caller's "+=" are one mechanism.
//
// MFMA wants 8 consecutive k (= m) per lane, while HBM has the channel axis contiguous, so the
// operands are transposed on the way through LDS with gfx950's transposing read ds_read_b64_tr_b16:
// a 16-lane group hands in 16 x 8-byte chunks covering a 4(k) x 16(col) row-major block (lane L supplies row
// L>>2, columns 4*(L&3)..+3) and lane L receives column L, i.e. 4 consecutive k of ITS output column
// (semantics pinned on the device by tests/test_gpu_kernels.py::test_probe_tr16).  Two reads give the 8 k
// slots of a v_mfma_f32_16x16x32_bf16 operand.  Lane group g reads tile rows 4g..4g+3 and 16+4g..16+4g+3 — a
// k permutation applied identically to A and B, so the contraction is unchanged — which, with LDS rows of
// (tile width + 16) bf16 (row stride = 8 dwords mod 64), makes every 32-lane half of the read hit 64
// distinct banks.  USE_TR = false keeps the plain 2-byte gather (same k order) for A/B comparison.
#include "common.h"

struct TnArgs {
    const bf16_t* A;  // [Mrows][lda]  (i contiguous)
    const bf16_t* B;  // [Mrows][ldb]  (j contiguous)
    long lda, ldb;
    int Mk;           // contraction length (rows of B)
    int I, J;         // output dims, I % 8 == 0, J % 8 == 0
    int k_per_split;  // multiple of 32
    int grp, skip;    // plain mode: physical A row = m + (m/grp)*skip ; A row offset added below
    long a_row_off;   // plain mode: extra element offset applied to every A row (conv5 second tap row)
    int cW, cH, cC;   // conv mode: A = activations [Nb, cW, cH, cC], taps = gridDim.x / itiles
    float* out;       // conv: [9][cC][J] ; plain: [I][ldo]
    long ldo;
    float scale;      // multiply before accumulation (e.g. 1.0)
    float* colsum;    // optional: colsum[j] += sum_m B[m][j]  (bias gradient fused into the weight-gradient pass)
};

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// 8 k-slots of output column (col0 + lane&15) from a row-major [32][ld] bf16 tile.
// slot j<4 <-> tile row 4g+j, slot 4+j <-> tile row 16+4g+j  (g = lane>>4)
template <bool USE_TR>
__device__ __forceinline__ bf16x8 load_frag(const bf16_t* tile, int ld, int col0, int lane) {
    const int g = lane >> 4, L = lane & 15;
    if (USE_TR) {
        const bf16_t* p0 = tile + (4 * g + (L >> 2)) * ld + col0 + (L & 3) * 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * ld));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    } else {
        u16x8 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = tile[(4 * g + j) * ld + col0 + L];
            t[4 + j] = tile[(16 + 4 * g + j) * ld + col0 + L];
        }
        return __builtin_bit_cast(bf16x8, t);
    }
}

template <int MODE /*0 plain, 1 conv3x3*/, int FI, int FJ, bool USE_TR = true>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs g) {
    constexpr int BI = 32 * FI, BJ = 32 * FJ;
    constexpr int LDA = BI + 16, LDB = BJ + 16;
    constexpr int CPA = BI / 8, CPB = BJ / 8;          // 16-B chunks per tile row
    constexpr int NCA = 32 * CPA / 256, NCB = 32 * CPB / 256;
    static_assert(NCA >= 1 && NCB >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) bf16_t As[32 * LDA];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[32 * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int itiles = ceil_div(g.I, BI);
    const int tap = (MODE == 1) ? (int)(blockIdx.x / itiles) : 0;
    const int i0 = (int)(blockIdx.x % itiles) * BI;
    const int j0 = blockIdx.y * BJ;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.Mk, kbeg + g.k_per_split);
    const int dw = tap / 3 - 1, dh = tap % 3 - 1;
    // the blocks that see every B row exactly once (first i-tile, centre tap) also produce the column sums of B
    const bool do_cs = g.colsum != nullptr && i0 == 0 && (MODE == 0 || tap == 4);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    u32x4 ast[NCA], bst[NCB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int t = 0; t < NCA; ++t) {
            int c = tid + t * 256;
            int r = c / CPA, col = (c % CPA) * 8;
            int m = k0 + r;
            u32x4 v = {0, 0, 0, 0};
            bool ok = m < kend && (i0 + col) < g.I;
            if (ok) {
                if (MODE == 0) {
                    long phys = (long)m + (g.grp > 0 ? (long)(m / g.grp) * g.skip : 0);
                    v = *(const u32x4*)(g.A + phys * g.lda + g.a_row_off + i0 + col);
                } else {
                    int h = m % g.cH, q = m / g.cH, w = q % g.cW;
                    int ww = w + dw, hh = h + dh;
                    if ((unsigned)ww < (unsigned)g.cW && (unsigned)hh < (unsigned)g.cH)
                        v = *(const u32x4*)(g.A + ((long)m + (long)dw * g.cH + dh) * g.cC + i0 + col);
                }
            }
            ast[t] = v;
        }
#pragma unroll
        for (int t = 0; t < NCB; ++t) {
            int c = tid + t * 256;
            int r = c / CPB, col = (c % CPB) * 8;
            int m = k0 + r;
            u32x4 v = {0, 0, 0, 0};
            if (m < kend && (j0 + col) < g.J) v = *(const u32x4*)(g.B + (long)m * g.ldb + j0 + col);
            bst[t] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int t = 0; t < NCA; ++t) {
            int c = tid + t * 256;
            *(u32x4*)(&As[(c / CPA) * LDA + (c % CPA) * 8]) = ast[t];
        }
#pragma unroll
        for (int t = 0; t < NCB; ++t) {
            int c = tid + t * 256;
            *(u32x4*)(&Bs[(c / CPB) * LDB + (c % CPB) * 8]) = bst[t];
        }
    };

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (kbeg < kend) {
        load_tiles(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            __syncthreads();
            store_tiles();
            if (do_cs) {
#pragma unroll
                for (int t = 0; t < NCB; ++t) {
                    cs[0] += bf_lo(bst[t].x); cs[1] += bf_hi(bst[t].x); cs[2] += bf_lo(bst[t].y); cs[3] += bf_hi(bst[t].y);
                    cs[4] += bf_lo(bst[t].z); cs[5] += bf_hi(bst[t].z); cs[6] += bf_lo(bst[t].w); cs[7] += bf_hi(bst[t].w);
                }
            }
            __syncthreads();
            if (k0 + 32 < kend) load_tiles(k0 + 32);
            bf16x8 af[FI], bfr[FJ];
#pragma unroll
            for (int a = 0; a < FI; ++a) af[a] = load_frag<USE_TR>(As, LDA, wi * 16 * FI + a * 16, lane);
#pragma unroll
            for (int b = 0; b < FJ; ++b) bfr[b] = load_frag<USE_TR>(Bs, LDB, wj * 16 * FJ + b * 16, lane);
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    }

    if (do_cs) {   // every thread's 8 columns are (tid % CPB)*8 .. +7; reduce the 256/CPB row lanes through LDS
        __shared__ float red[256 * 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
        __syncthreads();
        if (tid < BJ && j0 + tid < g.J) {
            float t = 0.f;
            for (int u = tid >> 3; u < 256; u += CPB) t += red[u * 8 + (tid & 7)];
            atomicAdd(g.colsum + j0 + tid, t * g.scale);
        }
    }
    // lane owns i = ib + (lane>>4)*4 + r, j = jb + (lane&15)
    float* out = g.out;
    long ldo = g.ldo;
    if (MODE == 1) { out += (long)tap * g.cC * g.J; ldo = g.J; }
#pragma unroll
    for (int a = 0; a < FI; ++a) {
#pragma unroll
        for (int b = 0; b < FJ; ++b) {
            int j = j0 + wj * 16 * FJ + b * 16 + (lane & 15);
            if (j >= g.J) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int i = i0 + wi * 16 * FI + a * 16 + (lane >> 4) * 4 + r;
                if (i < g.I) atomicAdd(out + (long)i * ldo + j, acc[a][b][r] * g.scale);
            }
        }
    }
}

template <int MODE, int FI, int FJ>
static int launch_tn(const TnArgs& g, int taps, hipStream_t stream) {
    constexpr int BI = 32 * FI, BJ = 32 * FJ;
    dim3 grid(ceil_div(g.I, BI) * taps, ceil_div(g.J, BJ), ceil_div(g.Mk, g.k_per_split));
    gemm_tn_kernel<MODE, FI, FJ><<<grid, 256, 0, stream>>>(g);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

static int pick_splits(long tiles, int Mk) {
    // aim for >= ~1024 workgroups, at least 4 K-steps (128 rows) per split
    int s = (int)((1024 + tiles - 1) / tiles);
    int maxs = Mk / 128; if (maxs < 1) maxs = 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return s;
}

// LDS-DMA generation (gemm_tn2.hip); -1 = shape not covered
int tn2_try_dispatch(const void* A, long lda, const void* B, long ldb, float* out, long ldo, int Mk, int I, int J, int mode,
                     int grp, int skip, long a_row_off, int cW, int cH, int cC, float scale, int splits, float* colsum,
                     hipStream_t stream, int nbatch = 1, long sA = 0, long sB = 0, long sO = 0, long sC = 0);
int wgrad9_try_dispatch(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin, int Cout,
                        void* workspace, size_t ws_bytes, hipStream_t stream, void* defer_job = nullptr, int* defer_blocks = nullptr);
static int g_use_tn2 = 1, g_use_w9 = 1;
// 2 (default): nine-tap slab kernel (wgrad9.hip) where a workspace is given and the shape is covered, else as 1;
// 1: LDS-DMA per-tap tiles with fp32 atomics (gemm_tn2.hip); 0: register-staged kernel (this file)
extern "C" int ocr_set_wgrad_engine(int engine) { g_use_tn2 = engine != 0; g_use_w9 = engine >= 2; return OCR_OK; }

// out[I][ldo] += scale * A^T B   with A[Mk][lda] (row-group skip + fixed offset), B[Mk][ldb]
extern "C" int ocr_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* out, long ldo,
                                int Mk, int I, int J, int row_group, int row_skip, long a_row_off,
                                float scale, int splits, float* colsum, void* stream) {
    if (!A || !B || !out || Mk <= 0 || I <= 0 || J <= 0 || (I & 7) || (J & 7)) return OCR_ERR_INVALID;
    TnArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.lda = lda; g.ldb = ldb; g.Mk = Mk; g.I = I; g.J = J;
    if (g_use_tn2) {
        int rc = tn2_try_dispatch(A, lda, B, ldb, out, ldo, Mk, I, J, 0, row_group, row_skip, a_row_off, 0, 0, 0, scale, splits,
                                  colsum, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    g.grp = row_group; g.skip = row_skip; g.a_row_off = a_row_off; g.out = out; g.ldo = ldo; g.scale = scale; g.colsum = colsum;
    bool big = (I >= 128 && J >= 128);
    long tiles = big ? (long)ceil_div(I, 128) * ceil_div(J, 128) : (long)ceil_div(I, 64) * ceil_div(J, 64);
    if (splits <= 0) splits = pick_splits(tiles, Mk);
    g.k_per_split = ceil_div(ceil_div(Mk, splits), 32) * 32;
    return big ? launch_tn<0, 4, 4>(g, 1, (hipStream_t)stream) : launch_tn<0, 2, 2>(g, 1, (hipStream_t)stream);
}

// `nbatch` independent products of one shape in ONE launch: problem b uses A + b*strideA, B + b*strideB, out + b*strideOut,
// colsum + b*strideColsum (element strides).  Used for the two directions of the BiLSTM weight gradient (network.py:104-107).
extern "C" int ocr_gemm_tn_batched_bf16(const void* A, long lda, long strideA, const void* B, long ldb, long strideB, float* out,
                                        long ldo, long strideOut, int Mk, int I, int J, int nbatch, float scale, int splits,
                                        float* colsum, long strideColsum, void* stream) {
    if (!A || !B || !out || Mk <= 0 || I <= 0 || J <= 0 || (I & 7) || (J & 7) || nbatch <= 0) return OCR_ERR_INVALID;
    if (g_use_tn2 && nbatch > 1) {
        int rc = tn2_try_dispatch(A, lda, B, ldb, out, ldo, Mk, I, J, 0, 0, 0, 0, 0, 0, 0, scale, splits, colsum, (hipStream_t)stream,
                                  nbatch, strideA, strideB, strideOut, strideColsum);
        if (rc >= 0) return rc;
    }
    for (int b = 0; b < nbatch; ++b) {
        int rc = ocr_gemm_tn_bf16((const bf16_t*)A + b * strideA, lda, (const bf16_t*)B + b * strideB, ldb, out + b * strideOut, ldo, Mk,
                                  I, J, 0, 0, 0, scale, splits, colsum ? colsum + b * strideColsum : nullptr, stream);
        if (rc != OCR_OK) return rc;
    }
    return OCR_OK;
}

// the same with a caller-owned workspace (ocr_conv3x3_wgrad_workspace_size): deterministic slab reduction instead of atomics
extern "C" int ocr_conv3x3_wgrad_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H,
                                      int Cin, int Cout, int splits, void* stream);
extern "C" int ocr_conv3x3_wgrad_ws_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin,
                                         int Cout, int splits, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || Nb <= 0 || W <= 0 || H <= 0 || (Cin & 7) || (Cout & 7)) return OCR_ERR_INVALID;
    if (g_use_w9 && workspace && splits <= 0) {
        int rc = wgrad9_try_dispatch(x, dy, dw, dbias, Nb, W, H, Cin, Cout, workspace, workspace_bytes, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    return ocr_conv3x3_wgrad_bf16(x, dy, dw, dbias, Nb, W, H, Cin, Cout, splits, stream);
}

// Deferred form for a backward pass that reduces the slabs of all its layers in ONE launch (ocr_wgrad9_reduce_jobs): where the slab
// kernel covers the shape, only it runs here, `job` (64 bytes of HOST memory) receives the description of the pending reduction,
// *job_blocks its block count and *deferred = 1; the workspace must then stay untouched until that reduction has run.  Otherwise the
// whole weight gradient is computed at once as by ocr_conv3x3_wgrad_ws_bf16 and *deferred = 0.
extern "C" int ocr_conv3x3_wgrad_defer_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin,
                                            int Cout, void* workspace, size_t workspace_bytes, void* job, int* job_blocks,
                                            int* deferred, void* stream) {
    if (!x || !dy || !dw || !job || !job_blocks || !deferred || Nb <= 0 || W <= 0 || H <= 0 || (Cin & 7) || (Cout & 7)) return OCR_ERR_INVALID;
    *deferred = 0;
    if (g_use_w9 && workspace) {
        int rc = wgrad9_try_dispatch(x, dy, dw, dbias, Nb, W, H, Cin, Cout, workspace, workspace_bytes, (hipStream_t)stream, job, job_blocks);
        if (rc >= 0) { if (rc == OCR_OK) *deferred = 1; return rc; }
    }
    return ocr_conv3x3_wgrad_bf16(x, dy, dw, dbias, Nb, W, H, Cin, Cout, 0, stream);
}

// dW[3][3][Cin][Cout] (fp32, TF layout) += sum over pixels of x[shifted pixel][ci] * dy[pixel][co]
extern "C" int ocr_conv3x3_wgrad_bf16(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H,
                                      int Cin, int Cout, int splits, void* stream) {
    if (!x || !dy || !dw || Nb <= 0 || W <= 0 || H <= 0 || (Cin & 7) || (Cout & 7)) return OCR_ERR_INVALID;
    if (g_use_tn2) {
        int rc = tn2_try_dispatch(x, Cin, dy, Cout, dw, Cout, Nb * W * H, Cin, Cout, 1, 0, 0, 0, W, H, Cin, 1.0f, splits, dbias,
                                  (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    TnArgs g = {};
    g.A = (const bf16_t*)x; g.B = (const bf16_t*)dy; g.lda = Cin; g.ldb = Cout;
    g.Mk = Nb * W * H; g.I = Cin; g.J = Cout; g.cW = W; g.cH = H; g.cC = Cin;
    g.out = dw; g.ldo = Cout; g.scale = 1.0f; g.colsum = dbias;
    bool big = (Cin >= 128 && Cout >= 128);
    long tiles = 9L * (big ? (long)ceil_div(Cin, 128) * ceil_div(Cout, 128) : (long)ceil_div(Cin, 64) * ceil_div(Cout, 64));
    if (splits <= 0) splits = pick_splits(tiles, g.Mk);
    g.k_per_split = ceil_div(ceil_div(g.Mk, splits), 32) * 32;
    return big ? launch_tn<1, 4, 4>(g, 9, (hipStream_t)stream) : launch_tn<1, 2, 2>(g, 9, (hipStream_t)stream);
}
