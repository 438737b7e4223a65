// Bidirectional LSTM time-step kernels for gfx950.
//
// Reference semantics: tf.contrib.rnn.LSTMCell(num_hids//2) x 2 under tf.nn.bidirectional_dynamic_rnn
// with per-sample sequence lengths (lib/networks/network.py:104-109).  TF-1.0 LSTMCell:
//   z = [x_t, h_{t-1}] W + b, split (i, j, f, o);  c = sigmoid(f + 1.0) c_prev + sigmoid(i) tanh(j);
//   h = sigmoid(o) tanh(c);  beyond a sample's length the output is 0 and the state is frozen;
//   the backward direction runs over the sequence reversed WITHIN its length.
//
// Design: the input projection x W_x (+ b) for all time steps and both directions is hoisted into one
// MFMA GEMM (gemm.hip); what remains per step is z += h_{t-1} W_h with M = batch, K = U, N = 4U - far too
// small and too sequential for a tiled GEMM.  One launch per step covers both directions:
//   grid = (U/16 unit tiles, 2 directions, batch/64), 4 waves per workgroup, wave = 16 batch rows.
//   Gate columns are re-packed so that a workgroup's 64 rows of W_h^T are (gate g, local unit ul):
//   the 16x16 MFMA accumulator of fragment g then hands every lane the SAME 4 consecutive units of
//   gate g, i.e. all four gates of a unit meet in one lane's registers with no LDS exchange, and
//   h / c / gate stores are 8- and 16-byte vectors.
//   h_{t-1} is read straight from the layer output tensor (row of the previous step), so there is no
//   separate recurrent state buffer to ping-pong; c_{t-1} likewise comes from the saved cell tensor.
//
// Packed gate column:  p(g, u) = (u / 16) * 64 + g * 16 + (u % 16)      (U % 16 == 0)
#include "common.h"
#include <math.h>

struct LstmFwdArgs {
    const float* xproj;     // [R][2][4U] packed columns, bias already added
    const bf16_t* whT;      // [2][4U packed rows][U]
    const int* seq_len;     // [Nb]
    bf16_t* hout;           // [R][2U]   (fw | bw)
    float* gates;           // [2][R][4U] packed, post-activation i, j, f, o
    float* cell;            // [2][R][U]
    int Nb, T, U;
    int step;
    float forget_bias;
    int ND;                 // directions: 2 = fw | bw (bi_lstm), 1 = forward only (Network.lstm); row strides are ND * U / ND * 4U
};

__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, d = blockIdx.y;
    const int U = a.U, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = blockIdx.z * 64 + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const int s = a.step;
    const bool active = nvalid && s < len;
    const int t = active ? (d == 0 ? s : len - 1 - s) : s;           // inactive rows zero-fill frame s
    const int tprev = (d == 0) ? t - 1 : t + 1;
    const long R = (long)a.Nb * T;
    const long row = (long)nn * T + t;
    const long rowp = (long)nn * T + ((active && s > 0) ? tprev : 0);

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // prefetch the operands that do not depend on the MFMA result
    const int ul0 = q * 4;
    const float* xp = a.xproj + row * ((long)a.ND * 4 * U) + (long)d * 4 * U + (long)ub * 64 + ul0;
    f32x4 xi = {0.f, 0.f, 0.f, 0.f}, xj = xi, xf = xi, xo = xi, cprev = xi;
    if (active) {
        xi = *(const f32x4*)(xp + 0); xj = *(const f32x4*)(xp + 16); xf = *(const f32x4*)(xp + 32); xo = *(const f32x4*)(xp + 48);
        if (s > 0) cprev = *(const f32x4*)(a.cell + ((long)d * R + rowp) * U + ub * 16 + ul0);
    }
    if (s > 0) {
        const bf16_t* wbase = a.whT + ((long)d * 4 * U + (long)ub * 64 + nl) * U + q * 8;
        const bf16_t* hbase = a.hout + rowp * ((long)a.ND * U) + (long)d * U + q * 8;
        // K = U is consumed in chunks of 128 with all 20 operand loads of a chunk in flight at once
        for (int kc = 0; kc < U; kc += 128) {
            bf16x8 b[4], w[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bool in = kc + kk * 32 < U;          // U % 32 == 0; tail chunks contribute zeros
                const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                b[kk] = in ? *(const bf16x8*)(hbase + kc + kk * 32) : zero;
#pragma unroll
                for (int g = 0; g < 4; ++g) w[g][kk] = in ? *(const bf16x8*)(wbase + (long)g * 16 * U + kc + kk * 32) : zero;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][kk], b[kk], acc[g], 0, 0, 0);
        }
    }
    // lane owns units u = ub*16 + q*4 + r (r = 0..3) of batch row n, gate g in acc[g][r]
    if (!nvalid) return;
    bf16_t* hdst = a.hout + row * ((long)a.ND * U) + (long)d * U + ub * 16 + ul0;
    if (!active) {
        if (s < T) { u32x2 z = {0u, 0u}; *(u32x2*)hdst = z; }
        return;
    }
    f32x4 zi = xi + acc[0], zj = xj + acc[1], zf = xf + acc[2], zo = xo + acc[3];
    f32x4 gi, gj, gf, go, cn, hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        gi[r] = sigmoidf_(zi[r]);
        gj[r] = tanhf_(zj[r]);
        gf[r] = sigmoidf_(zf[r] + a.forget_bias);
        go[r] = sigmoidf_(zo[r]);
        cn[r] = gf[r] * cprev[r] + gi[r] * gj[r];
        hn[r] = go[r] * tanhf_(cn[r]);
    }
    float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0;
    *(f32x4*)(gdst + 0) = gi;
    *(f32x4*)(gdst + 16) = gj;
    *(f32x4*)(gdst + 32) = gf;
    *(f32x4*)(gdst + 48) = go;
    *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = cn;
    u32x2 hp = {pack_bf2(hn[0], hn[1]), pack_bf2(hn[2], hn[3])};
    *(u32x2*)hdst = hp;
}

struct LstmBwdArgs {
    const bf16_t* wh;       // [2][U rows (h unit k)][4U] = rows U_in.. of the TF weight, master column order, ld = ldw
    long ldw;               // row stride (elements) of wh
    long w_dir_stride;      // element offset between the two directions' matrices
    const int* seq_len;
    const bf16_t* dhout;    // [R][2U] gradient w.r.t. the layer output
    const float* gates;     // [2][R][4U] packed
    const float* cell;      // [2][R][U]
    bf16_t* dz;             // [R][2][4U] master gate-column order (g*U + u)
    float* dc_state;        // [2][Nb][U]  (zeroed before the first call)
    int Nb, T, U;
    int step;
    int ND;
};

__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, d = blockIdx.y;
    const int U = a.U, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = blockIdx.z * 64 + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const int s = a.step;
    const bool active = nvalid && s < len;
    const bool has_next = nvalid && (s + 1 < len);                    // step s+1 exists for this sample
    const int t = active ? (d == 0 ? s : len - 1 - s) : s;
    const int tnext = (d == 0) ? t + 1 : t - 1;                        // frame processed at step s+1
    const int tprev = (d == 0) ? t - 1 : t + 1;                        // frame processed at step s-1
    const long R = (long)a.Nb * T;
    const long row = (long)nn * T + t;
    const long rown = (long)nn * T + (has_next ? tnext : 0);

    // dh_rec[u][n] = sum_k Wh[u][k] * dz_{s+1}[n][k],  k over the 4U gate columns (master order)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    if (s + 1 < T) {
        const bf16_t* wbase = a.wh + (long)d * a.w_dir_stride + ((long)ub * 16 + nl) * a.ldw + q * 8;
        const bf16_t* zbase = a.dz + rown * ((long)a.ND * 4 * U) + (long)d * 4 * U + q * 8;
        const int K = 4 * U;
        for (int k0 = 0; k0 < K; k0 += 256) {
            bf16x8 w[8], z[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bool in = k0 + kk * 32 < K;
                const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                w[kk] = in ? *(const bf16x8*)(wbase + k0 + kk * 32) : zero;
                z[kk] = in ? *(const bf16x8*)(zbase + k0 + kk * 32) : zero;
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], z[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], z[1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], z[2], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[3], z[3], acc3, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[4], z[4], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[5], z[5], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[6], z[6], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[7], z[7], acc3, 0, 0, 0);
        }
    }
    if (!nvalid) return;
    const int ul0 = q * 4;                    // lane owns units ub*16 + ul0 + r
    const int u0 = ub * 16 + ul0;
    bf16_t* zdst = a.dz + row * ((long)a.ND * 4 * U) + (long)d * 4 * U + u0;
    float* dcs = a.dc_state + ((long)d * a.Nb + nn) * U + u0;
    if (!active) {
        if (s < T) {
            u32x2 z = {0u, 0u};
#pragma unroll
            for (int g = 0; g < 4; ++g) *(u32x2*)(zdst + (long)g * U) = z;
        }
        return;
    }
    f32x4 dh = (acc0 + acc1) + (acc2 + acc3);
    if (!has_next) dh = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        u32x2 g2 = *(const u32x2*)(a.dhout + row * ((long)a.ND * U) + (long)d * U + u0);
        dh[0] += bf_lo(g2.x); dh[1] += bf_hi(g2.x); dh[2] += bf_lo(g2.y); dh[3] += bf_hi(g2.y);
    }
    const float* gsrc = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0;
    f32x4 gi = *(const f32x4*)(gsrc + 0), gj = *(const f32x4*)(gsrc + 16);
    f32x4 gf = *(const f32x4*)(gsrc + 32), go = *(const f32x4*)(gsrc + 48);
    f32x4 c = *(const f32x4*)(a.cell + ((long)d * R + row) * U + u0);
    f32x4 cprev = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) cprev = *(const f32x4*)(a.cell + ((long)d * R + (long)nn * T + tprev) * U + u0);
    f32x4 dcv = *(const f32x4*)dcs;
    f32x4 di, dj, df, dov, dcn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float tc = tanhf_(c[r]);
        float dc = dcv[r] + dh[r] * go[r] * (1.f - tc * tc);
        dov[r] = dh[r] * tc * go[r] * (1.f - go[r]);
        di[r] = dc * gj[r] * gi[r] * (1.f - gi[r]);
        dj[r] = dc * gi[r] * (1.f - gj[r] * gj[r]);
        df[r] = dc * cprev[r] * gf[r] * (1.f - gf[r]);
        dcn[r] = dc * gf[r];
    }
    *(f32x4*)dcs = dcn;
    u32x2 p;
    p.x = pack_bf2(di[0], di[1]); p.y = pack_bf2(di[2], di[3]); *(u32x2*)(zdst + 0L * U) = p;
    p.x = pack_bf2(dj[0], dj[1]); p.y = pack_bf2(dj[2], dj[3]); *(u32x2*)(zdst + 1L * U) = p;
    p.x = pack_bf2(df[0], df[1]); p.y = pack_bf2(df[2], df[3]); *(u32x2*)(zdst + 2L * U) = p;
    p.x = pack_bf2(dov[0], dov[1]); p.y = pack_bf2(dov[2], dov[3]); *(u32x2*)(zdst + 3L * U) = p;
}

// hprev[d][row(n,t)][U] = h at the step before (n,t) in direction d's own order, 0 at the first step / padding.
// (operand of the W_h weight-gradient GEMM)
__global__ void lstm_hprev_kernel(const bf16_t* __restrict__ hout, const int* __restrict__ seq_len,
                                  bf16_t* __restrict__ hprev, int Nb, int T, int U, int ND) {
    const int groups = U >> 3;
    const long R = (long)Nb * T;
    const long total = (long)ND * R * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int gq = (int)(idx % groups);
        long q = idx / groups;
        long row = q % R;
        int d = (int)(q / R);
        int n = (int)(row / T), t = (int)(row % T);
        int len = min(seq_len[n], T);
        u32x4 v = {0, 0, 0, 0};
        if (t < len) {
            int tp = (d == 0) ? t - 1 : t + 1;
            if (tp >= 0 && tp < len) v = *(const u32x4*)(hout + ((long)n * T + tp) * ((long)ND * U) + (long)d * U + gq * 8);
        }
        *(u32x4*)(hprev + ((long)d * R + row) * U + gq * 8) = v;
    }
}

// xh[d][row(n,t)][D + U] = [x(n,t) | h at the step before (n,t) in direction d's own order]: the operand of ONE weight-gradient
// GEMM per direction for the whole TF LSTMCell matrix [D + U, 4U] (concat([x_t, h_{t-1}]) of network.py:104-107)
__global__ void lstm_xh_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ hout, const int* __restrict__ seq_len,
                               bf16_t* __restrict__ xh, int Nb, int T, int D, int U, int ND) {
    const int groups = (D + U) >> 3, dgroups = D >> 3;
    const long R = (long)Nb * T;
    const long total = (long)ND * R * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int gq = (int)(idx % groups);
        const long q = idx / groups;
        const long row = q % R;
        const int d = (int)(q / R);
        u32x4 v = {0, 0, 0, 0};
        if (gq < dgroups) {
            v = *(const u32x4*)(x + row * D + gq * 8);
        } else {
            const int n = (int)(row / T), t = (int)(row % T);
            const int len = min(seq_len[n], T);
            if (t < len) {
                const int tp = (d == 0) ? t - 1 : t + 1;
                if (tp >= 0 && tp < len) v = *(const u32x4*)(hout + ((long)n * T + tp) * ((long)ND * U) + (long)d * U + (gq - dgroups) * 8);
            }
        }
        *(u32x4*)(xh + ((long)d * R + row) * (D + U) + gq * 8) = v;
    }
}

// packed-column permutation used by the forward operands (see header)
__global__ void lstm_pack_bias_kernel(const float* __restrict__ b_fw, const float* __restrict__ b_bw,
                                      float* __restrict__ out, int U, int ND) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ND * 4 * U) {
        int d = i / (4 * U), c = i % (4 * U);
        int g = c / U, u = c % U;
        int p = (u / 16) * 64 + g * 16 + (u % 16);
        out[d * 4 * U + p] = (d == 0 ? b_fw : b_bw)[c];
    }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int ocr_lstm_fwd_step(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                 float* gates, float* cell, int Nb, int T, int U, int step, float forget_bias,
                                 int ndir, void* stream) {
    if (!xproj || !whT_packed || !seq_len || !hout || !gates || !cell) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || U <= 0 || (U & 31) || step < 0 || step >= T || (ndir != 1 && ndir != 2)) return OCR_ERR_INVALID;
    LstmFwdArgs a = {xproj, (const bf16_t*)whT_packed, seq_len, (bf16_t*)hout, gates, cell, Nb, T, U, step, forget_bias, ndir};
    dim3 grid(U / 16, ndir, ceil_div(Nb, 64));
    lstm_fwd_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_lstm_bwd_step(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                 const float* gates, const float* cell, void* dz, float* dc_state, int Nb, int T,
                                 int U, int step, int ndir, void* stream) {
    if (!wh || !seq_len || !dhout || !gates || !cell || !dz || !dc_state) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || U <= 0 || (U & 31) || step < 0 || step >= T || (ndir != 1 && ndir != 2)) return OCR_ERR_INVALID;
    LstmBwdArgs a = {(const bf16_t*)wh, ldw, w_dir_stride, seq_len, (const bf16_t*)dhout, gates, cell, (bf16_t*)dz,
                     dc_state, Nb, T, U, step, ndir};
    dim3 grid(U / 16, ndir, ceil_div(Nb, 64));
    lstm_bwd_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_lstm_hprev(const void* hout, const int* seq_len, void* hprev, int Nb, int T, int U, int ndir, void* stream) {
    if (!hout || !seq_len || !hprev || (U & 7) || (ndir != 1 && ndir != 2)) return OCR_ERR_INVALID;
    long total = (long)ndir * Nb * T * (U >> 3);
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    lstm_hprev_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)hout, seq_len, (bf16_t*)hprev, Nb, T, U, ndir);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_lstm_xh(const void* x, const void* hout, const int* seq_len, void* xh, int Nb, int T, int D, int U, int ndir, void* stream) {
    if (!x || !hout || !seq_len || !xh || (U & 7) || (D & 7) || (ndir != 1 && ndir != 2)) return OCR_ERR_INVALID;
    long total = (long)ndir * Nb * T * ((D + U) >> 3);
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    lstm_xh_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)hout, seq_len, (bf16_t*)xh, Nb, T, D, U, ndir);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_lstm_pack_bias(const float* b_fw, const float* b_bw, float* out, int U, int ndir, void* stream) {
    if (!b_fw || (ndir == 2 && !b_bw) || !out || (U & 15) || (ndir != 1 && ndir != 2)) return OCR_ERR_INVALID;
    lstm_pack_bias_kernel<<<ceil_div(ndir * 4 * U, 256), 256, 0, (hipStream_t)stream>>>(b_fw, b_bw, out, U, ndir);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
