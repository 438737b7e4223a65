// Whole-sequence (persistent) bidirectional LSTM kernels for gfx950.
//
// Same arithmetic, operand packing and tensor layouts as the per-step kernels in lstm.hip (reference semantics:
// lib/networks/network.py:104-109, TF-1.0 LSTMCell) — but ONE launch walks all T steps, which removes the
// ~10 us dependent-kernel boundary that dominated the step kernels (2 x 63 launches per direction pass):
//   * the workgroup's slice of W_h (forward: 64 packed gate rows x U; backward: 16 unit rows x 4U) is loaded
//     into registers once (128 VGPRs at U = 256) and stays there for the whole sequence;
//   * the cell state c (forward) / the cell gradient (backward) lives in registers;
//   * the only cross-workgroup traffic per step is the 64 x 16-unit slice of h_t (forward) / of dz_t (backward)
//     that the 16 workgroups of a (direction, batch-tile) group exchange.  Hand-off protocol (placement
//     independent, MI355X guide G16/R1): write-through (sc1, agent-scope relaxed atomic) 8-byte payload stores ->
//     every wave drains vmcnt -> __syncthreads -> one lane bumps a monotonic agent-scope counter; consumers:
//     one lane polls the counter relaxed (bounded, s_sleep) -> ONE agent-scope acquire fence -> __syncthreads ->
//     plain 16-byte loads.  Every step writes fresh rows, so there is no write-after-read hazard.
//   * residency: the grid is (U/16) x 2 x ceil(N/64) workgroups of 256 threads, one per CU; the C entry point
//     refuses grids above 256 workgroups (the caller then uses the per-step kernels).  Spins are bounded and
//     report through an error word instead of hanging the device.
#include "common.h"

typedef unsigned long long u64;

#define SPIN_LIMIT (1u << 22)

__device__ __forceinline__ void store_wt8(void* p, u32x2 v) {
    u64 x = ((u64)v.y << 32) | (u64)v.x;
    __hip_atomic_store((u64*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // lowers to a global_store_dwordx2 sc1
}

// all waves: drain own stores, rendezvous, one lane publishes
__device__ __forceinline__ void group_arrive(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until `counter` >= target, then make other workgroups' write-through stores visible to plain loads
__device__ __forceinline__ void group_wait(unsigned* counter, unsigned target, int* err) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { atomicExch(err, 1); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

struct LstmSeqFwdArgs {
    const float* xproj; const bf16_t* whT; const int* seq_len; bf16_t* hout; float* gates; float* cell;
    unsigned* counters; int* err;
    int Nb, T, U; float forget_bias;
};

template <int KS /* U / 32 */>
__global__ __launch_bounds__(256) void lstm_fwd_seq_kernel(LstmSeqFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, d = blockIdx.y, zb = blockIdx.z;
    const int U = KS * 32, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * 64 + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int ul0 = q * 4;
    unsigned* counter = a.counters + (d * gridDim.z + zb);
    const unsigned group = gridDim.x;

    // W_h^T slice: 4 gate fragments x KS k-steps, resident for the whole sequence
    bf16x8 w[4][KS];
    {
        const bf16_t* wbase = a.whT + ((long)d * 4 * U + (long)ub * 64 + nl) * U + q * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) w[g][kk] = *(const bf16x8*)(wbase + (long)g * 16 * U + kk * 32);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < T; ++s) {
        const bool active = nvalid && s < len;
        const int t = active ? (d == 0 ? s : len - 1 - s) : s;
        const int tprev = (d == 0) ? t - 1 : t + 1;
        const long row = (long)nn * T + t;
        const long rowp = (long)nn * T + ((active && s > 0) ? tprev : 0);
        // operands that do not depend on h_{t-1}: issue before waiting for the other workgroups
        const float* xp = a.xproj + row * (8L * U) + (long)d * 4 * U + (long)ub * 64 + ul0;
        f32x4 xi = {0.f, 0.f, 0.f, 0.f}, xj = xi, xf = xi, xo = xi;
        if (active) { xi = *(const f32x4*)(xp + 0); xj = *(const f32x4*)(xp + 16); xf = *(const f32x4*)(xp + 32); xo = *(const f32x4*)(xp + 48); }
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            group_wait(counter, group * (unsigned)s, a.err);
            const bf16_t* hbase = a.hout + rowp * (2L * U) + (long)d * U + q * 8;
            bf16x8 b[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) b[kk] = *(const bf16x8*)(hbase + kk * 32);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][kk], b[kk], acc[g], 0, 0, 0);
        }
        if (nvalid) {
            bf16_t* hdst = a.hout + row * (2L * U) + (long)d * U + ub * 16 + ul0;
            if (!active) {
                u32x2 z = {0u, 0u};
                store_wt8(hdst, z);
            } else {
                f32x4 zi = xi + acc[0], zj = xj + acc[1], zf = xf + acc[2], zo = xo + acc[3];
                f32x4 gi, gj, gf, go, hn;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gi[r] = sigmoidf_(zi[r]);
                    gj[r] = tanhf_(zj[r]);
                    gf[r] = sigmoidf_(zf[r] + a.forget_bias);
                    go[r] = sigmoidf_(zo[r]);
                    c[r] = gf[r] * c[r] + gi[r] * gj[r];
                    hn[r] = go[r] * tanhf_(c[r]);
                }
                u32x2 hp = {pack_bf2(hn[0], hn[1]), pack_bf2(hn[2], hn[3])};
                store_wt8(hdst, hp);
                float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0;
                *(f32x4*)(gdst + 0) = gi; *(f32x4*)(gdst + 16) = gj; *(f32x4*)(gdst + 32) = gf; *(f32x4*)(gdst + 48) = go;
                *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = c;
            }
        }
        if (s + 1 < T) group_arrive(counter);
    }
}

struct LstmSeqBwdArgs {
    const bf16_t* wh; long ldw; long w_dir_stride; const int* seq_len; const bf16_t* dhout; const float* gates;
    const float* cell; bf16_t* dz; unsigned* counters; int* err;
    int Nb, T, U;
};

template <int KS /* 4U / 32 */>
__global__ __launch_bounds__(256) void lstm_bwd_seq_kernel(LstmSeqBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, d = blockIdx.y, zb = blockIdx.z;
    const int U = KS * 8, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * 64 + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int u0 = ub * 16 + q * 4;
    unsigned* counter = a.counters + (d * gridDim.z + zb);
    const unsigned group = gridDim.x;

    bf16x8 w[KS];   // W_h rows (16 units of this workgroup) x K = 4U, resident
    {
        const bf16_t* wbase = a.wh + (long)d * a.w_dir_stride + ((long)ub * 16 + nl) * a.ldw + q * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) w[kk] = *(const bf16x8*)(wbase + kk * 32);
    }
    f32x4 dcs = {0.f, 0.f, 0.f, 0.f};
    for (int s = T - 1, it = 0; s >= 0; --s, ++it) {
        const bool active = nvalid && s < len;
        const bool has_next = nvalid && (s + 1 < len);
        const int t = active ? (d == 0 ? s : len - 1 - s) : s;
        const int tnext = (d == 0) ? t + 1 : t - 1;
        const int tprev = (d == 0) ? t - 1 : t + 1;
        const long row = (long)nn * T + t;
        const long rown = (long)nn * T + (has_next ? tnext : 0);
        // h-independent operands first
        f32x4 gi = {0.f, 0.f, 0.f, 0.f}, gj = gi, gf = gi, go = gi, c = gi, cprev = gi;
        u32x2 g2 = {0u, 0u};
        if (active) {
            const float* gsrc = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + q * 4;
            gi = *(const f32x4*)(gsrc + 0); gj = *(const f32x4*)(gsrc + 16); gf = *(const f32x4*)(gsrc + 32); go = *(const f32x4*)(gsrc + 48);
            c = *(const f32x4*)(a.cell + ((long)d * R + row) * U + u0);
            if (s > 0) cprev = *(const f32x4*)(a.cell + ((long)d * R + (long)nn * T + tprev) * U + u0);
            g2 = *(const u32x2*)(a.dhout + row * (2L * U) + (long)d * U + u0);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        if (it > 0) {
            group_wait(counter, group * (unsigned)it, a.err);
            const bf16_t* zbase = a.dz + rown * (8L * U) + (long)d * 4 * U + q * 8;
            // all K/32 operand loads in flight at once: one L2 round trip per step instead of four
            bf16x8 z[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) z[kk] = *(const bf16x8*)(zbase + kk * 32);
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += 4) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 0], z[k0 + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 1], z[k0 + 1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 2], z[k0 + 2], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 3], z[k0 + 3], acc3, 0, 0, 0);
            }
        }
        if (nvalid) {
            bf16_t* zdst = a.dz + row * (8L * U) + (long)d * 4 * U + u0;
            if (!active) {
                u32x2 z = {0u, 0u};
#pragma unroll
                for (int g = 0; g < 4; ++g) store_wt8(zdst + (long)g * U, z);
            } else {
                f32x4 dh = (acc0 + acc1) + (acc2 + acc3);
                if (!has_next) dh = (f32x4){0.f, 0.f, 0.f, 0.f};
                dh[0] += bf_lo(g2.x); dh[1] += bf_hi(g2.x); dh[2] += bf_lo(g2.y); dh[3] += bf_hi(g2.y);
                f32x4 di, dj, df, dov;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float tc = tanhf_(c[r]);
                    float dc = dcs[r] + dh[r] * go[r] * (1.f - tc * tc);
                    dov[r] = dh[r] * tc * go[r] * (1.f - go[r]);
                    di[r] = dc * gj[r] * gi[r] * (1.f - gi[r]);
                    dj[r] = dc * gi[r] * (1.f - gj[r] * gj[r]);
                    df[r] = dc * cprev[r] * gf[r] * (1.f - gf[r]);
                    dcs[r] = dc * gf[r];
                }
                u32x2 p;
                p.x = pack_bf2(di[0], di[1]); p.y = pack_bf2(di[2], di[3]); store_wt8(zdst + 0L * U, p);
                p.x = pack_bf2(dj[0], dj[1]); p.y = pack_bf2(dj[2], dj[3]); store_wt8(zdst + 1L * U, p);
                p.x = pack_bf2(df[0], df[1]); p.y = pack_bf2(df[2], df[3]); store_wt8(zdst + 2L * U, p);
                p.x = pack_bf2(dov[0], dov[1]); p.y = pack_bf2(dov[2], dov[3]); store_wt8(zdst + 3L * U, p);
            }
        }
        if (s > 0) group_arrive(counter);
    }
}

// ------------------------------------------------------------------------------------------
// C ABI.  `sync` is a caller-owned device block of at least (2 * ceil(Nb/64) + 1) 32-bit words: the group
// counters followed by an error word; it is zeroed on the stream before every launch (hipGraph-replayable).
// Returns OCR_ERR_INVALID for shapes the persistent kernels do not cover (caller falls back to the step kernels).
// ------------------------------------------------------------------------------------------
extern "C" int ocr_lstm_seq_supported(int Nb, int U) {
    if (U != 256) return 0;
    int groups = 2 * ceil_div(Nb, 64);
    return (U / 16) * groups <= 256;
}

extern "C" int ocr_lstm_fwd_seq(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!xproj || !whT_packed || !seq_len || !hout || !gates || !cell || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    int nz = ceil_div(Nb, 64);
    if (hipMemsetAsync(sync, 0, (size_t)(2 * nz + 1) * sizeof(unsigned), stream) != hipSuccess) return OCR_ERR_MEMOPS;
    LstmSeqFwdArgs a = {xproj, (const bf16_t*)whT_packed, seq_len, (bf16_t*)hout, gates, cell, (unsigned*)sync,
                        (int*)sync + 2 * nz, Nb, T, U, forget_bias};
    dim3 grid(U / 16, 2, nz);
    lstm_fwd_seq_kernel<8><<<grid, 256, 0, stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" int ocr_lstm_bwd_seq(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!wh || !seq_len || !dhout || !gates || !cell || !dz || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    int nz = ceil_div(Nb, 64);
    if (hipMemsetAsync(sync, 0, (size_t)(2 * nz + 1) * sizeof(unsigned), stream) != hipSuccess) return OCR_ERR_MEMOPS;
    LstmSeqBwdArgs a = {(const bf16_t*)wh, ldw, w_dir_stride, seq_len, (const bf16_t*)dhout, gates, cell, (bf16_t*)dz,
                        (unsigned*)sync, (int*)sync + 2 * nz, Nb, T, U};
    dim3 grid(U / 16, 2, nz);
    lstm_bwd_seq_kernel<32><<<grid, 256, 0, stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
