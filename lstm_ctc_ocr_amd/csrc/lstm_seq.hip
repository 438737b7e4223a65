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
//     one lane polls the counter relaxed (bounded, s_sleep) -> __syncthreads -> 16-byte `buffer_load ... sc1`
//     loads of the exchanged rows (write-through stores + L1-bypassing loads on both sides: no ~1.7 us
//     buffer_inv acquire per step).  Every step writes fresh rows, so there is no write-after-read hazard.
//   * residency: the grid is (U/16) x 2 x ceil(N/16) one-wave workgroups (or ceil(N/64) four-wave ones), one per CU; the C entry point
//     refuses grids above 256 workgroups (the caller then uses the per-step kernels).  Spins are bounded and
//     report through an error word instead of hanging the device.
#include "common.h"
#include <stdlib.h>

typedef unsigned long long u64;

#define SPIN_LIMIT (1u << 22)
#define CNT_STRIDE 64     // 32-bit words between group counters: pollers of different groups never share a cache line
// optional phase timestamps (wall_clock64, 100 MHz) of workgroup 0: dbg[step*4 + {0 loop top, 1 after wait, 2 after MFMA, 3 after arrive}]
#define DBG_STAMP(slot) do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.dbg[dbgi * 4 + (slot)] = wall_clock64(); } while (0)

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
// same, but only the OLDEST stores have to be drained: `younger` later-issued stores may remain outstanding
__device__ __forceinline__ void group_arrive_after(unsigned* counter, int younger) {
    if (younger == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until `counter` >= target; the exchanged rows are then read with sc1 (L1-bypassing) loads
__device__ __forceinline__ void group_wait(unsigned* counter, unsigned target, int* err) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { atomicExch(err, 1); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ bf16x8 load_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, /*sc1*/ 16);
    return __builtin_bit_cast(bf16x8, v);
}

struct LstmSeqFwdArgs {
    const float* xproj; const bf16_t* whT; const int* seq_len; bf16_t* hout; float* gates; float* cell;
    unsigned* counters; int* err;
    int Nb, T, U; float forget_bias; long long* dbg;
};

template <int KS /* U / 32 */, int WPB /* waves (16-row batch tiles) per workgroup */>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_seq_kernel(LstmSeqFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, d = blockIdx.y, zb = blockIdx.z;
    const int U = KS * 32, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * (16 * WPB) + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int ul0 = q * 4;
    unsigned* counter = a.counters + (d * gridDim.z + zb) * CNT_STRIDE;   // one 256-B line per group counter
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
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.hout, 0, (int)(R * 2 * U * 2), 0x00020000);
    for (int s = 0; s < T; ++s) {
        const int dbgi = s;
        DBG_STAMP(0);
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
            DBG_STAMP(1);
            const unsigned hoff = (unsigned)((rowp * (2L * U) + (long)d * U + q * 8) * 2);
            bf16x8 b[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) b[kk] = load_sc1(hrsrc, hoff + kk * 64);
            __builtin_amdgcn_sched_barrier(0);      // keep ALL loads in flight before the first MFMA: one round trip, not KS
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][kk], b[kk], acc[g], 0, 0, 0);
        }
        DBG_STAMP(2);
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
                store_wt8(hdst, hp);                       // the hand-off payload goes out FIRST ...
                asm volatile("" ::: "memory");
                float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0;
                *(f32x4*)(gdst + 0) = gi; *(f32x4*)(gdst + 16) = gj; *(f32x4*)(gdst + 32) = gf; *(f32x4*)(gdst + 48) = go;
                *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = c;
            }
        }
        // ... so the arrive only has to wait for IT: stores retire in issue order, the five 16-byte saves for the backward
        // pass (gates, cell) may still be in flight when the counter is bumped
        if (s + 1 < T) group_arrive_after(counter, 5);
        DBG_STAMP(3);
    }
}

struct LstmSeqBwdArgs {
    const bf16_t* wh; long ldw; long w_dir_stride; const int* seq_len; const bf16_t* dhout; const float* gates;
    const float* cell; bf16_t* dz; unsigned* counters; int* err;
    int Nb, T, U; long long* dbg;
};

template <int KS /* 4U / 32 */, int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_seq_kernel(LstmSeqBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, d = blockIdx.y, zb = blockIdx.z;
    const int U = KS * 8, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * (16 * WPB) + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int u0 = ub * 16 + q * 4;
    unsigned* counter = a.counters + (d * gridDim.z + zb) * CNT_STRIDE;   // one 256-B line per group counter
    const unsigned group = gridDim.x;

    bf16x8 w[KS];   // W_h rows (16 units of this workgroup) x K = 4U, resident
    {
        const bf16_t* wbase = a.wh + (long)d * a.w_dir_stride + ((long)ub * 16 + nl) * a.ldw + q * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) w[kk] = *(const bf16x8*)(wbase + kk * 32);
    }
    f32x4 dcs = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, (int)(R * 8 * U * 2), 0x00020000);
    for (int s = T - 1, it = 0; s >= 0; --s, ++it) {
        const int dbgi = it;
        DBG_STAMP(0);
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
            DBG_STAMP(1);
            const unsigned zoff = (unsigned)((rown * (8L * U) + (long)d * 4 * U + q * 8) * 2);
            // all K/32 operand loads in flight at once: one L2 round trip per step instead of four
            bf16x8 z[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) z[kk] = load_sc1(zrsrc, zoff + kk * 64);
            __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise ping-pongs two registers: 16 serial sc1 round trips
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += 4) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 0], z[k0 + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 1], z[k0 + 1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 2], z[k0 + 2], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 3], z[k0 + 3], acc3, 0, 0, 0);
            }
        }
        DBG_STAMP(2);
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
        DBG_STAMP(3);
    }
}

// ------------------------------------------------------------------------------------------
// C ABI.  `sync` is a caller-owned device block of at least (2 * ceil(Nb/64) + 1) 32-bit words: the group
// counters followed by an error word; it is zeroed on the stream before every launch (hipGraph-replayable).
// Returns OCR_ERR_INVALID for shapes the persistent kernels do not cover (caller falls back to the step kernels).
// ------------------------------------------------------------------------------------------
// counters are cleared by a kernel, not hipMemsetAsync: under hipGraph replay a memset NODE of these sizes was observed to
// fill the block with a stale non-zero pattern (ROCm 7.2), which makes every wait time out
__global__ void zero_words_kernel(unsigned* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

static long long* g_lstm_dbg = nullptr;
// test/diagnostic hook: device buffer of 4*T int64 receiving wall-clock stamps of workgroup 0 (NULL = off)
extern "C" int ocr_lstm_seq_debug(void* dbg) { g_lstm_dbg = (long long*)dbg; return OCR_OK; }

// batch rows per workgroup: 16 (one wave, 4x the CUs, 4x less hand-off payload per CU and step) while the grid still
// fits one workgroup per CU, else 64
static int seq_rows_per_wg_default(int Nb, int U) {
    for (int rows = 16; rows <= 64; rows *= 2)
        if ((U / 16) * 2 * ceil_div(Nb, rows) <= 256) return rows;
    return 0;
}
static int seq_rows_per_wg(int Nb, int U) {
    // measured at N = 64 (us per step fwd / bwd, group counters on separate cache lines): 16 rows 2.6 / 3.2, 32 rows
    // 3.0 / 4.1, 64 rows 3.6 / 6.2 - the per-step cost is the hand-off payload a CU has to pull (about 65 GB/s per CU)
    // plus a fixed ~2 us of store-drain + counter + poll, so the smallest batch tile that still fits one WG per CU wins.
    const char* e = getenv("OCR_LSTM_ROWS");            // experiment knob: force 16 / 32 / 64 rows per workgroup
    int want = e ? atoi(e) : 0;
    for (int rows = 16; rows <= 64; rows *= 2) {
        if (want && rows != want) continue;
        if ((U / 16) * 2 * ceil_div(Nb, rows) <= 256) return rows;
    }
    if (want) return seq_rows_per_wg_default(Nb, U);
    return 0;
}
extern "C" int ocr_lstm_seq_supported(int Nb, int U) { return U == 256 && seq_rows_per_wg(Nb, U) != 0; }
// int32 words the caller must provide in `sync` (group counters + error word)
extern "C" int ocr_lstm_seq_sync_words(int Nb) { return (2 * ceil_div(Nb, 16) + 1) * CNT_STRIDE; }

extern "C" int ocr_lstm_fwd_seq(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!xproj || !whT_packed || !seq_len || !hout || !gates || !cell || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    const int rows = seq_rows_per_wg(Nb, U);
    const int nz = ceil_div(Nb, rows), words = ocr_lstm_seq_sync_words(Nb);
    zero_words_kernel<<<1, 256, 0, stream>>>((unsigned*)sync, words);
    OCR_CHECK_LAUNCH();
    LstmSeqFwdArgs a = {xproj, (const bf16_t*)whT_packed, seq_len, (bf16_t*)hout, gates, cell, (unsigned*)sync,
                        (int*)sync + words - 1, Nb, T, U, forget_bias, g_lstm_dbg};
    dim3 grid(U / 16, 2, nz);
    if (rows == 16) lstm_fwd_seq_kernel<8, 1><<<grid, 64, 0, stream>>>(a);
    else if (rows == 32) lstm_fwd_seq_kernel<8, 2><<<grid, 128, 0, stream>>>(a);
    else lstm_fwd_seq_kernel<8, 4><<<grid, 256, 0, stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" int ocr_lstm_bwd_seq(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!wh || !seq_len || !dhout || !gates || !cell || !dz || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    const int rows = seq_rows_per_wg(Nb, U);
    const int nz = ceil_div(Nb, rows), words = ocr_lstm_seq_sync_words(Nb);
    zero_words_kernel<<<1, 256, 0, stream>>>((unsigned*)sync, words);
    OCR_CHECK_LAUNCH();
    LstmSeqBwdArgs a = {(const bf16_t*)wh, ldw, w_dir_stride, seq_len, (const bf16_t*)dhout, gates, cell, (bf16_t*)dz,
                        (unsigned*)sync, (int*)sync + words - 1, Nb, T, U, g_lstm_dbg};
    dim3 grid(U / 16, 2, nz);
    if (rows == 16) lstm_bwd_seq_kernel<32, 1><<<grid, 64, 0, stream>>>(a);
    else if (rows == 32) lstm_bwd_seq_kernel<32, 2><<<grid, 128, 0, stream>>>(a);
    else lstm_bwd_seq_kernel<32, 4><<<grid, 256, 0, stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
