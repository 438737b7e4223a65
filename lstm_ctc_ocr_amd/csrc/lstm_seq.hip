// Whole-sequence (persistent) bidirectional LSTM kernels for gfx950.
//
// Same arithmetic, operand packing and tensor layouts as the per-step kernels in lstm.hip (reference semantics:
// lib/networks/network.py:104-109, TF-1.0 LSTMCell) — but ONE launch walks all T steps, which removes the
// ~10 us dependent-kernel boundary that dominated the step kernels (2 x 63 launches per direction pass):
//   * the workgroup's slice of W_h (forward: 64 packed gate rows x U; backward: 16 unit rows x 4U) is loaded
//     into registers once (128 VGPRs at U = 256) and stays there for the whole sequence;
//   * the cell state c (forward) / the cell gradient (backward) lives in registers;
//   * the only cross-workgroup traffic per step is the 64 x 16-unit slice of h_t (forward) / of dz_t (backward)
//     that the 16 workgroups of a (direction, batch-tile) group exchange.  Three hand-off protocols are kept (A/B knob
//     OCR_LSTM_PROTO): 2 (default) = data-as-flag inside one XCD's L2, 1 = data-as-flag through memory (sc1), both
//     described further down, and 0 = counters (placement independent, MI355X guide G16/R1): write-through (sc1, agent-scope relaxed atomic) 8-byte payload stores ->
//     every wave drains vmcnt -> __syncthreads -> one lane bumps a monotonic agent-scope counter; consumers:
//     one lane polls the counter relaxed (bounded, s_sleep) -> __syncthreads -> 16-byte `buffer_load ... sc1`
//     loads of the exchanged rows (write-through stores + L1-bypassing loads on both sides: no ~1.7 us
//     buffer_inv acquire per step).  Every step writes fresh rows, so there is no write-after-read hazard.
//     Measured at N = 64, T = 63 (us per launch, forward / backward): counters 166 / 187, data-as-flag sc1 151 / 203,
//     XCD-local 143 / 170.  What remains per step (~2.2 us) is one L2 write plus one L2 read round trip and the gate math.
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

// ---- data-as-flag hand-off (PROTO 1) -------------------------------------------------------------------------------------
// The exchanged tensor itself carries the "ready" information: it is pre-filled with the bf16 bit pattern 0xFFFF (a NaN
// that sigma(.)*tanh(.) and the gate gradients never produce; a diverged run yields the canonical quiet NaN 0x7FC0/0xFFC0),
// producers publish with write-through stores and move on WITHOUT draining them or bumping a counter, and consumers re-issue
// their L1-bypassing operand loads until no element of the rows they need still holds the fill pattern.  Every 16-bit
// element is checked (packed u16 max), so a partially landed row is simply polled again.  This removes the store drain
// (~0.8 us) and the counter round trip from the per-step critical path of the counter protocol above.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned fold_fill(unsigned m, bf16x8 v) {      // running packed max over the 8 elements of v
    u32x4 w = __builtin_bit_cast(u32x4, v);
    return pk_max_u16(pk_max_u16(m, pk_max_u16(w.x, w.y)), pk_max_u16(w.z, w.w));
}
__device__ __forceinline__ bool holds_fill(unsigned m) { return (m & 0xFFFFu) == 0xFFFFu || (m >> 16) == 0xFFFFu; }
__global__ void fill_words_kernel(unsigned* p, long n, unsigned v, unsigned* zero, int nzero) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < nzero; i += blockDim.x) zero[i] = 0u;
}

// ---- XCD-local hand-off (PROTO 2) -------------------------------------------------------------------------------------
// The 16 workgroups of a (direction, batch-tile) group are placed on ONE XCD (the dispatcher deals workgroup ids round-robin
// over the 8 XCDs: id & 7), so their exchange can stay inside that XCD's L2: plain stores (L1 is write-through) and loads
// that only skip the L1 — no write-through to memory, no memory-side read.  Still data-as-flag, so a placement that is not
// what we assumed cannot produce wrong numbers: the consumer would keep seeing the fill pattern and report a timeout.
// A failed poll costs a whole extra round trip (and its traffic), so each wave sleeps `presleep` x 64 clocks before its first
// poll and adapts that to what it sees: a miss adds two units, eight first-try hits in a row remove one.
#define POLL_ADAPT() do { if (spins > 0) { presleep += 2; streak = 0; } else if (++streak >= 8) { streak = 0; if (presleep > 0) --presleep; } } while (0)
template <int PROTO> __device__ __forceinline__ void store_pub(void* p, u32x2 v) {
    if (PROTO >= 2) *(u32x2*)p = v;
    else store_wt8(p, v);
}
template <int PROTO> __device__ __forceinline__ bf16x8 load_pub(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    // `nt` loads are not kept in the L1 (every poll re-reads the XCD's L2) but, unlike sc1 loads, are served by that L2.
    // Tried and rejected on gfx950: sc0 loads and `buffer_inv sc0` + plain loads both kept hitting the stale L1 line.
    if (PROTO >= 2) return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, /*nt*/ 2));
    return load_sc1(rsrc, byte_off);
}
// workgroup -> (unit block, direction, batch tile); PROTO >= 2: 1-D grid, id = ((g >> 3) * 16 + ub) * 8 + (g & 7), g = zb * 2 + d
template <int PROTO> __device__ __forceinline__ bool seq_decode(int ngroups, int& ub, int& d, int& zb) {
    if (PROTO < 2) { ub = blockIdx.x; d = blockIdx.y; zb = blockIdx.z; return true; }
    const int id = blockIdx.x, r = id >> 3;
    const int g = (r >> 4) * 8 + (id & 7);
    ub = r & 15; d = g & 1; zb = g >> 1;
    return g < ngroups;
}

// ---- in-kernel fill (PROTO 3, experimental: OCR_LSTM_PROTO=3) -------------------------------------------------------------
// As PROTO 2, but no fill kernel in front: every lane stores the fill pattern into the 8 bytes it will publish FILL_AHEAD steps
// later (and, before the first step, into those of steps 0 .. FILL_AHEAD - 1, followed by ONE rendezvous of the group's 16
// workgroups, so that nobody can poll a row that still holds the previous launch's values).  Why: a fill kernel leaves the rows
// in memory (at best the Infinity Cache), so the first poll of every row is an L2 miss and every producer's 32-byte store lands
// in a line the L2 has to complete from memory; filled from inside the XCD a few microseconds ahead, the row is a fully valid L2
// line when producers and consumers touch it (measured background: filling ~0.3 ms EARLIER than the fill kernel does costs
// 16 us forward / 40 us backward — OCR_FUSE_FILLS).  Safety of the running fill: a consumer polls the row of step s - 1
// only after it has seen every producer's row of step s - 2, which that producer stored after (program order) its fill for step
// s - 1 — issued FILL_AHEAD - 1 steps earlier still; same-address stores of one wave stay ordered, so the real row always
// overwrites the fill.
#define FILL_AHEAD 2
// self-resetting rendezvous of `group` workgroups on two words {count, generation}; bounded
__device__ __forceinline__ bool group_rendezvous(unsigned* words, unsigned group, int* err) {
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's fill stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned gen = __hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_fetch_add(words, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == group - 1) {
            __hip_atomic_store(words, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(words + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(words + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { atomicExch(err, 1); ok = 0; break; }
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

struct LstmSeqFwdArgs {
    const float* xproj; const bf16_t* whT; const int* seq_len; bf16_t* hout; float* gates; float* cell;
    unsigned* counters; int* err;
    int Nb, T, U; float forget_bias; long long* dbg; int presleep;
};

template <int KS /* U / 32 */, int WPB /* waves (16-row batch tiles) per workgroup */, int PROTO /* 0 counters, 1 data-as-flag */>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_seq_kernel(LstmSeqFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ub, d, zb;
    const int nzb = (a.Nb + 16 * WPB - 1) / (16 * WPB);
    if (!seq_decode<PROTO>(2 * nzb, ub, d, zb)) return;
    const int U = KS * 32, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * (16 * WPB) + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int ul0 = q * 4;
    unsigned* counter = a.counters + (d * nzb + zb) * CNT_STRIDE;   // one 256-B line per group counter
    const unsigned group = U / 16;

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
    int presleep = a.presleep, streak = 0;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.hout, 0, (int)(R * 2 * U * 2), 0x00020000);
    auto fill_step = [&](int s2) {                     // the 8 bytes this lane publishes at step s2 := fill pattern
        if (!nvalid || s2 >= T) return;
        const int t2 = (s2 < len) ? (d == 0 ? s2 : len - 1 - s2) : s2;
        const u32x2 f = {0xFFFFFFFFu, 0xFFFFFFFFu};
        *(u32x2*)(a.hout + ((long)nn * T + t2) * (2L * U) + (long)d * U + ub * 16 + ul0) = f;
    };
    if (PROTO == 3) {
        for (int s2 = 0; s2 < FILL_AHEAD; ++s2) fill_step(s2);
        if (!group_rendezvous(counter, group, a.err)) return;
    }
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
            const unsigned hoff = (unsigned)((rowp * (2L * U) + (long)d * U + q * 8) * 2);
            bf16x8 b[KS];
            if (PROTO == 0) {
                group_wait(counter, group * (unsigned)s, a.err);
                DBG_STAMP(1);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = load_sc1(hrsrc, hoff + kk * 64);
            } else {
                unsigned spins = 0;
                for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
                while (true) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) b[kk] = load_pub<PROTO>(hrsrc, hoff + kk * 64);
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned m = 0;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, b[kk]);
                    if (!__any(active && holds_fill(m))) break;       // rows past their length read a don't-care row
                    if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); return; }
                }
                DBG_STAMP(1);
                POLL_ADAPT();
                if (!active) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) b[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
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
                store_pub<PROTO>(hdst, z);
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
                store_pub<PROTO>(hdst, hp);                // the hand-off payload goes out FIRST ...
                asm volatile("" ::: "memory");
                float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0;
                *(f32x4*)(gdst + 0) = gi; *(f32x4*)(gdst + 16) = gj; *(f32x4*)(gdst + 32) = gf; *(f32x4*)(gdst + 48) = go;
                *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = c;
            }
        }
        if (PROTO == 3) fill_step(s + FILL_AHEAD);
        // ... so the arrive only has to wait for IT: stores retire in issue order, the five 16-byte saves for the backward
        // pass (gates, cell) may still be in flight when the counter is bumped
        if (PROTO == 0 && s + 1 < T) group_arrive_after(counter, 5);
        DBG_STAMP(3);
    }
}

struct LstmSeqBwdArgs {
    const bf16_t* wh; long ldw; long w_dir_stride; const int* seq_len; const bf16_t* dhout; const float* gates;
    const float* cell; bf16_t* dz; unsigned* counters; int* err;
    int Nb, T, U; long long* dbg; int presleep;
};

template <int KS /* 4U / 32 */, int WPB, int PROTO>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_seq_kernel(LstmSeqBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ub, d, zb;
    const int nzb = (a.Nb + 16 * WPB - 1) / (16 * WPB);
    if (!seq_decode<PROTO>(2 * nzb, ub, d, zb)) return;
    const int U = KS * 8, T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * (16 * WPB) + wave * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int u0 = ub * 16 + q * 4;
    unsigned* counter = a.counters + (d * nzb + zb) * CNT_STRIDE;   // one 256-B line per group counter
    const unsigned group = U / 16;

    bf16x8 w[KS];   // W_h rows (16 units of this workgroup) x K = 4U, resident
    {
        const bf16_t* wbase = a.wh + (long)d * a.w_dir_stride + ((long)ub * 16 + nl) * a.ldw + q * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) w[kk] = *(const bf16x8*)(wbase + kk * 32);
    }
    f32x4 dcs = {0.f, 0.f, 0.f, 0.f};
    int presleep = a.presleep, streak = 0;
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, (int)(R * 8 * U * 2), 0x00020000);
    auto fill_step = [&](int s2) {                     // the 4 x 8 bytes this lane publishes at step s2 := fill pattern
        if (!nvalid || s2 < 0) return;
        const int t2 = (s2 < len) ? (d == 0 ? s2 : len - 1 - s2) : s2;
        const u32x2 f = {0xFFFFFFFFu, 0xFFFFFFFFu};
        bf16_t* p = a.dz + ((long)nn * T + t2) * (8L * U) + (long)d * 4 * U + u0;
#pragma unroll
        for (int g = 0; g < 4; ++g) *(u32x2*)(p + (long)g * U) = f;
    };
    if (PROTO == 3) {
        for (int k = 0; k < FILL_AHEAD; ++k) fill_step(T - 1 - k);
        if (!group_rendezvous(counter, group, a.err)) return;
    }
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
            const unsigned zoff = (unsigned)((rown * (8L * U) + (long)d * 4 * U + q * 8) * 2);
            // all K/32 operand loads in flight at once: one L2 round trip per step instead of four
            bf16x8 z[KS];
            if (PROTO == 0) {
                group_wait(counter, group * (unsigned)it, a.err);
                DBG_STAMP(1);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) z[kk] = load_sc1(zrsrc, zoff + kk * 64);
            } else {
                unsigned spins = 0;
                for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
                while (true) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) z[kk] = load_pub<PROTO>(zrsrc, zoff + kk * 64);
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned m = 0;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, z[kk]);
                    if (!__any(has_next && holds_fill(m))) break;     // rows without a successor step read a don't-care row
                    if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); return; }
                }
                DBG_STAMP(1);
                POLL_ADAPT();
                if (!has_next) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) z[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
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
                for (int g = 0; g < 4; ++g) store_pub<PROTO>(zdst + (long)g * U, z);
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
                p.x = pack_bf2(di[0], di[1]); p.y = pack_bf2(di[2], di[3]); store_pub<PROTO>(zdst + 0L * U, p);
                p.x = pack_bf2(dj[0], dj[1]); p.y = pack_bf2(dj[2], dj[3]); store_pub<PROTO>(zdst + 1L * U, p);
                p.x = pack_bf2(df[0], df[1]); p.y = pack_bf2(df[2], df[3]); store_pub<PROTO>(zdst + 2L * U, p);
                p.x = pack_bf2(dov[0], dov[1]); p.y = pack_bf2(dov[2], dov[3]); store_pub<PROTO>(zdst + 3L * U, p);
            }
        }
        if (PROTO == 3) fill_step(s - FILL_AHEAD);
        if (PROTO == 0 && s > 0) group_arrive(counter);
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
// environment knobs are looked up ONCE per process (the launch path runs every step when graphs are off)
static int seq_env_int(const char* name, int slot, int dflt) {
    static int val[2], seen[2];
    if (!seen[slot]) { const char* e = getenv(name); val[slot] = e ? atoi(e) : dflt; seen[slot] = 1; }
    return val[slot];
}
static int seq_rows_per_wg(int Nb, int U) {
    // measured at N = 64 (us per step fwd / bwd, group counters on separate cache lines): 16 rows 2.6 / 3.2, 32 rows
    // 3.0 / 4.1, 64 rows 3.6 / 6.2 - the per-step cost is the hand-off payload a CU has to pull (about 65 GB/s per CU)
    // plus a fixed ~2 us of store-drain + counter + poll, so the smallest batch tile that still fits one WG per CU wins.
    static int want = -1;                               // experiment knob, read once: force 16 / 32 / 64 rows per workgroup
    if (want < 0) { const char* e = getenv("OCR_LSTM_ROWS"); want = e ? atoi(e) : 0; if (want < 0) want = 0; }
    for (int rows = 16; rows <= 64; rows *= 2) {
        if (want && rows != want) continue;
        if ((U / 16) * 2 * ceil_div(Nb, rows) <= 256) return rows;
    }
    if (want) return seq_rows_per_wg_default(Nb, U);
    return 0;
}
static int g_seq_proto = -1;
// hand-off protocol: 0 counters (sc1), 1 data-as-flag (sc1), 2 data-as-flag inside one XCD (default).  Environment
// OCR_LSTM_PROTO (A/B) wins over the setter, which the host uses to fall back to 1 when the device does not co-locate
// workgroups with equal (id & 7) on one XCD (checked once with ocr_probe_xcc).
extern "C" int ocr_set_lstm_proto(int proto) {
    if (proto < 0 || proto > 3) return OCR_ERR_INVALID;
    g_seq_proto = proto;
    return OCR_OK;
}
static int seq_proto() {
    static int env = -2;
    if (env == -2) { const char* e = getenv("OCR_LSTM_PROTO"); env = e ? atoi(e) : -1; if (env < -1 || env > 3) env = -1; }
    if (env >= 0) return env;
    return g_seq_proto >= 0 ? g_seq_proto : 2;
}
extern "C" int ocr_lstm_seq_supported(int Nb, int U) { return U == 256 && seq_rows_per_wg(Nb, U) != 0; }
// int32 words the caller must provide in `sync` (group counters + error word)
extern "C" int ocr_lstm_seq_sync_words(int Nb) { return (2 * ceil_div(Nb, 16) + 1) * CNT_STRIDE; }

static int lstm_fwd_seq_(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                         float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                         void* stream_, bool prefilled) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!xproj || !whT_packed || !seq_len || !hout || !gates || !cell || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    const int rows = seq_rows_per_wg(Nb, U);
    const int nz = ceil_div(Nb, rows), words = ocr_lstm_seq_sync_words(Nb);
    const int proto = seq_proto();
    if (prefilled || proto == 3) {}                     // the caller's own fill pass did both (ocr_fill_jobs) / the kernel fills as it goes
    else if (proto == 0) zero_words_kernel<<<1, 256, 0, stream>>>((unsigned*)sync, words);
    else fill_words_kernel<<<256, 256, 0, stream>>>((unsigned*)hout, (long)Nb * T * 2 * U / 2, 0xFFFFFFFFu, (unsigned*)sync, words);
    OCR_CHECK_LAUNCH();
    LstmSeqFwdArgs a = {xproj, (const bf16_t*)whT_packed, seq_len, (bf16_t*)hout, gates, cell, (unsigned*)sync,
                        (int*)sync + words - 1, Nb, T, U, forget_bias, g_lstm_dbg, seq_env_int("OCR_LSTM_PRESLEEP_F", 0, 0)};
    dim3 grid(U / 16, 2, nz);
    const dim3 g1(16 * 8 * ceil_div(2 * nz, 8));
#define LAUNCH_FWD(P, G) do { if (rows == 16) lstm_fwd_seq_kernel<8, 1, P><<<G, 64, 0, stream>>>(a); \
        else if (rows == 32) lstm_fwd_seq_kernel<8, 2, P><<<G, 128, 0, stream>>>(a); \
        else lstm_fwd_seq_kernel<8, 4, P><<<G, 256, 0, stream>>>(a); } while (0)
    if (proto == 0) LAUNCH_FWD(0, grid);
    else if (proto == 1) LAUNCH_FWD(1, grid);
    else if (proto == 3) LAUNCH_FWD(3, g1);
    else LAUNCH_FWD(2, g1);
#undef LAUNCH_FWD
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_lstm_fwd_seq(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                                void* stream) {
    return lstm_fwd_seq_(xproj, whT_packed, seq_len, hout, gates, cell, Nb, T, U, forget_bias, sync, stream, false);
}
// The same without the call's own fill launch: the caller has, since the last use of these buffers, stored 0xFFFF into every 16-bit
// element of hout [Nb*T][2U] and zero into all ocr_lstm_seq_sync_words(Nb) words of sync (a training step does all its fills — this
// one, the backward one, the gradient buffer — in ONE launch at its start: ocr_fill_jobs)
extern "C" int ocr_lstm_fwd_seq_prefilled(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                          float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                                          void* stream) {
    return lstm_fwd_seq_(xproj, whT_packed, seq_len, hout, gates, cell, Nb, T, U, forget_bias, sync, stream, true);
}

static int lstm_bwd_seq_(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                         const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                         void* stream_, bool prefilled) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!wh || !seq_len || !dhout || !gates || !cell || !dz || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    const int rows = seq_rows_per_wg(Nb, U);
    const int nz = ceil_div(Nb, rows), words = ocr_lstm_seq_sync_words(Nb);
    const int proto = seq_proto();
    if (prefilled || proto == 3) {}
    else if (proto == 0) zero_words_kernel<<<1, 256, 0, stream>>>((unsigned*)sync, words);
    else fill_words_kernel<<<256, 256, 0, stream>>>((unsigned*)dz, (long)Nb * T * 8 * U / 2, 0xFFFFFFFFu, (unsigned*)sync, words);
    OCR_CHECK_LAUNCH();
    LstmSeqBwdArgs a = {(const bf16_t*)wh, ldw, w_dir_stride, seq_len, (const bf16_t*)dhout, gates, cell, (bf16_t*)dz,
                        (unsigned*)sync, (int*)sync + words - 1, Nb, T, U, g_lstm_dbg, seq_env_int("OCR_LSTM_PRESLEEP", 1, 4)};
    dim3 grid(U / 16, 2, nz);
    const dim3 g1(16 * 8 * ceil_div(2 * nz, 8));
#define LAUNCH_BWD(P, G) do { if (rows == 16) lstm_bwd_seq_kernel<32, 1, P><<<G, 64, 0, stream>>>(a); \
        else if (rows == 32) lstm_bwd_seq_kernel<32, 2, P><<<G, 128, 0, stream>>>(a); \
        else lstm_bwd_seq_kernel<32, 4, P><<<G, 256, 0, stream>>>(a); } while (0)
    if (proto == 0) LAUNCH_BWD(0, grid);
    else if (proto == 1) LAUNCH_BWD(1, grid);
    else if (proto == 3) LAUNCH_BWD(3, g1);
    else LAUNCH_BWD(2, g1);
#undef LAUNCH_BWD
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
extern "C" int ocr_lstm_bwd_seq(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                                void* stream) {
    return lstm_bwd_seq_(wh, ldw, w_dir_stride, seq_len, dhout, gates, cell, dz, Nb, T, U, sync, stream, false);
}
// as ocr_lstm_fwd_seq_prefilled: the caller has stored 0xFFFF into every 16-bit element of dz [Nb*T][8U] and zero into all sync words
extern "C" int ocr_lstm_bwd_seq_prefilled(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                          const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                                          void* stream) {
    return lstm_bwd_seq_(wh, ldw, w_dir_stride, seq_len, dhout, gates, cell, dz, Nb, T, U, sync, stream, true);
}
