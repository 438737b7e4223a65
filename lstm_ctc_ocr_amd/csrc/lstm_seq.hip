// Whole-sequence (persistent) bidirectional LSTM kernels for gfx950.
//
// Same arithmetic, operand packing and tensor layouts as the per-step kernels in lstm.hip (reference semantics:
// lib/networks/network.py:104-109, TF-1.0 LSTMCell) — but ONE launch walks all T steps, which removes the
// ~10 us dependent-kernel boundary that dominated the step kernels (2 x 63 launches per direction pass):
//   * a workgroup owns 16 hidden units of one (direction, batch tile).  Its slice of W_h (forward: 64 packed gate rows x U;
//     backward: 16 unit rows x 4U) is loaded into registers once and stays there for the whole sequence: 128 VGPRs per wave
//     at U = 256; at U = 512 (BASELINE configs[4]: 2 x BiLSTM(512)) the slice is 256 VGPRs, so the contraction axis is split
//     over TWO waves of the workgroup (KSP = 2, 128 VGPRs each) whose partial sums meet in LDS once per step;
//   * the cell state c (forward) / the cell gradient (backward) lives in registers;
//   * the only cross-workgroup traffic per step is the slice of h_t (forward) / of dz_t (backward) that the U/16 workgroups
//     of a (direction, batch-tile) group exchange.  Hand-off protocols (ocr_set_lstm_proto / OCR_LSTM_PROTO):
//       0  counters (placement independent, MI355X guide G16/R1): write-through (sc1) payload stores -> every wave drains
//          vmcnt -> __syncthreads -> one lane bumps a monotonic agent-scope counter; consumers poll it relaxed, then read the
//          rows with sc1 loads.  Used when the device does not co-locate workgroups as protocol 4 assumes (ocr_probe_xcc).
//       4  (default, round 3) data-as-flag through a small RING in one XCD's L2: see below.  (Round 2's protocol 2 exchanged through
//          the output tensor itself: 141 / 170 us forward / backward at N = 64, U = 256 against 138 / 133 with the ring, 158 / 262
//          against 150 / 162 at U = 512 — removed.)
//   * residency: the grid is (U/16) x 2 x ceil(N/rows) workgroups, at most one per CU (the C entry point refuses larger grids;
//     the caller then uses the per-step kernels).  Spins are bounded and report through an error word instead of hanging.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef unsigned long long u64;

#define SPIN_LIMIT (1u << 22)
#define CNT_STRIDE 64     // 32-bit words between group counters: pollers of different groups never share a cache line
#define RING 4            // slots of the protocol-4 exchange ring (see there why 4)
// optional phase timestamps (wall_clock64, 100 MHz) of workgroup 0: dbg[step*4 + {0 loop top, 1 after wait, 2 after MFMA, 3 after arrive}]
#define DBG_STAMP(slot) do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.dbg[dbgi * 4 + (slot)] = wall_clock64(); } while (0)
// Test hooks.  SKEW_HOOK: ocr_lstm_seq_test_skew(n, at) makes unit block 0 of EVERY group sleep n x 64 clocks at the top of iteration `at` (at < 0:
// of every iteration) — the skew between workgroups that HBM contention provides once in ~10^4 launches, made deterministic
// (tests/test_gpu_stress.py: the hand-off must be right for ANY relative speed of the workgroups).  A delay in EVERY iteration is absorbed by the
// other workgroups' adaptive pre-poll sleep (they slow down with it, free iterations included — measured: the round-4 rule survives it); the delay at
// ONE iteration (the tile's last active step forward, the first iteration backward) is what lets the others run ahead.  The four-wave kernels carry the
// hook in instances of their own (template parameter SKEW, launched only while a skew is set): compiled into the product instance the extra
// block boundary at the top of every step cost the backward kernel 4 us per launch (95.9 against 91.5 us, profiles/r06a_kernel_stats.md).  RING_LIVE: the round-5 rule (a workgroup none of whose rows is inside its sequence leaves the
// ring alone); the experiments build can be told to follow the rule of rounds 3-4 again (OCR_LSTM_RING_RULE=always: every iteration stores and
// refills) so that the same tests can be shown to FAIL on it — the product library has no such switch.
#define SKEW_HOOK() do { if (a.skew && ub == 0 && (a.skew_at < 0 || a.skew_at == dbgi)) for (int i_ = 0; i_ < a.skew; ++i_) __builtin_amdgcn_s_sleep(1); } while (0)
#ifdef OCR_EXPERIMENTS
#define RING_LIVE(act) (__any(act) || a.ring_always)
#else
#define RING_LIVE(act) (__any(act))
#endif

__device__ __forceinline__ void store_wt8(void* p, u32x2 v) {
    u64 x = ((u64)v.y << 32) | (u64)v.x;
    __hip_atomic_store((u64*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // lowers to a global_store_dwordx2 sc1
}

// ---- counters (PROTO 0) ----------------------------------------------------------------------------------------------
// all waves: drain own stores (only the OLDEST have to be drained: `younger` later-issued stores may remain outstanding),
// rendezvous, one lane publishes
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

// ---- data-as-flag hand-off (PROTO 4) ------------------------------------------------------------------------------------
// The exchanged rows themselves carry the "ready" information: they start as the bf16 bit pattern 0xFFFF (a NaN that
// sigma(.)*tanh(.) and the gate gradients never produce; a diverged run yields the canonical quiet NaN 0x7FC0/0xFFC0),
// producers publish with plain stores and move on WITHOUT draining them or bumping a counter, and consumers re-issue their
// operand loads until no element of the rows they need still holds the fill pattern.  Every 16-bit element is checked
// (packed u16 max), so a partially landed row is simply polled again.
// The U/16 workgroups of a (direction, batch-tile) group are placed on ONE XCD (the dispatcher deals workgroup ids round-robin
// over the 8 XCDs: id & 7; checked once per process with ocr_probe_xcc), so the exchange stays inside that XCD's L2: plain
// stores (the L1 is write-through) and `nt` loads, which skip the L1 but are served by the L2 (sc0 loads and `buffer_inv sc0`
// both kept hitting the stale L1 line).  A placement that is not what we assumed cannot produce wrong numbers: the consumer would
// keep seeing the fill pattern and report a time-out.
// A failed poll costs a whole extra round trip (and its traffic), so each wave sleeps `presleep` x 64 clocks before its first
// poll and adapts that to what it sees: a miss adds two units, eight first-try hits in a row remove one.
//
// Round 2 exchanged through the layer's own output tensor: every step touched rows nobody had touched since the fill kernel
// (4 MB of 0xFFFF per launch at N = 64, T = 63), the first poll of every row came from the Infinity Cache at best, and the
// 32-byte pieces of the 16 producers landed in cache lines that the L2 had to complete (filling 0.3 ms earlier, i.e. rows in HBM
// instead of the Infinity Cache, cost 16 us forward / 40 us backward — the recurrence is that sensitive to where its flag rows live).
// Now the exchange goes through a ring of RING = 4 step slots per group, [slot][unit block][batch sub-tile]([gate])[16 rows][16 units]:
//   * a producer wave's piece is 512 B (forward) / 2 KB (backward) of whole cache lines, written by nobody else;
//   * the ring (32 KB forward / 128 KB backward per group at U = 256) never leaves the XCD's L2: every poll is an L2 hit;
//   * slots are recycled by their producers: at step s, after its own poll for step s has succeeded — every workgroup of the group
//     has then finished step s - 1, so nobody will read slot s - 2 again — a producer stores the fill pattern over its own piece of
//     slot s - 2, which is next written at step s + 2 and next polled at step s + 3.  Every wave drains its stores (s_waitcnt
//     vmcnt(0), while it would be waiting for the others anyway) at the top of each step, so whoever has seen a producer's
//     rows of step s + 1 knows that its refill of step s has landed: a consumer can never take old rows for new ones;
//   * the layer's real output rows (hout / dz) are written behind the ring piece with plain stores nobody waits for, and need no
//     fill: the per-launch fill shrinks from 4 MB / 16 MB to the ring (256 KB / 1 MB).
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
#define POLL_ADAPT() do { if (spins > 0) { presleep += 2; streak = 0; } else if (++streak >= 8) { streak = 0; if (presleep > 0) --presleep; } } while (0)
template <int PROTO> __device__ __forceinline__ bf16x8 load_pub(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned uniform_off = 0) {
    if (PROTO >= 2) return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)uniform_off, /*nt*/ 2));
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)uniform_off, /*sc1*/ 16));
}
// workgroup -> (unit block, direction, batch tile); PROTO >= 2: 1-D grid, id = ((g >> 3) * UB + ub) * 8 + (g & 7), g = zb * 2 + d:
// all UB workgroups of group g have id & 7 == g & 7
template <int PROTO, int UB> __device__ __forceinline__ bool seq_decode(int ngroups, int& ub, int& d, int& zb) {
    if (PROTO < 2) { ub = blockIdx.x; d = blockIdx.y; zb = blockIdx.z; return true; }
    const int id = blockIdx.x, r = id >> 3;
    const int g = (r / UB) * 8 + (id & 7);
    ub = r % UB; d = g & 1; zb = g >> 1;
    return g < ngroups;
}

struct LstmSeqFwdArgs {
    const float* xproj; const bf16_t* whT; const int* seq_len; bf16_t* hout; float* gates; float* cell;
    unsigned* counters; unsigned char* ring; int* err;
    int Nb, T; float forget_bias; long long* dbg; int presleep; int skew; int skew_at; int ring_always;
};

// U: hidden units per direction; WPB: batch sub-tiles (waves) per workgroup; KSP: waves that share the contraction axis of one
// sub-tile (1, or 2 at U = 512).  (8-row sub-tiles — twice the groups, half the hand-off payload per CU — were measured and dropped:
// 146 / 154 us against 138 / 133 at N = 64, U = 256.)
template <int U, int WPB, int KSP, int PROTO>
__global__ __launch_bounds__(64 * WPB * KSP) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_seq_kernel(LstmSeqFwdArgs a) {
    constexpr int RPW = 16, KS = U / 32 / KSP, UB = U / 16, ROWS = RPW * WPB;
    constexpr unsigned SLOT = (unsigned)UB * WPB * 512u;                 // ring bytes per step and group
    __shared__ f32x4 red[KSP == 2 ? 2 : 1][WPB][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = wave / WPB, bw = wave % WPB;
    int ub, d, zb;
    const int nzb = (a.Nb + ROWS - 1) / ROWS;
    if (!seq_decode<PROTO, UB>(2 * nzb, ub, d, zb)) return;
    const int T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * ROWS + bw * RPW + nl;
    const bool nvalid = nl < RPW && n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int ul0 = q * 4;
    unsigned* counter = a.counters + (d * nzb + zb) * CNT_STRIDE;   // one 256-B line per group counter
    const unsigned group = UB;

    // W_h^T slice: 4 gate fragments x KS k-steps of this wave's part of the contraction axis, resident for the whole sequence
    bf16x8 w[4][KS];
    {
        const bf16_t* wbase = a.whT + ((long)d * 4 * U + (long)ub * 64 + nl) * U + kh * (KS * 32) + q * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) w[g][kk] = *(const bf16x8*)(wbase + (long)g * 16 * U + kk * 32);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    int presleep = a.presleep, streak = 0;
    bool dead = false;                                 // a poll of this wave has timed out: error word set, no further polling
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.hout, 0, (int)(R * 2 * U * 2), 0x00020000);
    // ring (PROTO 4): this group's RING slots; producer piece = ((ub * WPB + bw) * 16 + nl) * 32 + q * 8, the operand of k-step kk is
    // unit block 2 (kh KS + kk) + (q >> 1), 16-byte half (q & 1) of the same row
    unsigned char* const gring = a.ring + (size_t)(zb * 2 + d) * RING * SLOT;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gring, 0, (int)(RING * SLOT), 0x00020000);
    const unsigned rd0 = (unsigned)(((kh * KS * 2 + (q >> 1)) * WPB + bw) * 512 + nl * 32 + (q & 1) * 16);
    const unsigned wr0 = (unsigned)((ub * WPB + bw) * 512 + nl * 32 + q * 8);
    for (int s = 0; s < T; ++s) {
        const int dbgi = s;
        DBG_STAMP(0);
        SKEW_HOOK();
        // last step's slot refill (and everything older) has landed: stores retire in issue order, the six issued behind it may stay in flight
        if (PROTO == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        const bool active = nvalid && s < len;
        const int t = active ? (d == 0 ? s : len - 1 - s) : s;
        const int tprev = (d == 0) ? t - 1 : t + 1;
        const long row = (long)nn * T + t;
        const long rowp = (long)nn * T + ((active && s > 0) ? tprev : 0);
        // operands that do not depend on h_{t-1}: issue before waiting for the other workgroups
        const float* xp = a.xproj + row * (8L * U) + (long)d * 4 * U + (long)ub * 64 + ul0;
        f32x4 xi = {0.f, 0.f, 0.f, 0.f}, xj = xi, xf = xi, xo = xi;
        if (active && kh == 0) { xi = *(const f32x4*)(xp + 0); xj = *(const f32x4*)(xp + 16); xf = *(const f32x4*)(xp + 32); xo = *(const f32x4*)(xp + 48); }
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const unsigned hoff = (PROTO == 4) ? (unsigned)((s - 1) & (RING - 1)) * SLOT + rd0
                                               : (unsigned)((rowp * (2L * U) + (long)d * U + kh * (KS * 32) + q * 8) * 2);
            constexpr unsigned kstride = (PROTO == 4) ? WPB * 1024u : 64u;
            const __amdgpu_buffer_rsrc_t rs = (PROTO == 4) ? rrsrc : hrsrc;
            bf16x8 b[KS];
            if (PROTO == 0) {
                group_wait(counter, group * (unsigned)s, a.err);
                DBG_STAMP(1);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = load_pub<0>(rs, hoff + kk * kstride);
            } else {
                unsigned spins = 0;
                for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
                while (true) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) b[kk] = load_pub<PROTO>(rs, hoff + kk * kstride);
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned m = 0;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, b[kk]);
                    if (dead || !__any(active && holds_fill(m))) break;       // rows past their length read a don't-care row
                    if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); dead = true; break; }
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
            if (KSP == 2) {                         // partial sums of the upper half of the contraction axis -> the wave that owns the tail
                if (kh == 1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) red[s & 1][bw][g][lane] = acc[g];
                }
                __syncthreads();
                if (kh == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] += red[s & 1][bw][g][lane];
                }
            }
        }
        DBG_STAMP(2);
        if (nvalid && kh == 0) {
            bf16_t* hdst = a.hout + row * (2L * U) + (long)d * U + ub * 16 + ul0;
            u32x2* rdst = (u32x2*)(gring + (unsigned)(s & (RING - 1)) * SLOT + wr0);
            if (PROTO == 4) {
                // One store sequence for every lane (rows past their length store zeros / throw-away gate values into rows nobody reads):
                // ring piece, slot refill, output row, the five saves for the backward pass.  The wave then knows how many stores are
                // younger than its refill, and the top of the next step waits for exactly the others (vmcnt(6)) instead of for all of them
                // — the 16-byte saves took ~0.4 us of every step's critical path while it waited for vmcnt(0).
                f32x4 zi = xi + acc[0], zj = xj + acc[1], zf = xf + acc[2], zo = xo + acc[3];
                f32x4 gi, gj, gf, go, hn, cn;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gi[r] = sigmoidf_(zi[r]);
                    gj[r] = tanhf_(zj[r]);
                    gf[r] = sigmoidf_(zf[r] + a.forget_bias);
                    go[r] = sigmoidf_(zo[r]);
                    cn[r] = gf[r] * c[r] + gi[r] * gj[r];
                    hn[r] = go[r] * tanhf_(cn[r]);
                }
                if (active) c = cn;
                u32x2 hp = {pack_bf2(hn[0], hn[1]), pack_bf2(hn[2], hn[3])};
                if (!active) hp = (u32x2){0u, 0u};
                // (round 5) a workgroup whose rows are ALL past their length touches the ring no more: it polls nothing, so it runs free, and
                // its stores would land in slots that a slower workgroup — still at the tile's last active step — is reading: the slot refill
                // below destroyed the piece that workgroup was polling for (a time-out seen ~3 times in 30 000 live-pipeline iterations at
                // W = 88, where every sequence is one step shorter than T), and three free steps later the payload of an inactive step (zeros)
                // would overwrite the h it still needs.  Nothing polls these slots again in this launch; the next launch refills the ring.
                const bool ring_live = RING_LIVE(active);
                if (ring_live) *rdst = hp;                  // the hand-off payload goes out FIRST
                asm volatile("" ::: "memory");
                if (s >= 2 && ring_live) {                  // recycle this wave's piece of the slot that nobody reads any more
                    const u32x2 f = {0xFFFFFFFFu, 0xFFFFFFFFu};
                    *(u32x2*)(gring + (unsigned)((s - 2) & (RING - 1)) * SLOT + wr0) = f;
                }
                asm volatile("" ::: "memory");
                *(u32x2*)hdst = hp;
                float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0;
                *(f32x4*)(gdst + 0) = gi; *(f32x4*)(gdst + 16) = gj; *(f32x4*)(gdst + 32) = gf; *(f32x4*)(gdst + 48) = go;
                *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = c;
            } else if (!active) {
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
        if (PROTO == 0 && s + 1 < T) group_arrive_after(counter, 5);
        DBG_STAMP(3);
    }
}

struct LstmSeqBwdArgs {
    const bf16_t* wh; long ldw; long w_dir_stride; const int* seq_len; const bf16_t* dhout; const float* gates;
    const float* cell; bf16_t* dz; unsigned* counters; unsigned char* ring; int* err;
    int Nb, T; long long* dbg; int presleep; int skew; int skew_at; int ring_always;
};

template <int U, int WPB, int KSP, int PROTO>
__global__ __launch_bounds__(64 * WPB * KSP) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_seq_kernel(LstmSeqBwdArgs a) {
    constexpr int RPW = 16, KS = 4 * U / 32 / KSP, UB = U / 16, ROWS = RPW * WPB;
    constexpr unsigned SLOT = (unsigned)UB * WPB * 2048u;                // ring bytes per step and group: [ub][bw][gate][16 rows][16 units]
    __shared__ f32x4 red[KSP == 2 ? 2 : 1][WPB][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = wave / WPB, bw = wave % WPB;
    int ub, d, zb;
    const int nzb = (a.Nb + ROWS - 1) / ROWS;
    if (!seq_decode<PROTO, UB>(2 * nzb, ub, d, zb)) return;
    const int T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * ROWS + bw * RPW + nl;
    const bool nvalid = nl < RPW && n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int u0 = ub * 16 + q * 4;
    unsigned* counter = a.counters + (d * nzb + zb) * CNT_STRIDE;   // one 256-B line per group counter
    const unsigned group = UB;

    bf16x8 w[KS];   // W_h rows (16 units of this workgroup) x this wave's part of K = 4U, resident
    {
        const bf16_t* wbase = a.wh + (long)d * a.w_dir_stride + ((long)ub * 16 + nl) * a.ldw + kh * (KS * 32) + q * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) w[kk] = *(const bf16x8*)(wbase + kk * 32);
    }
    f32x4 dcs = {0.f, 0.f, 0.f, 0.f};
    int presleep = a.presleep, streak = 0;
    bool dead = false;
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, (int)(R * 8 * U * 2), 0x00020000);
    unsigned char* const gring = a.ring + (size_t)(zb * 2 + d) * RING * SLOT;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gring, 0, (int)(RING * SLOT), 0x00020000);
    // master gate column k = g U + u of k-step kk (this wave): g = K0 / U, unit block (K0 % U) / 16 + (q >> 1) with K0 = (kh KS + kk) 32
    const unsigned rdl = (unsigned)((q >> 1) * WPB * 2048 + bw * 2048 + nl * 32 + (q & 1) * 16);
    const unsigned wr0 = (unsigned)((ub * WPB + bw) * 2048 + nl * 32 + q * 8);     // + gate * 512
    for (int s = T - 1, it = 0; s >= 0; --s, ++it) {
        const int dbgi = it;
        DBG_STAMP(0);
        SKEW_HOOK();
        if (PROTO == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // as in the forward kernel: the four dz stores sit behind the refill
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
        if (active && kh == 0) {
            const float* gsrc = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + q * 4;
            gi = *(const f32x4*)(gsrc + 0); gj = *(const f32x4*)(gsrc + 16); gf = *(const f32x4*)(gsrc + 32); go = *(const f32x4*)(gsrc + 48);
            c = *(const f32x4*)(a.cell + ((long)d * R + row) * U + u0);
            if (s > 0) cprev = *(const f32x4*)(a.cell + ((long)d * R + (long)nn * T + tprev) * U + u0);
            g2 = *(const u32x2*)(a.dhout + row * (2L * U) + (long)d * U + u0);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        if (it > 0) {
            const unsigned zoff = (unsigned)((rown * (8L * U) + (long)d * 4 * U + kh * (KS * 32) + q * 8) * 2);
            const unsigned roff = (unsigned)((it - 1) & (RING - 1)) * SLOT + rdl;
            // all K/32 operand loads in flight at once: one L2 round trip per step instead of four
            bf16x8 z[KS];
            auto issue = [&]() {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    if (PROTO == 4) {
                        const int K0 = (kh * KS + kk) * 32;
                        z[kk] = load_pub<4>(rrsrc, roff, (unsigned)(((K0 % U) / 16) * WPB * 2048 + (K0 / U) * 512));
                    } else z[kk] = load_pub<0>(zrsrc, zoff + kk * 64);
                }
            };
            if (PROTO == 0) {
                group_wait(counter, group * (unsigned)it, a.err);
                DBG_STAMP(1);
                issue();
            } else {
                unsigned spins = 0;
                for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
                while (true) {
                    issue();
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned m = 0;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, z[kk]);
                    if (dead || !__any(has_next && holds_fill(m))) break;     // rows without a successor step read a don't-care row
                    if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); dead = true; break; }
                }
                DBG_STAMP(1);
                POLL_ADAPT();
                if (!has_next) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) z[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise ping-pongs two registers: 16 serial round trips
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += 4) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 0], z[k0 + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 1], z[k0 + 1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 2], z[k0 + 2], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 3], z[k0 + 3], acc3, 0, 0, 0);
            }
        }
        f32x4 dh = (acc0 + acc1) + (acc2 + acc3);
        if (KSP == 2 && it > 0) {
            if (kh == 1) red[it & 1][bw][lane] = dh;
            __syncthreads();
            if (kh == 0) dh += red[it & 1][bw][lane];
        }
        DBG_STAMP(2);
        if (nvalid && kh == 0) {
            bf16_t* zdst = a.dz + row * (8L * U) + (long)d * 4 * U + u0;
            unsigned char* rdst = gring + (unsigned)(it & (RING - 1)) * SLOT + wr0;
            // one instruction sequence for every lane (rows past their length publish zeros and keep their state): ring pieces, slot
            // refill, the dz rows — the top of the next step then waits for everything but the four stores behind the refill
            if (!has_next) dh = (f32x4){0.f, 0.f, 0.f, 0.f};
            dh[0] += bf_lo(g2.x); dh[1] += bf_hi(g2.x); dh[2] += bf_lo(g2.y); dh[3] += bf_hi(g2.y);
            f32x4 di, dj, df, dov, dcn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tc = tanhf_(c[r]);
                float dc = dcs[r] + dh[r] * go[r] * (1.f - tc * tc);
                dov[r] = dh[r] * tc * go[r] * (1.f - go[r]);
                di[r] = dc * gj[r] * gi[r] * (1.f - gi[r]);
                dj[r] = dc * gi[r] * (1.f - gj[r] * gj[r]);
                df[r] = dc * cprev[r] * gf[r] * (1.f - gf[r]);
                dcn[r] = dc * gf[r];
            }
            if (active) dcs = dcn;
            u32x2 p[4];
            p[0].x = pack_bf2(di[0], di[1]); p[0].y = pack_bf2(di[2], di[3]);
            p[1].x = pack_bf2(dj[0], dj[1]); p[1].y = pack_bf2(dj[2], dj[3]);
            p[2].x = pack_bf2(df[0], df[1]); p[2].y = pack_bf2(df[2], df[3]);
            p[3].x = pack_bf2(dov[0], dov[1]); p[3].y = pack_bf2(dov[2], dov[3]);
            if (!active) {
#pragma unroll
                for (int g = 0; g < 4; ++g) p[g] = (u32x2){0u, 0u};
            }
            if (PROTO == 4) {
                // (round 5) as in the forward kernels: a workgroup none of whose rows is inside its sequence yet polls nothing and runs free — here
                // through the FIRST iterations of a tile whose sequences are all shorter than T; it leaves the ring alone, so that a slot never
                // holds the (zero) payload of a free iteration that a faster workgroup, already at its first polling iteration, could take
                // for the gradient of four iterations later.  Rows only wait for pieces of iterations in which they were active (has_next).
                const bool ring_live = RING_LIVE(active);
                if (ring_live) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) *(u32x2*)(rdst + g * 512) = p[g];
                }
                asm volatile("" ::: "memory");
                if (it >= 2 && ring_live) {
                    const u32x2 f = {0xFFFFFFFFu, 0xFFFFFFFFu};
                    unsigned char* old = gring + (unsigned)((it - 2) & (RING - 1)) * SLOT + wr0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *(u32x2*)(old + g * 512) = f;
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int g = 0; g < 4; ++g) *(u32x2*)(zdst + (long)g * U) = p[g];
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) store_wt8(zdst + (long)g * U, p[g]);
            }
        }
        if (PROTO == 0 && s > 0) group_arrive_after(counter, 0);
        DBG_STAMP(3);
    }
}

// ---- four waves per workgroup (round 4): the step's work split FOUR ways instead of waiting four times as long ------------------------------
// The one-wave kernels above spend 2.0 us per step: 0.84 waiting for the other workgroups' rows, 0.44 in 32 dependent-issue MFMAs of ONE
// SIMD and 0.84 in the gate math of 4 units x 5 transcendentals per lane plus eight stores — three quarters of the CU idle throughout.
// Here a workgroup is four waves (one per SIMD) over the same 16 batch rows x 16 units:
//   * each wave holds a QUARTER of the contraction axis (32 / 64 VGPRs of W_h at U = 256 / 512), polls only the ring rows of that quarter
//     (2 instead of 8 operand loads forward, 8 instead of 32 backward) and issues a quarter of the MFMAs;
//   * the partial sums meet in LDS (barrier A), and each wave then OWNS one of the four units every lane held — unit q * 4 + wave of batch
//     row (lane & 15): a quarter of the gate math, its c / dc in one register;
//   * the results go back through LDS (barrier B) into the one-wave kernels' register layout (four consecutive units per lane), so the
//     stores are the same wide ones — but spread over the waves: forward wave 0 the ring piece / refill / output row, waves 1 - 2 two saved
//     gates each, wave 3 the cell; backward wave g the ring piece / refill / dz row of gate g.  (A first version let every owner store
//     its own element: 2-byte and 4-byte stores scattered over 16 rows cost ~0.1 us EACH to issue — the backward pass, with twelve
//     of them, ran 163 us against the one-wave kernel's 133, `profiles/r04o_lstm_bench.jsonl`.)
//   * the operands that do not depend on the recurrence (x-projection; saved gates, cells and the incoming dh) are loaded wide by ONE
//     wave each, a step AHEAD, and reach the owners through LDS under barrier A (the saved gates / cells as they are; the x-projection of
//     gate g and the incoming dh added to the partial sums of the wave that loaded them): vmcnt retires in issue order, so a load issued at
//     the top of a step (as the one-wave kernels do) has to return — from the Infinity Cache or HBM — before the first poll of that step
//     can be looked at; issued right behind the previous step's successful poll it has a whole step.  For the same reason there is no
//     counted wait at the top of a step: the poll's own wait covers every older store of the wave, the refill included, before this
//     step's ring piece is issued;
//   * ring layout, hand-off protocol (4: data-as-flag, the storing wave recycles its piece two steps behind — after barrier A, i.e. after
//     ALL four waves' polls of the step have succeeded and every workgroup of the group is known to have finished the step before),
//     output tensors and saved gates / cell are those of the one-wave kernels: the two families are interchangeable per launch
//     (OCR_LSTM_KSPLIT = 1 / 4, ocr_set_lstm_ksplit).
// LDS buffers are single: what is written before barrier A (B) of a step is read between A and B (behind B), and every wave has passed
// the other barrier before anybody writes the buffer again.
// The sums over the contraction axis are taken in a different order than in the one-wave kernels (four partial sums of K / 4 each): equal up
// to fp32 rounding, not bit-identical.
// The activations with v_rcp_f32 (1 ulp) in place of the correctly rounded reciprocal of sigmoidf_ / tanhf_: the IEEE division is a chain of
// ten dependent instructions, five of them per step sat on the forward recurrence's critical path (one wave per SIMD: nothing hides it).
// the four units of this lane from an element-major LDS exchange array v[value][unit][lane] (four conflict-free 4-byte reads)
#define OUTS4(v_) ((f32x4){outs[v_][0][lane], outs[v_][1][lane], outs[v_][2][lane], outs[v_][3][lane]})
__device__ __forceinline__ float sigmoid_q(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_q(float x) {
    const float e = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}
template <int U, bool SKEW = false /* instance with the test hook (ocr_lstm_seq_test_skew); the product launch is the plain one */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_seq4_kernel(LstmSeqFwdArgs a) {
    constexpr int KS = U / 32 / 4, UB = U / 16;
    constexpr unsigned SLOT = (unsigned)UB * 512u;                       // ring bytes per step and group
    __shared__ f32x4 red[4][4][64];                                      // [source wave][unit r of the lane's four][lane] = partial {i, j, f, o}; wave g's carries x of gate g
    __shared__ float outs[6][4][64];                                     // [h, i, j, f, o, c][unit r of the lane's four: written by wave r][lane] — element-major, so that the
                                                                         // one-element writes of a wave are 64 consecutive words (round 6: as f32x4[lane] they were 8-way bank conflicts)
    const int lane = threadIdx.x & 63;
    const int kh = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ub, d, zb;
    const int nzb = (a.Nb + 15) / 16;
    if (!seq_decode<4, UB>(2 * nzb, ub, d, zb)) return;
    const int T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int ul0 = q * 4;

    bf16x8 w[4][KS];
    {
        const bf16_t* wbase = a.whT + ((long)d * 4 * U + (long)ub * 64 + nl) * U + kh * (KS * 32) + q * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) w[g][kk] = *(const bf16x8*)(wbase + (long)g * 16 * U + kk * 32);
    }
    float c = 0.f;                                                       // cell state of (row nl, unit ul0 + kh)
    int presleep = a.presleep, streak = 0;
    bool dead = false;
    unsigned char* const gring = a.ring + (size_t)(zb * 2 + d) * RING * SLOT;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gring, 0, (int)(RING * SLOT), 0x00020000);
    const unsigned rd0 = (unsigned)((kh * KS * 2 + (q >> 1)) * 512 + nl * 32 + (q & 1) * 16);
    const unsigned wr0 = (unsigned)(ub * 512 + nl * 32 + q * 8);
    // gate kh's x-projection of step s1 for the lane's four units (zero for rows past their length)
    auto xload = [&](int s1) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (nvalid && s1 < len) {
            const int t1 = d == 0 ? s1 : len - 1 - s1;
            x = *(const f32x4*)(a.xproj + ((long)nn * T + t1) * (8L * U) + (long)d * 4 * U + (long)ub * 64 + kh * 16 + ul0);
        }
        return x;
    };
    f32x4 xn = xload(0);
    for (int s = 0; s < T; ++s) {
        const int dbgi = s;
        DBG_STAMP(0);
        if constexpr (SKEW) SKEW_HOOK();
        const bool active = nvalid && s < len;
        const int t = active ? (d == 0 ? s : len - 1 - s) : s;
        const long row = (long)nn * T + t;
        bf16x8 b[KS];
        if (s > 0) {
            const unsigned hoff = (unsigned)((s - 1) & (RING - 1)) * SLOT + rd0;
            unsigned spins = 0;
            for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
            while (true) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = load_pub<4>(rrsrc, hoff + kk * 1024u);
                __builtin_amdgcn_sched_barrier(0);
                unsigned m = 0;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, b[kk]);
                if (dead || !__any(active && holds_fill(m))) break;
                if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); dead = true; break; }
            }
            DBG_STAMP(1);
            POLL_ADAPT();
            if (!active) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        f32x4 xc = xn;
        asm volatile("" : "+v"(xc));                           // taken HERE (behind the poll, whose wait covered it), not where the compiler likes
        xn = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s + 1 < T) xn = xload(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4];                                          // x of gate kh rides on this wave's partial sums (branch-free: the others start at x * 0)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = xc * (kh == g ? 1.f : 0.f);
        if (s > 0) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][kk], b[kk], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[kh][r][lane] = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        __syncthreads();                                        // A
        // gate pre-activations {i, j, f, o} of (row nl, unit ul0 + kh)
        const f32x4 z = (red[0][kh][lane] + red[1][kh][lane]) + (red[2][kh][lane] + red[3][kh][lane]);
        const float gi = sigmoid_q(z[0]), gj = tanh_q(z[1]), gf = sigmoid_q(z[2] + a.forget_bias), go = sigmoid_q(z[3]);
        const float cn = gf * c + gi * gj;
        const float hn = go * tanh_q(cn);
        if (active) c = cn;
        outs[0][kh][lane] = active ? hn : 0.f; outs[1][kh][lane] = gi; outs[2][kh][lane] = gj; outs[3][kh][lane] = gf; outs[4][kh][lane] = go; outs[5][kh][lane] = c;
        __syncthreads();                                        // B
        DBG_STAMP(2);
        if (nvalid) {
            // the one-wave kernel's stores, one part per wave (rows past their length store zeros / throw-away gate values into rows nobody reads)
            if (kh == 0) {
                const f32x4 h = OUTS4(0);
                const u32x2 hp = {pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3])};
                const bool ring_live = RING_LIVE(active);          // a workgroup whose rows are all past their length leaves the ring alone: lstm_fwd_seq_kernel
                if (ring_live) *(u32x2*)(gring + (unsigned)(s & (RING - 1)) * SLOT + wr0) = hp;               // the hand-off payload goes out FIRST
                asm volatile("" ::: "memory");
                if (s >= 2 && ring_live) *(u32x2*)(gring + (unsigned)((s - 2) & (RING - 1)) * SLOT + wr0) = (u32x2){0xFFFFFFFFu, 0xFFFFFFFFu};
                asm volatile("" ::: "memory");
                *(u32x2*)(a.hout + row * (2L * U) + (long)d * U + ub * 16 + ul0) = hp;
            } else if (kh == 3) {
                *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = OUTS4(5);
            } else {
                float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0 + (kh - 1) * 32;
                *(f32x4*)(gdst + 0) = OUTS4(2 * kh - 1);
                *(f32x4*)(gdst + 16) = OUTS4(2 * kh);
            }
        }
        DBG_STAMP(3);
    }
}

// ---- the forward recurrence WITH the input projection (round 5) -----------------------------------------------------------------------------------
// x_t W_x does not depend on the recurrence: the four-wave kernel above takes it from a tensor that a GEMM launch wrote (33 MB of fp32 at
// N = 64, T = 63, U = 256: written once, read once, 17 - 21 us of igemm in front of the kernel).  Here each wave also holds ITS quarter of the
// input features' W_x slice in registers (the same 64 packed gate rows as its W_h slice: D / 4 columns = 64 VGPRs at D = 512), loads the bf16
// x_t fragments of its quarter one step ahead, and issues the D / 32 / 4 x 4 MFMAs of the input part BETWEEN issuing a step's first poll and
// looking at it — the poll's L2 round trip (~0.5 us) is longer than those 16 MFMAs (256 clocks), so the projection costs no time on the
// recurrence's critical path, and its launch, its tensor and their traffic are gone.  The partial sums of input and recurrent part share the
// accumulators (the input part is their start value) and meet in the same LDS exchange; the bias is added by the owning lane.
struct LstmSeqFwdXArgs {
    const bf16_t* x; const bf16_t* wxT; const float* bias; const bf16_t* whT; const int* seq_len; bf16_t* hout; float* gates; float* cell;
    unsigned char* ring; int* err;
    int Nb, T; float forget_bias; long long* dbg; int presleep; int skew; int skew_at; int ring_always;
};
template <int U, int D, bool SKEW = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_seq4x_kernel(LstmSeqFwdXArgs a) {
    constexpr int KS = U / 32 / 4, KX = D / 32 / 4, UB = U / 16;
    constexpr unsigned SLOT = (unsigned)UB * 512u;                       // ring bytes per step and group
    __shared__ f32x4 red[4][4][64];                                      // [source wave][unit r of the lane's four][lane] = partial {i, j, f, o}
    __shared__ float outs[6][4][64];                                     // [h, i, j, f, o, c][unit r of the lane's four: written by wave r][lane] — element-major, so that the
                                                                         // one-element writes of a wave are 64 consecutive words (round 6: as f32x4[lane] they were 8-way bank conflicts)
    const int lane = threadIdx.x & 63;
    const int kh = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ub, d, zb;
    const int nzb = (a.Nb + 15) / 16;
    if (!seq_decode<4, UB>(2 * nzb, ub, d, zb)) return;
    const int T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int ul0 = q * 4;

    bf16x8 w[4][KS], wx[4][KX];
    {
        const bf16_t* wbase = a.whT + ((long)d * 4 * U + (long)ub * 64 + nl) * U + kh * (KS * 32) + q * 8;
        const bf16_t* xbase = a.wxT + ((long)d * 4 * U + (long)ub * 64 + nl) * D + kh * (KX * 32) + q * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) w[g][kk] = *(const bf16x8*)(wbase + (long)g * 16 * U + kk * 32);
#pragma unroll
            for (int kk = 0; kk < KX; ++kk) wx[g][kk] = *(const bf16x8*)(xbase + (long)g * 16 * D + kk * 32);
        }
    }
    // bias of the four gates of the unit this lane finishes (packed gate-column order, as the projection GEMM's bias vector)
    f32x4 bo;
    {
        const float* bp = a.bias + (long)d * 4 * U + ub * 64 + ul0 + kh;
        bo = (f32x4){bp[0], bp[16], bp[32], bp[48]};
    }
    float c = 0.f;                                                       // cell state of (row nl, unit ul0 + kh)
    int presleep = a.presleep, streak = 0;
    bool dead = false;
    unsigned char* const gring = a.ring + (size_t)(zb * 2 + d) * RING * SLOT;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gring, 0, (int)(RING * SLOT), 0x00020000);
    const unsigned rd0 = (unsigned)((kh * KS * 2 + (q >> 1)) * 512 + nl * 32 + (q & 1) * 16);
    const unsigned wr0 = (unsigned)(ub * 512 + nl * 32 + q * 8);
    // this wave's quarter of the input row of step s1 as MFMA B fragments (zeros for rows past their length)
    auto xload = [&](int s1, bf16x8 (&xb)[KX]) {
#pragma unroll
        for (int kk = 0; kk < KX; ++kk) xb[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (nvalid && s1 < len) {
            const int t1 = d == 0 ? s1 : len - 1 - s1;
            const bf16_t* xp = a.x + ((long)nn * T + t1) * D + kh * (KX * 32) + q * 8;
#pragma unroll
            for (int kk = 0; kk < KX; ++kk) xb[kk] = *(const bf16x8*)(xp + kk * 32);
        }
    };
    bf16x8 xn[KX];
    xload(0, xn);
    // One step.  The first one (no recurrent part, no poll) is a copy of its own: with `if (s > 0)` around the poll the compiler merges a path
    // with and a path without poll loads in flight and then waits for ALL vector-memory operations before the input fragments are used —
    // i.e. for the poll, which the input part is meant to overlap.
    auto step = [&](const int s, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int dbgi = s;
        DBG_STAMP(0);
        if constexpr (SKEW) SKEW_HOOK();
        const bool active = nvalid && s < len;
        const int t = active ? (d == 0 ? s : len - 1 - s) : s;
        const long row = (long)nn * T + t;
        bf16x8 b[KS];
        const unsigned hoff = (unsigned)((s - 1) & (RING - 1)) * SLOT + rd0;
        if (!FIRST) {
            for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) b[kk] = load_pub<4>(rrsrc, hoff + kk * 1024u);          // the first poll goes out ...
        }
        __builtin_amdgcn_sched_barrier(0);
        // ... and the input part is multiplied while it is under way: its fragments were requested a step ago (vmcnt retires in order, so
        // they — and the last step's stores — land before the poll's data in any case)
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KX; ++kk)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wx[g][kk], xn[kk], acc[g], 0, 0, 0);
        // pinned HERE: the products are pure register arithmetic whose results are needed only behind the poll, and the compiler otherwise
        // sinks all sixteen MFMAs below the polling loop (seen in the ISA) — behind the wait they are meant to fill
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) :: "memory");
        if (!FIRST) {
            unsigned spins = 0;
            while (true) {
                unsigned m = 0;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, b[kk]);
                if (dead || !__any(active && holds_fill(m))) break;
                if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); dead = true; break; }
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = load_pub<4>(rrsrc, hoff + kk * 1024u);
                __builtin_amdgcn_sched_barrier(0);
            }
            DBG_STAMP(1);
            POLL_ADAPT();
            if (!active) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) b[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
            // the operands are final HERE, in front of the next step's loads: sunk below them (as the compiler did) the selects above waited
            // for those loads — on the critical path
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(b[kk]) :: "memory");
        }
        if (s + 1 < T) xload(s + 1, xn);                       // next step's fragments: behind the poll, in front of this step's stores
        __builtin_amdgcn_sched_barrier(0);
        if (!FIRST) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g][kk], b[kk], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[kh][r][lane] = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        __syncthreads();                                        // A
        // gate pre-activations {i, j, f, o} of (row nl, unit ul0 + kh)
        f32x4 z = (red[0][kh][lane] + red[1][kh][lane]) + (red[2][kh][lane] + red[3][kh][lane]);
        if (active) z += bo;                                   // (rows past their length: x = 0 and no bias — throw-away values in rows nobody reads)
        const float gi = sigmoid_q(z[0]), gj = tanh_q(z[1]), gf = sigmoid_q(z[2] + a.forget_bias), go = sigmoid_q(z[3]);
        const float cn = gf * c + gi * gj;
        const float hn = go * tanh_q(cn);
        if (active) c = cn;
        outs[0][kh][lane] = active ? hn : 0.f; outs[1][kh][lane] = gi; outs[2][kh][lane] = gj; outs[3][kh][lane] = gf; outs[4][kh][lane] = go; outs[5][kh][lane] = c;
        __syncthreads();                                        // B
        DBG_STAMP(2);
        if (nvalid) {
            if (kh == 0) {
                const f32x4 h = OUTS4(0);
                const u32x2 hp = {pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3])};
                const bool ring_live = RING_LIVE(active);          // a workgroup whose rows are all past their length leaves the ring alone: lstm_fwd_seq_kernel
                if (ring_live) *(u32x2*)(gring + (unsigned)(s & (RING - 1)) * SLOT + wr0) = hp;               // the hand-off payload goes out FIRST
                asm volatile("" ::: "memory");
                if (s >= 2 && ring_live) *(u32x2*)(gring + (unsigned)((s - 2) & (RING - 1)) * SLOT + wr0) = (u32x2){0xFFFFFFFFu, 0xFFFFFFFFu};
                asm volatile("" ::: "memory");
                *(u32x2*)(a.hout + row * (2L * U) + (long)d * U + ub * 16 + ul0) = hp;
            } else if (kh == 3) {
                *(f32x4*)(a.cell + ((long)d * R + row) * U + ub * 16 + ul0) = OUTS4(5);
            } else {
                float* gdst = a.gates + ((long)d * R + row) * (4L * U) + (long)ub * 64 + ul0 + (kh - 1) * 32;
                *(f32x4*)(gdst + 0) = OUTS4(2 * kh - 1);
                *(f32x4*)(gdst + 16) = OUTS4(2 * kh);
            }
        }
        DBG_STAMP(3);
    };
    step(0, std::true_type{});
    for (int s = 1; s < T; ++s) step(s, std::false_type{});
}

template <int U, bool SKEW = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_seq4_kernel(LstmSeqBwdArgs a) {
    constexpr int KS = U / 32, UB = U / 16;                              // wave kh multiplies gate kh's columns of W_h: K = kh U .. (kh + 1) U of 4U
    constexpr unsigned SLOT = (unsigned)UB * 2048u;                      // ring bytes per step and group: [ub][gate][16 rows][16 units]
    // element-major exchange arrays (round 6): every wave reads / writes ONE element of a lane's four per access — as f32x4[lane] those were 4-byte
    // accesses at a 16-byte stride, 8-way bank conflicts on 14 LDS instructions per step (PMC: lds_conflict_frac 0.62); as [element][lane] they are
    // 64 consecutive words, and the wide side pays four 4-byte accesses instead of one 16-byte one (the same LDS cycles)
    __shared__ float red[4][4][64];                                      // [source wave][unit e of the lane's four][lane] = partial dh
    __shared__ float sv[6][4][64];                                       // [i, j, f, o, c, c_prev][unit e][lane] = saved operands
    __shared__ float outs[4][4][64];                                     // [gate][unit e: written by wave e][lane] = gate gradients
    const int lane = threadIdx.x & 63;
    const int kh = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ub, d, zb;
    const int nzb = (a.Nb + 15) / 16;
    if (!seq_decode<4, UB>(2 * nzb, ub, d, zb)) return;
    const int T = a.T;
    const int nl = lane & 15, q = lane >> 4;
    const int n = zb * 16 + nl;
    const bool nvalid = n < a.Nb;
    const int nn = nvalid ? n : 0;
    const int len = min(a.seq_len[nn], T);
    const long R = (long)a.Nb * T;
    const int u0 = ub * 16 + q * 4;

    bf16x8 w[KS];
    {
        const bf16_t* wbase = a.wh + (long)d * a.w_dir_stride + ((long)ub * 16 + nl) * a.ldw + kh * (KS * 32) + q * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) w[kk] = *(const bf16x8*)(wbase + kk * 32);
    }
    float dcs = 0.f;                                                     // cell gradient of (row nl, unit q * 4 + kh)
    int presleep = a.presleep, streak = 0;
    bool dead = false;
    unsigned char* const gring = a.ring + (size_t)(zb * 2 + d) * RING * SLOT;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gring, 0, (int)(RING * SLOT), 0x00020000);
    // k-step kk of this wave: gate kh, units kk 32 + q 8 .. + 8 = unit block 2 kk + (q >> 1), 16-byte half (q & 1)
    const unsigned rdl = (unsigned)((q >> 1) * 2048 + kh * 512 + nl * 32 + (q & 1) * 16);
    const unsigned wr0 = (unsigned)(ub * 2048 + kh * 512 + nl * 32 + q * 8);       // this wave stores gate kh
    // the saved operands of step s1, two per wave (wave 0: i, j; 1: f, o; 2: c, c_prev; 3: incoming dh as bf16 bits in A[0], A[1])
    auto sload = [&](int s1, f32x4& A, f32x4& B) {
        A = (f32x4){0.f, 0.f, 0.f, 0.f}; B = A;
        if (nvalid && s1 < len) {
            const int t1 = d == 0 ? s1 : len - 1 - s1;
            const int tp = d == 0 ? t1 - 1 : t1 + 1;
            const long row1 = (long)nn * T + t1;
            if (kh < 2) {
                const float* gsrc = a.gates + ((long)d * R + row1) * (4L * U) + (long)ub * 64 + q * 4 + kh * 32;
                A = *(const f32x4*)gsrc; B = *(const f32x4*)(gsrc + 16);
            } else if (kh == 2) {
                A = *(const f32x4*)(a.cell + ((long)d * R + row1) * U + u0);
                if (s1 > 0) B = *(const f32x4*)(a.cell + ((long)d * R + (long)nn * T + tp) * U + u0);
            } else {
                const u32x2 g2 = *(const u32x2*)(a.dhout + row1 * (2L * U) + (long)d * U + u0);
                A[0] = __uint_as_float(g2.x); A[1] = __uint_as_float(g2.y);
            }
        }
    };
    f32x4 nA, nB;
    sload(T - 1, nA, nB);
    for (int s = T - 1, it = 0; s >= 0; --s, ++it) {
        const int dbgi = it;
        DBG_STAMP(0);
        if constexpr (SKEW) SKEW_HOOK();
        const bool active = nvalid && s < len;
        const bool has_next = nvalid && (s + 1 < len);
        const int t = active ? (d == 0 ? s : len - 1 - s) : s;
        const long row = (long)nn * T + t;
        bf16x8 z[KS];
        if (it > 0) {
            const unsigned roff = (unsigned)((it - 1) & (RING - 1)) * SLOT + rdl;
            unsigned spins = 0;
            for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(1);
            while (true) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) z[kk] = load_pub<4>(rrsrc, roff, (unsigned)(kk * 2 * 2048));
                __builtin_amdgcn_sched_barrier(0);
                unsigned m = 0;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) m = fold_fill(m, z[kk]);
                if (dead || !__any(has_next && holds_fill(m))) break;
                if (++spins > (SPIN_LIMIT >> 4)) { if (lane == 0) atomicExch(a.err, 1); dead = true; break; }
            }
            DBG_STAMP(1);
            POLL_ADAPT();
            if (!has_next) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) z[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        f32x4 A = nA, B = nB;
        asm volatile("" : "+v"(A), "+v"(B));                   // taken behind the poll
        nA = (f32x4){0.f, 0.f, 0.f, 0.f}; nB = nA;
        if (s > 0) sload(s - 1, nA, nB);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;        // (rows without a successor step multiplied zeros)
        if (it > 0) {
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 0], z[k0 + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k0 + 1], z[k0 + 1], acc1, 0, 0, 0);
            }
        }
        if (kh < 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sv[2 * kh][e][lane] = A[e]; sv[2 * kh + 1][e][lane] = B[e]; }
        }
        else {                                                  // the incoming dh rides on this wave's partial sums
            const unsigned gx = __float_as_uint(A[0]), gy = __float_as_uint(A[1]);
            acc1 += (f32x4){bf_lo(gx), bf_hi(gx), bf_lo(gy), bf_hi(gy)};
        }
        {
            const f32x4 part = acc0 + acc1;
#pragma unroll
            for (int e = 0; e < 4; ++e) red[kh][e][lane] = part[e];
        }
        __syncthreads();                                        // A
        const float gi = sv[0][kh][lane], gj = sv[1][kh][lane], gf = sv[2][kh][lane], go = sv[3][kh][lane], c = sv[4][kh][lane], cprev = sv[5][kh][lane];
        const float dh = (red[0][kh][lane] + red[1][kh][lane]) + (red[2][kh][lane] + red[3][kh][lane]);
        const float tc = tanh_q(c);
        const float dc = dcs + dh * go * (1.f - tc * tc);
        float dg[4];
        dg[3] = dh * tc * go * (1.f - go);
        dg[0] = dc * gj * gi * (1.f - gi);
        dg[1] = dc * gi * (1.f - gj * gj);
        dg[2] = dc * cprev * gf * (1.f - gf);
        if (active) dcs = dc * gf;
#pragma unroll
        for (int g = 0; g < 4; ++g) outs[g][kh][lane] = active ? dg[g] : 0.f;
        __syncthreads();                                        // B
        DBG_STAMP(2);
        if (nvalid) {
            const f32x4 v = OUTS4(kh);
            const u32x2 p = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            const bool ring_live = RING_LIVE(active);              // a workgroup none of whose rows is inside its sequence yet leaves the ring alone: lstm_bwd_seq_kernel
            if (ring_live) *(u32x2*)(gring + (unsigned)(it & (RING - 1)) * SLOT + wr0) = p;
            asm volatile("" ::: "memory");
            if (it >= 2 && ring_live) *(u32x2*)(gring + (unsigned)((it - 2) & (RING - 1)) * SLOT + wr0) = (u32x2){0xFFFFFFFFu, 0xFFFFFFFFu};
            asm volatile("" ::: "memory");
            *(u32x2*)(a.dz + row * (8L * U) + (long)d * 4 * U + (long)kh * U + u0) = p;
        }
        DBG_STAMP(3);
    }
}

// ------------------------------------------------------------------------------------------
// C ABI.  `sync` is a caller-owned device block of ocr_lstm_seq_sync_words(Nb, U) 32-bit words: the group counters, the
// protocol-4 ring and, as its LAST word, an error word; the call's own fill launch initialises what its protocol needs
// (hipGraph-replayable).  Returns OCR_ERR_INVALID for shapes the persistent kernels do not cover (caller falls back to the
// step kernels).
// ------------------------------------------------------------------------------------------
// counters are cleared by a kernel, not hipMemsetAsync: under hipGraph replay a memset NODE of these sizes was observed to
// fill the block with a stale non-zero pattern (ROCm 7.2), which makes every wait time out
__global__ void zero_words_kernel(unsigned* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

static long long* g_lstm_dbg = nullptr;
// test/diagnostic hook: device buffer of 4*T int64 receiving wall-clock stamps of workgroup 0 (NULL = off)
extern "C" int ocr_lstm_seq_debug(void* dbg) { g_lstm_dbg = (long long*)dbg; return OCR_OK; }

// test hook (SKEW_HOOK above): unit block 0 of every group sleeps `units` x 64 clocks at iteration `at` (every iteration if at < 0) in the launches that
// follow (units 0 = off, the default)
static int g_lstm_skew = 0, g_lstm_skew_at = -1;
extern "C" int ocr_lstm_seq_test_skew(int units, int at) {
    if (units < 0 || units > 4096) return OCR_ERR_INVALID;
    g_lstm_skew = units; g_lstm_skew_at = at;
    return OCR_OK;
}
// experiments build only: OCR_LSTM_RING_RULE=always restores the ring rule of rounds 3-4 (RING_LIVE above); the product library always returns 0
static int seq_ring_always() {
    static int v = -1;
    if (v < 0) { const char* e = ocr_tune_env("OCR_LSTM_RING_RULE"); v = (e && e[0] == 'a') ? 1 : 0; }
    return v;
}
// environment knobs are looked up ONCE per process (the launch path runs every step when graphs are off)
static int seq_env_int(const char* name, int slot, int dflt, bool tuning = false /* read by experiments builds only */) {
    static int val[5], seen[5];
    if (!seen[slot]) { const char* e = tuning ? ocr_tune_env(name) : getenv(name); val[slot] = e ? atoi(e) : dflt; seen[slot] = 1; }
    return val[slot];
}
// batch rows per workgroup: the smallest tile whose grid still fits one workgroup per CU (measured at N = 64, U = 256, us per
// step forward / backward: 16 rows 2.6 / 3.2, 32 rows 3.0 / 4.1, 64 rows 3.6 / 6.2 — the per-step cost is the hand-off payload a
// CU has to pull plus a fixed part).  OCR_LSTM_ROWS forces 16 / 32 / 64 where that fits.
static int seq_rows_per_wg(int Nb, int U) {
    if (U != 256 && U != 512) return 0;
    const int want = seq_env_int("OCR_LSTM_ROWS", 2, 0);
    for (int pass = 0; pass < 2; ++pass)
        for (int rows = 16; rows <= 64; rows *= 2) {
            if (pass == 0 && rows != want) continue;
            if (U == 512 && rows == 64) continue;                        // 8 waves of 256 VGPRs: the backward kernel would spill
            if ((U / 16) * 2 * ceil_div(Nb, rows) <= 256) return rows;
        }
    return 0;
}
static int g_seq_proto = -1;
// hand-off protocol: 0 counters (sc1), 4 data-as-flag through a ring inside one XCD (default).  Environment OCR_LSTM_PROTO (A/B) wins over the setter, which the host uses to fall back to 0 when the device
// does not co-locate workgroups with equal (id & 7) on one XCD (checked once with ocr_probe_xcc).
extern "C" int ocr_set_lstm_proto(int proto) {
    if (proto != 0 && proto != 4) return OCR_ERR_INVALID;
    g_seq_proto = proto;
    return OCR_OK;
}
static int seq_proto() {
    const int env = seq_env_int("OCR_LSTM_PROTO", 3, -1);
    if (env == 0 || env == 4) return env;
    return g_seq_proto >= 0 ? g_seq_proto : 4;
}
// waves per workgroup of the 16-row, protocol-4 kernels: 4 (default, round 4: lstm_*_seq4_kernel) or 1 (the one-wave kernels).  Environment
// OCR_LSTM_KSPLIT wins over the setter (tests run both families in one process through the setter).
static int g_seq_ksplit = -1;
extern "C" int ocr_set_lstm_ksplit(int k) {
    if (k != 1 && k != 4) return OCR_ERR_INVALID;
    g_seq_ksplit = k;
    return OCR_OK;
}
static int seq_ksplit() {
    const int env = seq_env_int("OCR_LSTM_KSPLIT", 4, -1);
    if (env == 1 || env == 4) return env;
    return g_seq_ksplit >= 0 ? g_seq_ksplit : 4;
}
extern "C" int ocr_lstm_seq_supported(int Nb, int U) { return Nb > 0 && seq_rows_per_wg(Nb, U) != 0; }
// int32 words the caller must provide in `sync`: group counters | ring (sized for the backward pass and the smallest tile) | tail
static long seq_counter_words(int Nb) { return 2L * ceil_div(Nb, 16) * CNT_STRIDE; }
static long seq_ring_bytes_max(int Nb, int U) { return (long)RING * (2L * ceil_div(Nb, 16) * 16 + 96) * 4 * U * 2; }
extern "C" long ocr_lstm_seq_sync_words(int Nb, int U) {
    if (Nb <= 0 || U <= 0) return 0;
    return seq_counter_words(Nb) + seq_ring_bytes_max(Nb, U) / 4 + CNT_STRIDE;
}

template <int U, int PROTO>
static void launch_fwd(const LstmSeqFwdArgs& a, int rows, dim3 grid3, dim3 grid1, hipStream_t stream) {
    constexpr int KSP = U / 256;
    const dim3 G = PROTO < 2 ? grid3 : grid1;
    if (rows == 16) lstm_fwd_seq_kernel<U, 1, KSP, PROTO><<<G, 64 * KSP, 0, stream>>>(a);
    else if (rows == 32) lstm_fwd_seq_kernel<U, 2, KSP, PROTO><<<G, 128 * KSP, 0, stream>>>(a);
    else if constexpr (KSP == 1) lstm_fwd_seq_kernel<U, 4, KSP, PROTO><<<G, 256 * KSP, 0, stream>>>(a);
}
template <int U, int PROTO>
static void launch_bwd(const LstmSeqBwdArgs& a, int rows, dim3 grid3, dim3 grid1, hipStream_t stream) {
    constexpr int KSP = U / 256;
    const dim3 G = PROTO < 2 ? grid3 : grid1;
    if (rows == 16) lstm_bwd_seq_kernel<U, 1, KSP, PROTO><<<G, 64 * KSP, 0, stream>>>(a);
    else if (rows == 32) lstm_bwd_seq_kernel<U, 2, KSP, PROTO><<<G, 128 * KSP, 0, stream>>>(a);
    else if constexpr (KSP == 1) lstm_bwd_seq_kernel<U, 4, KSP, PROTO><<<G, 256 * KSP, 0, stream>>>(a);
}

// the call's own fill launch: counters + error word := 0, the ring := 0xFFFF
static void seq_prepare(int proto, void* sync, long words, long cwords, long ring_words, hipStream_t stream) {
    unsigned* s = (unsigned*)sync;
    if (proto == 0) {
        zero_words_kernel<<<1, 256, 0, stream>>>(s, (int)cwords);
        zero_words_kernel<<<1, 64, 0, stream>>>(s + words - CNT_STRIDE, CNT_STRIDE);
    } else {
        fill_words_kernel<<<64, 256, 0, stream>>>(s + cwords, ring_words, 0xFFFFFFFFu, s + words - CNT_STRIDE, CNT_STRIDE);
    }
}

// flags & OCR_LSTM_PREPARED (1): the caller has set EVERY word of `sync` to 0xFFFFFFFF earlier on this stream (the training engine does it for
// all of a step's launches inside a kernel it runs anyway: ocr_conv1_pool_fwd_train) — the call's own fill launch is skipped.  Honoured under
// protocol 4 only (the counter protocol needs zeros: the call then prepares the block itself).  The error word of such a block reads
// 0xFFFFFFFF when nothing happened, 1 after a time-out.
extern "C" int ocr_lstm_fwd_seq2(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                 float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                                 int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!xproj || !whT_packed || !seq_len || !hout || !gates || !cell || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    const int rows = seq_rows_per_wg(Nb, U);
    const int nz = ceil_div(Nb, rows);
    const long words = ocr_lstm_seq_sync_words(Nb, U), cwords = seq_counter_words(Nb);
    const int proto = seq_proto();
    const int wpb = rows / 16;
    const long ring_words = (long)(2 * nz) * RING * ((long)(U / 16) * wpb * 512) / 4;
    if (!((flags & 1) && proto == 4)) {
        seq_prepare(proto, sync, words, cwords, ring_words, stream);
        OCR_CHECK_LAUNCH();
    }
    LstmSeqFwdArgs a = {xproj, (const bf16_t*)whT_packed, seq_len, (bf16_t*)hout, gates, cell, (unsigned*)sync,
                        (unsigned char*)((unsigned*)sync + cwords), (int*)sync + words - 1, Nb, T, forget_bias, g_lstm_dbg,
                        seq_env_int("OCR_LSTM_PRESLEEP_F", 0, 0, true), g_lstm_skew, g_lstm_skew_at, seq_ring_always()};
    const dim3 grid3(U / 16, 2, nz), grid1((U / 16) * 8 * ceil_div(2 * nz, 8));
#define FWD(UU) do { if (proto == 0) launch_fwd<UU, 0>(a, rows, grid3, grid1, stream); \
                     else if (rows == 16 && seq_ksplit() == 4) { if (a.skew) lstm_fwd_seq4_kernel<UU, true><<<grid1, 256, 0, stream>>>(a); else lstm_fwd_seq4_kernel<UU><<<grid1, 256, 0, stream>>>(a); } \
                     else launch_fwd<UU, 4>(a, rows, grid3, grid1, stream); } while (0)
    if (U == 256) FWD(256); else FWD(512);
#undef FWD
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" int ocr_lstm_fwd_seq(const float* xproj, const void* whT_packed, const int* seq_len, void* hout,
                                float* gates, float* cell, int Nb, int T, int U, float forget_bias, void* sync,
                                void* stream_) {
    return ocr_lstm_fwd_seq2(xproj, whT_packed, seq_len, hout, gates, cell, Nb, T, U, forget_bias, sync, 0, stream_);
}

extern "C" int ocr_lstm_bwd_seq2(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                 const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                                 int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!wh || !seq_len || !dhout || !gates || !cell || !dz || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_seq_supported(Nb, U)) return OCR_ERR_INVALID;
    const int rows = seq_rows_per_wg(Nb, U);
    const int nz = ceil_div(Nb, rows);
    const long words = ocr_lstm_seq_sync_words(Nb, U), cwords = seq_counter_words(Nb);
    const int proto = seq_proto();
    const int wpb = rows / 16;
    const long ring_words = (long)(2 * nz) * RING * ((long)(U / 16) * wpb * 2048) / 4;
    if (!((flags & 1) && proto == 4)) {
        seq_prepare(proto, sync, words, cwords, ring_words, stream);
        OCR_CHECK_LAUNCH();
    }
    LstmSeqBwdArgs a = {(const bf16_t*)wh, ldw, w_dir_stride, seq_len, (const bf16_t*)dhout, gates, cell, (bf16_t*)dz,
                        (unsigned*)sync, (unsigned char*)((unsigned*)sync + cwords), (int*)sync + words - 1, Nb, T, g_lstm_dbg,
                        seq_env_int("OCR_LSTM_PRESLEEP", 1, 4, true), g_lstm_skew, g_lstm_skew_at, seq_ring_always()};
    const dim3 grid3(U / 16, 2, nz), grid1((U / 16) * 8 * ceil_div(2 * nz, 8));
#define BWD(UU) do { if (proto == 0) launch_bwd<UU, 0>(a, rows, grid3, grid1, stream); \
                     else if (rows == 16 && seq_ksplit() == 4) { if (a.skew) lstm_bwd_seq4_kernel<UU, true><<<grid1, 256, 0, stream>>>(a); else lstm_bwd_seq4_kernel<UU><<<grid1, 256, 0, stream>>>(a); } \
                     else launch_bwd<UU, 4>(a, rows, grid3, grid1, stream); } while (0)
    if (U == 256) BWD(256); else BWD(512);
#undef BWD
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" int ocr_lstm_bwd_seq(const void* wh, long ldw, long w_dir_stride, const int* seq_len, const void* dhout,
                                const float* gates, const float* cell, void* dz, int Nb, int T, int U, void* sync,
                                void* stream_) {
    return ocr_lstm_bwd_seq2(wh, ldw, w_dir_stride, seq_len, dhout, gates, cell, dz, Nb, T, U, sync, 0, stream_);
}

// The forward recurrence with the input projection inside (lstm_fwd_seq4x_kernel): x bf16 [Nb * T][D], wxT_packed bf16 [2][4U][D] and bias
// fp32 [2][4U] in the packed gate order of the projection GEMM's operands (ocr_pack_transpose with lstm_units / ocr_lstm_pack_bias).
// Covered: the ring protocol, 16-row tiles, four-wave workgroups, D = 512 or 1024, U = 256 or 512 (ocr_lstm_fwd_seq_x_supported); otherwise
// OCR_ERR_INVALID and the caller runs the projection GEMM + ocr_lstm_fwd_seq2.
extern "C" int ocr_lstm_fwd_seq_x_supported(int Nb, int U, int D) {
    return Nb > 0 && (D == 512 || D == 1024) && (U == 256 || U == 512) && seq_rows_per_wg(Nb, U) == 16 && seq_proto() == 4 && seq_ksplit() == 4;
}
extern "C" int ocr_lstm_fwd_seq_x(const void* x, const void* wxT_packed, const float* bias_packed, int D, const void* whT_packed,
                                  const int* seq_len, void* hout, float* gates, float* cell, int Nb, int T, int U, float forget_bias,
                                  void* sync, int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !wxT_packed || !bias_packed || !whT_packed || !seq_len || !hout || !gates || !cell || !sync) return OCR_ERR_INVALID;
    if (Nb <= 0 || T <= 0 || !ocr_lstm_fwd_seq_x_supported(Nb, U, D)) return OCR_ERR_INVALID;
    const int nz = ceil_div(Nb, 16);
    const long words = ocr_lstm_seq_sync_words(Nb, U), cwords = seq_counter_words(Nb);
    const long ring_words = (long)(2 * nz) * RING * ((long)(U / 16) * 512) / 4;
    if (!(flags & 1)) {
        seq_prepare(4, sync, words, cwords, ring_words, stream);
        OCR_CHECK_LAUNCH();
    }
    LstmSeqFwdXArgs a = {(const bf16_t*)x, (const bf16_t*)wxT_packed, bias_packed, (const bf16_t*)whT_packed, seq_len, (bf16_t*)hout, gates, cell,
                         (unsigned char*)((unsigned*)sync + cwords), (int*)sync + words - 1, Nb, T, forget_bias, g_lstm_dbg,
                         seq_env_int("OCR_LSTM_PRESLEEP_F", 0, 0, true), g_lstm_skew, g_lstm_skew_at, seq_ring_always()};
    const dim3 grid1((U / 16) * 8 * ceil_div(2 * nz, 8));
    if (a.skew && U == 256 && D == 512) lstm_fwd_seq4x_kernel<256, 512, true><<<grid1, 256, 0, stream>>>(a);      // the test hook's instance (the stress tests' shape)
    else if (U == 256 && D == 512) lstm_fwd_seq4x_kernel<256, 512><<<grid1, 256, 0, stream>>>(a);
    else if (U == 512 && D == 512) lstm_fwd_seq4x_kernel<512, 512><<<grid1, 256, 0, stream>>>(a);
    else if (U == 256) lstm_fwd_seq4x_kernel<256, 1024><<<grid1, 256, 0, stream>>>(a);
    else lstm_fwd_seq4x_kernel<512, 1024><<<grid1, 256, 0, stream>>>(a);       // configs[4]'s second BiLSTM layer: 255 VGPRs + 54 AGPRs, no scratch
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
