// CTC loss + gradient and greedy decode for gfx950: one 64-lane wavefront per sample.
//
// Replaces the un-vendored baidu warp-ctc kernels the reference reaches through
// `warpctc_tensorflow.ctc(...)` (reference lib/networks/network.py:653-654) and the best-path
// part of the decode at network.py:656-657 / lib/lstm/test.py:30-31.
//
// Semantics (SURVEY.md Appendix A "CTC (warp-ctc)"):
//   y = softmax_C(act[t,n,:]);  l' = [b,l1,b,...,lL,b], S = 2L+1, b = blank_label
//   cost_n = -log p(l|x);  d cost/d act[t,n,k] = y[t,n,k] - sum_{s: l'_s = k} gamma_t(s)
//   for t < input_length, 0 beyond;  infeasible (L + repeats > T) => cost 0, grad 0.
//
// Mapping: lane s (+64*v) owns extended-label slot s; the alpha row lives in LDS so the
// s-1 / s-2 neighbours are one ds_read away; a single wave per workgroup means the
// step-to-step barrier is free. alpha[T][S] is parked in a caller-owned workspace in HBM
// (L2-resident at these sizes); beta is consumed on the fly while the gradient row is
// produced with all 64 lanes striding the class axis (one coalesced 256-B row at C = 64).
#include "common.h"
#include <math.h>

#define NEG_INF (-INFINITY)

__device__ __forceinline__ float lse2(float a, float b) {
    float m = fmaxf(a, b);
    if (m == NEG_INF) return NEG_INF;
    return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    float m = fmaxf(fmaxf(a, b), c);
    if (m == NEG_INF) return NEG_INF;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// hardware exp2 / log2 forms (v_exp_f32, v_log_f32: ~1e-6 relative) for the T-sequential recursions of the fast kernel,
// where the libm expf / logf sequences (about 40 instructions each, three per step) ARE the critical path
__device__ __forceinline__ float lse2_fast(float a, float b) {
    float m = fmaxf(a, b);
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
    float m = fmaxf(fmaxf(a, b), c);
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// whole-wave shifts by one lane as DPP moves (wave_shr:1 / wave_shl:1, one VALU instruction; the lane shifted in takes `fill`)
// instead of ds_bpermute round trips through the LDS crossbar: they sit on the T-sequential critical path of the recursions
__device__ __forceinline__ float wave_shr1(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// VT = extended-label slots per lane (S <= 64*VT).
template <int VT>
__global__ __launch_bounds__(64) void ctc_loss_grad_kernel(
    const float* __restrict__ act,      // [T, N, C] unnormalised
    float* __restrict__ grad,           // [T, N, C] or nullptr
    const int* __restrict__ flat_labels,
    const int* __restrict__ label_off,  // [N] exclusive prefix sum of label_lengths
    const int* __restrict__ label_len,  // [N]
    const int* __restrict__ input_len,  // [N]
    int T, int N, int C, int blank,
    float* __restrict__ costs,          // [N]
    float* __restrict__ ws,             // workspace: per sample T*(SMAX+1) floats
    int SMAX) {
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS carve-up: row[2][SMAX+2] (ping-pong alpha/beta rows, 2 guard slots in front),
    //               acc[C] (posterior per class), lab[SMAX] (extended labels as int)
    float* rowbuf = smem;                              // 2*(SMAX+2)
    float* acc = rowbuf + 2 * (SMAX + 2);              // C
    int* lab = (int*)(acc + C);                        // SMAX

    const int L = label_len[n];
    const int Tn = min(input_len[n], T);
    const int S = 2 * L + 1;
    const int* labels = flat_labels + label_off[n];
    float* alpha_ws = ws + (size_t)n * T * (SMAX + 1);   // alpha[t][s]; slot SMAX of row t = lse[t]
    const size_t tstride = (size_t)N * C;
    const float* a_n = act + (size_t)n * C;
    float* g_n = grad ? grad + (size_t)n * C : nullptr;

    // zero-fill the gradient of frames this sample does not own (t >= Tn)
    if (g_n) {
        for (int t = Tn; t < T; ++t)
            for (int k = lane; k < C; k += 64) g_n[t * tstride + k] = 0.f;
    }

    // extended labels + repeat count
    int repeats = 0;
    for (int s = lane; s < S; s += 64) {
        int v = (s & 1) ? labels[s >> 1] : blank;
        lab[s] = v;
        if ((s & 1) && s >= 3 && labels[s >> 1] == labels[(s >> 1) - 1]) repeats++;
    }
    repeats = (int)wave_sum((float)repeats);
    __syncthreads();

    if (L + repeats > Tn || Tn <= 0) {   // infeasible alignment: warp-ctc reports cost 0, grad 0
        if (lane == 0) costs[n] = 0.f;
        if (g_n)
            for (int t = 0; t < Tn; ++t)
                for (int k = lane; k < C; k += 64) g_n[t * tstride + k] = 0.f;
        return;
    }

    // ---- pass 1: log-sum-exp of every frame (softmax denominator), parked at alpha_ws[t][SMAX]
    for (int t = 0; t < Tn; ++t) {
        const float* row = a_n + t * tstride;
        float m = NEG_INF;
        for (int k = lane; k < C; k += 64) m = fmaxf(m, row[k]);
        m = wave_max(m);
        float sum = 0.f;
        for (int k = lane; k < C; k += 64) sum += expf(row[k] - m);
        sum = wave_sum(sum);
        if (lane == 0) alpha_ws[(size_t)t * (SMAX + 1) + SMAX] = m + logf(sum);
    }
    __syncthreads();

    // per-slot constants
    int my_lab[VT];
    bool skip_fwd[VT], skip_bwd[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        int s = lane + 64 * v;
        my_lab[v] = (s < S) ? lab[s] : blank;
        skip_fwd[v] = (s < S) && (s >= 2) && (my_lab[v] != blank) && (my_lab[v] != lab[s - 2]);
        skip_bwd[v] = (s + 2 < S) && (lab[s + 2] != blank) && (lab[s + 2] != my_lab[v]);
    }

    // ---- pass 2: alpha recursion, rows kept in the workspace
    float* cur = rowbuf + 2;                 // cur[-1], cur[-2] are guard slots (= -inf)
    float* nxt = rowbuf + (SMAX + 2) + 2;
    if (lane < 2) { cur[-1 - lane] = NEG_INF; nxt[-1 - lane] = NEG_INF; }
    {
        float lse0 = alpha_ws[SMAX];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            int s = lane + 64 * v;
            if (s < S) {
                float a0 = (s < 2) ? (a_n[my_lab[v]] - lse0) : NEG_INF;
                cur[s] = a0;
                alpha_ws[s] = a0;
            }
        }
    }
    __syncthreads();
    for (int t = 1; t < Tn; ++t) {
        const float* row = a_n + t * tstride;
        float lse_t = alpha_ws[(size_t)t * (SMAX + 1) + SMAX];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            int s = lane + 64 * v;
            if (s < S) {
                float p0 = cur[s], p1 = cur[s - 1];
                float p2 = skip_fwd[v] ? cur[s - 2] : NEG_INF;
                float a = lse3(p0, p1, p2);
                if (a != NEG_INF) a += row[my_lab[v]] - lse_t;
                nxt[s] = a;
                alpha_ws[(size_t)t * (SMAX + 1) + s] = a;
            }
        }
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    // log p(l|x)
    float logp;
    {
        float e1 = cur[S - 1];
        float e2 = (S >= 2) ? cur[S - 2] : NEG_INF;
        logp = lse2(e1, e2);
    }
    if (lane == 0) costs[n] = -logp;
    if (!g_n) return;
    __syncthreads();

    // ---- pass 3: beta recursion fused with the gradient rows (t = Tn-1 .. 0)
    // rows here carry 2 guard slots BEHIND the data (s = S, S+1) instead of in front
    float* bcur = rowbuf;
    float* bnxt = rowbuf + (SMAX + 2);
    for (int t = Tn - 1; t >= 0; --t) {
        const float* row = a_n + t * tstride;
        float lse_t = alpha_ws[(size_t)t * (SMAX + 1) + SMAX];
        for (int k = lane; k < C; k += 64) acc[k] = 0.f;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            int s = lane + 64 * v;
            if (s < S) {
                float ly = row[my_lab[v]] - lse_t;
                float b;
                if (t == Tn - 1) {
                    b = (s >= S - 2) ? ly : NEG_INF;
                } else {
                    float q0 = bcur[s];
                    float q1 = (s + 1 < S) ? bcur[s + 1] : NEG_INF;
                    float q2 = skip_bwd[v] ? bcur[s + 2] : NEG_INF;
                    b = lse3(q0, q1, q2);
                    if (b != NEG_INF) b += ly;
                }
                bnxt[s] = b;
                float al = alpha_ws[(size_t)t * (SMAX + 1) + s];
                if (al != NEG_INF && b != NEG_INF) {
                    float gamma = expf(al + b - ly - logp);   // posterior of (t, s), <= 1
                    atomicAdd(&acc[my_lab[v]], gamma);
                }
            }
        }
        __syncthreads();
        float* grow = g_n + t * tstride;
        for (int k = lane; k < C; k += 64) grow[k] = expf(row[k] - lse_t) - acc[k];
        float* tmp = bcur; bcur = bnxt; bnxt = tmp;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fast path (S = 2L+1 <= 64 and 3*T*S floats fit in LDS — every configuration of the reference): 4 waves per sample.
//   phase 1  all waves: softmax denominators of their frames (t = wave, wave+4, ...)
//   phase 2  all threads: logy[t][s] = act[t][l'_s] - lse[t] gathered into LDS — the recursion below never waits on HBM/L2
//   phase 3  wave 0 runs the alpha recursion, wave 1 the beta recursion CONCURRENTLY; the row lives in registers and the
//            s-1 / s-2 (s+1 / s+2) neighbours come from wave shuffles, so a step is ~60 VALU cycles with no barrier
//   phase 4  all waves: gradient rows of their frames (posterior per class via LDS float atomics, coalesced row store)
// The first version did all of this in one wave with a barrier and an L2 round trip per step: 3 x 63 serial steps,
// ~0.8 us each (152 us at T = 63).
// ------------------------------------------------------------------------------------------------------------------
__device__ long long* g_ctc_dbg = nullptr;
constexpr int CTC_NW = 16;        // waves per sample: the frame-parallel phases (log-sum-exp rows, gradient rows) are latency chains per
                                  // row (cross-lane reductions, LDS atomics): 16 waves x 4 rows instead of 4 x 16 rows
__global__ __launch_bounds__(64 * CTC_NW) void ctc_fast_kernel(
    const float* __restrict__ act, float* __restrict__ grad, const int* __restrict__ flat_labels,
    const int* __restrict__ label_off, const int* __restrict__ label_len, const int* __restrict__ input_len,
    int T, int N, int C, int blank, float* __restrict__ costs, int SMAX,
    bf16_t* __restrict__ grad_ntc /* optional: bf16 [N][T][C] = scale * gradient (the layout/dtype the backward GEMMs read) */,
    float scale) {
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* lse = fsm;                          // [T]
    float* logy = lse + T;                     // [T][SMAX]
    float* alpha = logy + T * SMAX;            // [T][SMAX]
    float* beta = alpha + T * SMAX;            // [T][SMAX]
    float* acc = beta + T * SMAX;              // [CTC_NW][C]   per-wave posterior accumulators
    int* lab = (int*)(acc + CTC_NW * C);       // [SMAX]
    __shared__ float s_logp;
    __shared__ int s_repeats, s_off;

    const int L = label_len[n];
    const int Tn = min(input_len[n], T);
    const int S = 2 * L + 1;
    const size_t tstride = (size_t)N * C;
    const float* a_n = act + (size_t)n * C;
    float* g_n = grad ? grad + (size_t)n * C : nullptr;
    bf16_t* gb_n = grad_ntc ? grad_ntc + (size_t)n * T * C : nullptr;
    const bool want_grad = g_n || gb_n;

    if (tid == 0) { s_repeats = 0; s_off = 0; }
    __syncthreads();
    if (label_off == nullptr) {            // exclusive prefix sum of the label lengths, done here instead of a scan launch
        int part = 0;
        for (int i = tid; i < n; i += 64 * CTC_NW) part += label_len[i];
        if (part) atomicAdd(&s_off, part);
        __syncthreads();
    }
    const int* labels = flat_labels + (label_off ? label_off[n] : s_off);
    int rep = 0;
    for (int s = tid; s < S; s += 64 * CTC_NW) {
        lab[s] = (s & 1) ? labels[s >> 1] : blank;
        if ((s & 1) && s >= 3 && labels[s >> 1] == labels[(s >> 1) - 1]) rep++;
    }
    if (rep) atomicAdd(&s_repeats, rep);
    __syncthreads();
    const bool feasible = (Tn > 0) && (L + s_repeats <= Tn);
    if (want_grad) {    // frames this sample does not own (and everything when infeasible): zero gradient
        const int t0 = feasible ? Tn : 0;
        for (int t = t0 + wave; t < T; t += CTC_NW)
            for (int k = lane; k < C; k += 64) {
                if (g_n) g_n[t * tstride + k] = 0.f;
                if (gb_n) gb_n[(size_t)t * C + k] = 0;
            }
    }
    if (!feasible) { if (tid == 0) costs[n] = 0.f; return; }
#define CSTAMP(k) do { if (g_ctc_dbg && n == 0 && tid == 0) g_ctc_dbg[k] = wall_clock64(); } while (0)
    CSTAMP(0);

    // phase 1.  A wave owns frames wave, wave+16, ...; when they fit (T <= 64, C <= 128) ALL its activation rows are loaded
    // first — independent loads, one memory round trip — and stay in registers for phase 4.  (Row after row, each frame paid
    // a dependent global-load latency: 12 us of the 41 us kernel here, 7.5 more in phase 4.)
    constexpr int MAXR = 4, MAXK = 2;
    const bool cached = (T <= CTC_NW * MAXR) && (C <= 64 * MAXK);
    float rowv[MAXR][MAXK];
    if (cached) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i)
#pragma unroll
            for (int j = 0; j < MAXK; ++j) {
                const int t = wave + CTC_NW * i, k = lane + 64 * j;
                rowv[i][j] = (t < Tn && k < C) ? a_n[t * tstride + k] : NEG_INF;
            }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int t = wave + CTC_NW * i;
            float m = fmaxf(rowv[i][0], rowv[i][1]);
            m = wave_max(m);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < MAXK; ++j) sum += (rowv[i][j] == NEG_INF) ? 0.f : expf(rowv[i][j] - m);
            sum = wave_sum(sum);
            if (lane == 0 && t < Tn) lse[t] = m + logf(sum);
        }
    } else
    for (int t = wave; t < Tn; t += CTC_NW) {
        const float* row = a_n + t * tstride;
        float m = NEG_INF;
        for (int k = lane; k < C; k += 64) m = fmaxf(m, row[k]);
        m = wave_max(m);
        float sum = 0.f;
        for (int k = lane; k < C; k += 64) sum += expf(row[k] - m);
        sum = wave_sum(sum);
        if (lane == 0) lse[t] = m + logf(sum);
    }
    __syncthreads();
    CSTAMP(1);
    // phase 2
    for (int i = tid; i < Tn * S; i += 64 * CTC_NW) {
        const int t = i / S, s = i - t * S;
        logy[t * SMAX + s] = a_n[t * tstride + lab[s]] - lse[t];
    }
    __syncthreads();
    CSTAMP(2);
    // phase 3
    if (wave == 0) {
        const int s = lane;
        const bool in = s < S;
        const int my = in ? lab[s] : blank;
        const bool skip = in && s >= 2 && my != blank && my != lab[s - 2];
        float a = (in && s < 2) ? logy[s] : NEG_INF;
        if (in) alpha[s] = a;
        for (int t = 1; t < Tn; ++t) {
            float p1 = wave_shr1(a, NEG_INF), p2 = wave_shr1(p1, NEG_INF);
            if (!skip) p2 = NEG_INF;
            float v = lse3_fast(a, p1, p2);
            if (v != NEG_INF && in) v += logy[t * SMAX + s]; else if (!in) v = NEG_INF;
            a = v;
            if (in) alpha[t * SMAX + s] = a;
        }
        float e1 = __shfl(a, S - 1, 64);
        float e2 = (S >= 2) ? __shfl(a, S - 2, 64) : NEG_INF;
        if (lane == 0) { s_logp = lse2(e1, e2); costs[n] = -s_logp; }
    } else if (wave == 1 && want_grad) {
        const int s = lane;
        const bool in = s < S;
        const int my = in ? lab[s] : blank;
        const bool skip = (s + 2 < S) && lab[s + 2] != blank && lab[s + 2] != my;
        float b = (in && s >= S - 2) ? logy[(Tn - 1) * SMAX + s] : NEG_INF;
        if (in) beta[(Tn - 1) * SMAX + s] = b;
        for (int t = Tn - 2; t >= 0; --t) {
            float q1 = wave_shl1(b, NEG_INF), q2 = wave_shl1(q1, NEG_INF);
            if (s + 1 >= S) q1 = NEG_INF;
            if (!skip) q2 = NEG_INF;
            float v = lse3_fast(b, q1, q2);
            if (v != NEG_INF && in) v += logy[t * SMAX + s]; else if (!in) v = NEG_INF;
            b = v;
            if (in) beta[t * SMAX + s] = b;
        }
    }
    __syncthreads();
    CSTAMP(3);
    if (!want_grad) return;
    // phase 4
    const float logp = s_logp;
    float* wacc = acc + wave * C;
    if (cached) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int t = wave + CTC_NW * i;
            if (t >= Tn) break;                       // uniform per wave
            for (int k = lane; k < C; k += 64) wacc[k] = 0.f;
            if (lane < S) {
                const float al = alpha[t * SMAX + lane], be = beta[t * SMAX + lane];
                if (al != NEG_INF && be != NEG_INF) atomicAdd(&wacc[lab[lane]], expf(al + be - logy[t * SMAX + lane] - logp));
            }
            const float l = lse[t];
#pragma unroll
            for (int j = 0; j < MAXK; ++j) {
                const int k = lane + 64 * j;
                if (k < C) {
                    const float v = expf(rowv[i][j] - l) - wacc[k];
                    if (g_n) g_n[t * tstride + k] = v;
                    if (gb_n) gb_n[(size_t)t * C + k] = f2bf(v * scale);
                }
            }
        }
    } else
    for (int t = wave; t < Tn; t += CTC_NW) {
        for (int k = lane; k < C; k += 64) wacc[k] = 0.f;
        if (lane < S) {
            const float al = alpha[t * SMAX + lane], be = beta[t * SMAX + lane];
            if (al != NEG_INF && be != NEG_INF) atomicAdd(&wacc[lab[lane]], expf(al + be - logy[t * SMAX + lane] - logp));
        }
        const float* row = a_n + t * tstride;
        const float l = lse[t];
        for (int k = lane; k < C; k += 64) {
            const float v = expf(row[k] - l) - wacc[k];
            if (g_n) g_n[t * tstride + k] = v;
            if (gb_n) gb_n[(size_t)t * C + k] = f2bf(v * scale);
        }
    }
    CSTAMP(4);
}

// Greedy (best-path) decode: per frame argmax (lowest index wins ties, like numpy.argmax),
// collapse repeats, drop `blank`. Output is dense [N, T] padded with `pad_value`, plus lengths.
__global__ __launch_bounds__(64) void ctc_greedy_kernel(
    const float* __restrict__ act, const int* __restrict__ input_len, int T, int N, int C,
    int blank, int pad_value, int* __restrict__ out, int* __restrict__ out_len) {
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ int am[64];
    const int Tn = min(input_len[n], T);
    int* o = out + (size_t)n * T;
    int prev = -1;
    int count = 0;
    for (int t0 = 0; t0 < Tn; t0 += 64) {
        int tcnt = min(64, Tn - t0);
        for (int tt = 0; tt < tcnt; ++tt) {
            const float* row = act + ((size_t)(t0 + tt) * N + n) * C;
            float bv = NEG_INF;
            int bi = 0x7fffffff;
            for (int k = lane; k < C; k += 64) {
                float v = row[k];
                if (v > bv) { bv = v; bi = k; }   // strict > keeps the lowest index per lane
            }
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) {
                float ov = __shfl_xor(bv, o2, 64);
                int oi = __shfl_xor(bi, o2, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) am[tt] = bi;
        }
        __syncthreads();
        int a = (lane < tcnt) ? am[lane] : blank;
        int before = __shfl_up(a, 1, 64);
        if (lane == 0) before = prev;
        bool keep = (lane < tcnt) && (a != blank) && (a != before);
        unsigned long long mask = __ballot(keep);
        int pos = __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) o[count + pos] = a;
        count += __popcll(mask);
        prev = am[tcnt - 1];
        __syncthreads();
    }
    for (int j = count + lane; j < T; j += 64) o[j] = pad_value;
    if (lane == 0) out_len[n] = count;
}

__global__ void exclusive_scan_small(const int* __restrict__ in, int* __restrict__ out, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < n; ++i) { out[i] = s; s += in[i]; }
    }
}

// ------------------------------------------------------------------------------------------
// C ABI (declared in include/ocr_hip.h)
// ------------------------------------------------------------------------------------------
static inline int smax_for(int max_label_len) { return 2 * max_label_len + 1; }
static int g_ctc_fast = 1;
extern "C" int ocr_set_ctc_engine(int fast) { g_ctc_fast = fast; return OCR_OK; }   // A/B + test knob: 0 = one-wave reference kernel

// the [T][S] tables of the fast kernel may need more than the 64 KiB of dynamic LDS a kernel gets by default (T = 79 of the
// variable-width workload: 65 KiB); the limit is raised once
constexpr size_t CTC_FAST_LDS_MAX = 128 * 1024;
static bool ctc_fast_allow_lds() {
    static int ok = -1;
    if (ok < 0) ok = hipFuncSetAttribute((const void*)ctc_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CTC_FAST_LDS_MAX) == hipSuccess;
    return ok == 1;
}

extern "C" int ocr_ctc_workspace_size(int max_label_len, int max_time, int minibatch,
                                      size_t* bytes) {
    if (!bytes || max_label_len < 0 || max_time <= 0 || minibatch <= 0) return OCR_ERR_INVALID;
    size_t smax = (size_t)smax_for(max_label_len);
    // alpha rows (+1 slot per row for the frame log-sum-exp) + label offsets
    *bytes = (size_t)minibatch * max_time * (smax + 1) * sizeof(float) +
             (size_t)minibatch * sizeof(int);
    *bytes = (*bytes + 255) & ~(size_t)255;
    return OCR_OK;
}

extern "C" int ocr_ctc_loss(const float* activations, float* gradients, const int* flat_labels,
                            const int* label_lengths, const int* input_lengths,
                            int alphabet_size, int minibatch, int max_time, int max_label_len,
                            int blank_label, float* costs, void* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!activations || !flat_labels || !label_lengths || !input_lengths || !costs || !workspace)
        return OCR_ERR_INVALID;
    if (alphabet_size <= 0 || minibatch <= 0 || max_time <= 0 || max_label_len < 0 ||
        blank_label < 0 || blank_label >= alphabet_size)
        return OCR_ERR_INVALID;
    const int SMAX = smax_for(max_label_len);
    if (SMAX > 256) return OCR_ERR_INVALID;   // 4 slots per lane is the largest instantiation
    float* ws = (float*)workspace;
    int* label_off = (int*)(ws + (size_t)minibatch * max_time * (SMAX + 1));
    exclusive_scan_small<<<1, 64, 0, stream>>>(label_lengths, label_off, minibatch);
    OCR_CHECK_LAUNCH();
    // fast path: S <= 64 and the three [T][S] tables fit in LDS
    {
        size_t flds = ((size_t)max_time * (1 + 3 * SMAX) + CTC_NW * (size_t)alphabet_size) * sizeof(float) + (size_t)SMAX * sizeof(int);
        flds = (flds + 15) & ~(size_t)15;
        if (SMAX <= 64 && flds <= CTC_FAST_LDS_MAX && g_ctc_fast && ctc_fast_allow_lds()) {
            ctc_fast_kernel<<<minibatch, 64 * CTC_NW, flds, stream>>>(activations, gradients, flat_labels, label_off, label_lengths,
                                                             input_lengths, max_time, minibatch, alphabet_size, blank_label, costs,
                                                             SMAX, nullptr, 1.0f);
            OCR_CHECK_LAUNCH();
            return OCR_OK;
        }
    }
    size_t lds = (size_t)(2 * (SMAX + 2) + alphabet_size) * sizeof(float) + (size_t)SMAX * sizeof(int);
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return OCR_ERR_INVALID;
#define LAUNCH_CTC(VT)                                                                         \
    ctc_loss_grad_kernel<VT><<<minibatch, 64, lds, stream>>>(                                  \
        activations, gradients, flat_labels, label_off, label_lengths, input_lengths, max_time, \
        minibatch, alphabet_size, blank_label, costs, ws, SMAX)
    if (SMAX <= 64) LAUNCH_CTC(1);
    else if (SMAX <= 128) LAUNCH_CTC(2);
    else LAUNCH_CTC(4);
#undef LAUNCH_CTC
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

static size_t ctc_fast_lds(int SMAX, int max_time, int alphabet_size) {
    size_t flds = ((size_t)max_time * (1 + 3 * SMAX) + CTC_NW * (size_t)alphabet_size) * sizeof(float) + (size_t)SMAX * sizeof(int);
    return (flds + 15) & ~(size_t)15;
}
// Training form: costs + the gradient ALREADY in the layout / dtype / scale the backward GEMMs consume
// (bf16 [N][T][C], multiplied by `scale` = 1/(batch*world)); label offsets are computed in-kernel.  One launch instead of
// scan + loss + transpose.  Covers S = 2L+1 <= 64 with the [T][S] tables in LDS; returns OCR_ERR_INVALID otherwise
// (callers then use ocr_ctc_loss + ocr_tnc_to_ntc_bf16).
extern "C" int ocr_ctc_debug(void* p) { long long* q = (long long*)p; return hipMemcpyToSymbol(HIP_SYMBOL(g_ctc_dbg), &q, sizeof(q)) == hipSuccess ? OCR_OK : OCR_ERR_MEMOPS; }
extern "C" int ocr_ctc_train_supported(int alphabet_size, int max_time, int max_label_len) {
    const int SMAX = smax_for(max_label_len);
    return SMAX <= 64 && ctc_fast_lds(SMAX, max_time, alphabet_size) <= CTC_FAST_LDS_MAX && ctc_fast_allow_lds();
}
extern "C" int ocr_ctc_loss_train(const float* activations, void* grad_ntc_bf16, float scale, const int* flat_labels,
                                  const int* label_lengths, const int* input_lengths, int alphabet_size, int minibatch,
                                  int max_time, int max_label_len, int blank_label, float* costs, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!activations || !grad_ntc_bf16 || !flat_labels || !label_lengths || !input_lengths || !costs) return OCR_ERR_INVALID;
    if (alphabet_size <= 0 || minibatch <= 0 || max_time <= 0 || max_label_len < 0 || blank_label < 0 ||
        blank_label >= alphabet_size || !ocr_ctc_train_supported(alphabet_size, max_time, max_label_len))
        return OCR_ERR_INVALID;
    const int SMAX = smax_for(max_label_len);
    ctc_fast_kernel<<<minibatch, 64 * CTC_NW, ctc_fast_lds(SMAX, max_time, alphabet_size), stream>>>(
        activations, nullptr, flat_labels, nullptr, label_lengths, input_lengths, max_time, minibatch, alphabet_size, blank_label,
        costs, SMAX, (bf16_t*)grad_ntc_bf16, scale);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" int ocr_ctc_greedy_decode(const float* activations, const int* input_lengths,
                                     int alphabet_size, int minibatch, int max_time,
                                     int blank_label, int pad_value, int* decoded,
                                     int* decoded_lengths, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!activations || !input_lengths || !decoded || !decoded_lengths) return OCR_ERR_INVALID;
    if (alphabet_size <= 0 || minibatch <= 0 || max_time <= 0) return OCR_ERR_INVALID;
    ctc_greedy_kernel<<<minibatch, 64, 0, stream>>>(activations, input_lengths, max_time, minibatch,
                                                    alphabet_size, blank_label, pad_value, decoded,
                                                    decoded_lengths);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
