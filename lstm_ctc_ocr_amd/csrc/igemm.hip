// Large-tile bf16 MFMA engine for gfx950 (second generation of gemm.hip's kernel, same contract):
//     out[m][n] = sum_k P[m][k] * Q[n][k]      P = activations / pixels, Q = weights, both K-contiguous
// used for the 3x3 SAME implicit-GEMM convolutions (forward and data gradient — reference
// lib/networks/network.py:160-191 via tf.nn.conv2d) and for the large plain GEMMs.
//
// What changed against gemm.hip and why (measured there: 370-570 TFLOP/s, every pipe ~equally loaded —
// MFMA 256, LDS 340, L1 fill 256, TA 256 cycles per 128x128x32 step — so nothing overlapped well):
//   * block tile 256 (m) x BN (n in {64,128,256}), BK = 64, 8 waves (512 threads): operand bytes per flop halve,
//     MFMA becomes the longest pipe (1024 cyc/SIMD/step vs 512 LDS-read, 768 L1-fill at BN = 128);
//   * global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave instruction = 8 rows x 128 B): no staging
//     VGPRs, no ds_write pass; two LDS stages, the next K tile streams in while the current one feeds the MFMAs;
//   * the LDS image is row-linear (DMA writes lane-linearly), so the XOR swizzle that makes the 16-row ds_read_b128
//     fragment reads conflict-free is applied to the SOURCE chunk (chunk ^ (row & 7)) and again on the read;
//   * SAME-padding halo and M/N tails read from a 64-B zero page instead of branching (DMA cannot skip a lane);
//   * workgroup ids are remapped so each XCD (private 4 MiB L2) owns a contiguous range of tiles: the 9 taps and
//     the neighbouring m-tiles re-read the same activation rows out of that XCD's L2.
// MFMA orientation and epilogue are unchanged: A := Q tile, B := P tile, each lane owns 4 consecutive n of one m.
#include "common.h"
#include <stdlib.h>

enum { IG_BIAS = 1, IG_RELU = 2, IG_OUT_F32 = 4, IG_MASK = 16, IG_ROWSWAP = 32 };

struct IgArgs {
    const bf16_t* P; const bf16_t* Q;
    long ldp, ldq;
    int M, N, K;              // K % 64 == 0
    int grp, skip;            // plain mode row groups
    int cW, cH, cC;           // conv mode geometry, cC % 64 == 0
    void* out; long ldo;
    const float* bias; const bf16_t* mask; long ldmask;
    int flags, swap_inner, swap_outer;
};

__device__ u32x4 ig_zero_page[4];   // 64 B of zeros (device globals are zero-initialised)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int MODE /*0 plain, 1 conv3x3*/, int STAGES /*2 or 3 LDS stages*/, int NW /*waves: 8 (256-row tiles) or 4 (128-row tiles, two workgroups per CU)*/>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm_kernel(IgArgs g) {
    constexpr int BM = 32 * NW, BK = 64;
    constexpr int WAVES_N = BN / 64, WAVES_M = NW / WAVES_N;
    constexpr int WM = BM / WAVES_M;               // 128 / 64 / 32
    constexpr int FM = WM / 16, FN = 4;
    constexpr int PB = BM * BK * 2, QB = BN * BK * 2, STAGE = PB + QB;
    constexpr int QI = BN / (8 * NW);              // Q DMA instructions per wave per stage (P: always 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile order: consecutive logical tiles live on one XCD
    const int mtiles = (g.M + BM - 1) / BM, ntiles = (g.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int L = blockIdx.x;
    if ((nblk & 7) == 0) L = (L & 7) * (nblk >> 3) + (L >> 3);
    const int m0 = (L / ntiles) * BM, n0 = (L % ntiles) * BN;

    // ---- per-thread DMA source rows (fixed over the K loop); LDS chunk position lane&7 holds source chunk (lane&7)^(lane>>3)
    const int rsub = lane >> 3;
    const int csrc = ((lane & 7) ^ rsub) * 8;       // element offset of the 16-B source chunk inside the 64-wide K tile
    const bf16_t* zero = (const bf16_t*)ig_zero_page;
    const bf16_t* prow[4];
    int pw[4], ph[4];
    bool pok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + (wave * 4 + j) * 8 + rsub;
        pok[j] = m < g.M;
        int mm = pok[j] ? m : 0;
        if (MODE == 0) {
            long phys = (long)mm + (g.grp > 0 ? (long)(mm / g.grp) * g.skip : 0);
            prow[j] = g.P + phys * g.ldp + csrc;
            pw[j] = ph[j] = 0;
        } else {
            ph[j] = mm % g.cH;
            pw[j] = (mm / g.cH) % g.cW;
            prow[j] = g.P + (long)mm * g.cC + csrc;
        }
    }
    const bf16_t* qrow[QI];
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        int n = n0 + (wave * QI + j) * 8 + rsub;
        qrow[j] = (n < g.N) ? g.Q + (long)n * g.ldq + csrc : nullptr;
    }

    auto stage_load = [&](int k0, int buf) {
        unsigned char* sp = smem + buf * STAGE;
        long shift = k0;
        int dw = 0, dh = 0;
        if (MODE == 1) {
            int tap = k0 / g.cC;                    // uniform: cC % 64 == 0 keeps a K tile inside one tap
            dw = tap / 3 - 1; dh = tap % 3 - 1;
            shift = ((long)dw * g.cH + dh) * g.cC + (k0 - tap * g.cC);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = pok[j];
            if (MODE == 1) ok = ok && (unsigned)(pw[j] + dw) < (unsigned)g.cW && (unsigned)(ph[j] + dh) < (unsigned)g.cH;
            const bf16_t* src = ok ? prow[j] + shift : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sp + (wave * 4 + j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const bf16_t* src = qrow[j] ? qrow[j] + k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sp + PB + (wave * QI + j) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row (lane&15) of a 16-row block, 16-B chunk (kk*4 + lane>>4) ^ (row & 7)
    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int ntk = g.K / BK;
    constexpr int NI = 4 + QI;        // DMA instructions one wave issues per stage
    // All fragment reads of a K tile are issued back to back as inline asm (the compiler neither counts nor waits for them),
    // then one counted wait per K-half: the first half's FN + FM fragments were issued first and LDS returns in order.
    // (hipcc's own schedule interleaved small read groups with the MFMAs and drained lgkmcnt(0) several times per tile.)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned qoff = PB + (wn * 64 + frow) * 128 + ((fq ^ fx) << 4);     // ^ 64 for the second K-half; + a * 2048
    const unsigned poff = (wm * WM + frow) * 128 + ((fq ^ fx) << 4);          // + b * 2048
    auto compute = [&](int buf) {
        const unsigned base = lds0 + buf * STAGE;
        u32x4 af[2][FN], bfr[2][FM];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int a = 0; a < FN; ++a)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kk][a]) : "v"(base + (qoff ^ (kk * 64))), "n"(a * 2048));
#pragma unroll
            for (int b = 0; b < FM; ++b)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[kk][b]) : "v"(base + (poff ^ (kk * 64))), "n"(b * 2048));
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FN + FM) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kk][a]),
                                                                        __builtin_bit_cast(bf16x8, bfr[kk][b]), acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (STAGES == 2) {
        stage_load(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < ntk; ++t) {
            const int cur = t & 1;
            if (t + 1 < ntk) stage_load((t + 1) * BK, cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage has landed (this wave's DMA pieces)
            __syncthreads();                                     // everyone's pieces landed, everyone done reading `cur`
        }
    } else {
        // three stages, two K tiles in flight: the wait at the top of step t only retires stage t (counted vmcnt leaves the
        // NI DMA instructions of stage t+1 outstanding), and the barrier is a RAW s_barrier — __syncthreads() would drain
        // the DMA queue (an LDS-DMA is a pending LDS write on the VM counter).  The same barrier orders "everyone finished
        // reading stage t-1" before stage t+2 is streamed into that buffer.
        stage_load(0, 0);
        if (ntk > 1) stage_load(BK, 1);
        int cur = 0;
        for (int t = 0; t < ntk; ++t) {
            if (t + 1 < ntk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 2 < ntk) { int nb = cur + 2; if (nb >= 3) nb -= 3; stage_load((t + 2) * BK, nb); }
            compute(cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of `cur` are complete
            cur = (cur == 2) ? 0 : cur + 1;
        }
    }

    // ---- epilogue: lane owns n = nb + (lane>>4)*4 .. +3 at m = mb + (lane&15).  Loop order m-block outer / n-fragment inner:
    // the FN stores that complete one 128-byte run of a pixel row are issued back to back (with n outer the same lines
    // were revisited FM stores apart and reached HBM as partial-line writes — WRITE_SIZE 2.5x the output bytes).
    const int flags = g.flags;
    f32x4 bv[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
        bv[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((flags & IG_BIAS) && n < g.N) bv[a] = *(const f32x4*)(g.bias + n);
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * WM + b * 16 + (lane & 15);
        if (m >= g.M) continue;
        long orow = m;
        if (flags & IG_ROWSWAP) orow = (long)(m % g.swap_inner) * g.swap_outer + m / g.swap_inner;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            f32x4 v = acc[a][b] + bv[a];
            if (flags & IG_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (flags & IG_MASK) {
                u32x2 mk = *(const u32x2*)(g.mask + (long)m * g.ldmask + n);
                if (!(bf_lo(mk.x) > 0.f)) v.x = 0.f;
                if (!(bf_hi(mk.x) > 0.f)) v.y = 0.f;
                if (!(bf_lo(mk.y) > 0.f)) v.z = 0.f;
                if (!(bf_hi(mk.y) > 0.f)) v.w = 0.f;
            }
            if (flags & IG_OUT_F32) {
                *(f32x4*)((float*)g.out + orow * g.ldo + n) = v;
            } else {
                u32x2 pk;
                pk.x = pack_bf2(v.x, v.y);
                pk.y = pack_bf2(v.z, v.w);
                *(u32x2*)((bf16_t*)g.out + orow * g.ldo + n) = pk;
            }
        }
    }
}

int g_ig_stages = 3;     // A/B knob (ocr_set_gemm_engine): 3 = three-stage counted-vmcnt pipeline where LDS allows, 2 = two-stage

template <int BN, int MODE, int STAGES, int NW>
static int launch_ig2(const IgArgs& g, hipStream_t stream) {
    constexpr int BM = 32 * NW;
    constexpr int LDS = STAGES * (BM * 128 + BN * 128);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)igemm_kernel<BN, MODE, STAGES, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return OCR_ERR_EXEC;
        attr_set = true;
    }
    int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    igemm_kernel<BN, MODE, STAGES, NW><<<mt * nt, 64 * NW, LDS, stream>>>(g);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
template <int BN, int MODE>
static int launch_ig(const IgArgs& g, hipStream_t stream) {
    if (BN <= 128 && g_ig_stages == 3) return launch_ig2<(BN <= 128 ? BN : 128), MODE, 3, 8>(g, stream);   // 3 x 48 KiB fits, 3 x 64 KiB does not
    return launch_ig2<BN, MODE, 2, 8>(g, stream);
}

// Shared with gemm.hip's entry points: returns -1 when the shape is not covered (caller uses the small-tile kernel).
int ig_try_dispatch(const void* P, long ldp, const void* Q, long ldq, void* out, long ldo, int M, int N, int K,
                    const float* bias, const void* mask, long ldmask, int flags, int mode, int grp, int skip, int cW,
                    int cH, int cC, int swap_inner, int swap_outer, hipStream_t stream) {
    if ((K & 63) || (N & 3) || M < 1024) return -1;
    if (mode == 1 && (cC & 63)) return -1;
    if (flags & ~(IG_BIAS | IG_RELU | IG_OUT_F32 | IG_MASK | IG_ROWSWAP)) return -1;   // atomics / accumulate: old kernel
    IgArgs g = {};
    g.P = (const bf16_t*)P; g.Q = (const bf16_t*)Q; g.ldp = ldp; g.ldq = ldq; g.M = M; g.N = N; g.K = K;
    g.grp = grp; g.skip = skip; g.cW = cW; g.cH = cH; g.cC = cC; g.out = out; g.ldo = ldo; g.bias = bias;
    g.mask = (const bf16_t*)mask; g.ldmask = ldmask; g.flags = flags; g.swap_inner = swap_inner; g.swap_outer = swap_outer;
    // 128-row tiles of 4 waves (72 KiB of LDS with three stages at BN = 64, 64 KiB with two at BN = 128: two workgroups per
    // CU) when the 256-row grid cannot give every CU two waves' worth of work — the M = 4032 GEMMs of conv5 / LSTM / FC.
    static int ig_nw = -1;                                 // A/B knob OCR_IG_NW: 8 / 4 force, unset = by grid size
    if (ig_nw < 0) { const char* e = ocr_tune_env("OCR_IG_NW"); ig_nw = e ? atoi(e) : 0; }
    if (mode == 0 && ig_nw != 8) {
        const long mt4 = (M + 127) / 128;
        const long wg8 = (long)((M + 255) / 256) * ((N + 127) / 128);
        if (ig_nw == 4 || wg8 < 400) {
            if (N >= 128 && mt4 * ((N + 127) / 128) >= 448) return launch_ig2<128, 0, 2, 4>(g, stream);
            return launch_ig2<64, 0, 3, 4>(g, stream);
        }
    }
    const int mt = (M + 255) / 256;
    int bn = 64;                                           // widest n-tile that still gives every CU a workgroup
    if (N >= 256 && (long)mt * ((N + 255) / 256) >= 256) bn = 256;
    else if (N >= 128 && (long)mt * ((N + 127) / 128) >= 256) bn = 128;
    else if (N >= 128 && (long)mt * ((N + 63) / 64) < 96) bn = 128;    // tiny grid either way: fewer, fatter tiles
#define IG_CASE(BNV)                                                                  \
    if (bn == BNV) return mode == 1 ? launch_ig<BNV, 1>(g, stream) : launch_ig<BNV, 0>(g, stream);
    IG_CASE(256) IG_CASE(128) IG_CASE(64)
#undef IG_CASE
    return -1;
}
