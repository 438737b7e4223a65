// 3x3 SAME implicit-GEMM convolution with TAP REUSE for gfx950 (third generation; same contract as
// ocr_conv3x3_bf16 — reference lib/networks/network.py:160-191 forward, and its data gradient).
//
// igemm.hip streams a fresh 256-pixel x 64-channel activation tile for every (tap, channel chunk): the same
// activation rows cross L2 -> LDS nine times, and PMC shows the kernel limited by that on-chip traffic, not by
// MFMA (46 % pipe utilisation at 1 PFLOP/s).  Here a workgroup stages, per 64-channel chunk, ONE halo tile: the
// 256 output pixels are a contiguous run of the flat pixel index, and every tap of every one of them lies in the
// contiguous run [m0 - H - 1, m0 + 256 + H + 1) — so 256 + 2H + 2 rows of 128 B are DMA'd once and the nine taps
// are nine shifted views of that LDS image.  Per step (tap, chunk) only the 16 KiB weight tile is new:
// 20 KiB of LDS-DMA per 32 MFMAs per wave instead of 48 KiB.
// SAME-padding: whether a tap of an output pixel falls outside the image depends on (pixel, tap), so the halo
// image cannot be pre-zeroed; instead every lane carries a 9-bit validity mask per pixel fragment and selects
// zeros for the MFMA operand (4 v_cndmask per fragment).
// Pipeline: weight tiles double-buffered (one step ahead), halo tiles double-buffered (one chunk ahead, issued at
// tap 0 AFTER the weight DMA so the counted wait of tap 1 does not have to drain it), raw s_barrier per step.
#include "common.h"
#include <stdlib.h>
#include <string.h>

enum { IGH_BIAS = 1, IGH_RELU = 2, IGH_MASK = 16, IGH_ACCUM = 64 /* out (bf16) += value: a data gradient added to what another consumer already delivered */ };

struct HaloArgs {
    const bf16_t* P; const bf16_t* Q;     // P [M pixels][C] ; Q [N][9*C]
    int M, N, C;                          // C % 64 == 0
    int cW, cH;
    bf16_t* out; const float* bias; const bf16_t* mask; int flags;
    bf16_t* pool; int pool_kind;          // max-pool of the (ReLU'd) output written by the same epilogue: 0 none, 1 feature pairs (1 x 2), 2 2 x 2
    int stagger, stagger_bit;             // experiment OCR_HALO_STAGGER=units,bit: workgroups whose in-XCD index has `bit` set start `units` x 64 clocks late
    int prio;                             // OCR_HALO_PRIO (default 1): s_setprio 1 around the MFMA clusters (guide T5)
};

__device__ u32x4 igh_zero_page[4];
// diagnostic (ocr_conv_halo_clock_debug): workgroup 0 stamps {shader-clock counter, 100 MHz wall clock} at entry and exit, so the
// tool can tell the clock the kernel really ran at (tools/clock_probe.py)
__device__ long long* g_halo_clk;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int NW /* waves per workgroup: 8 (256 pixels, one workgroup per CU) or 4 (128 pixels, two per CU) */,
          int ABL = 0 /* timing ablations (wrong results): 1 no fragment reads, 2 no DMA after the prologue, 5 neither (MFMA + barrier only) */,
          int RPW = 32 /* pixels of the tile per wave: 32, or 16 = "dense": 8 waves on a 128-pixel tile, two such workgroups per CU = 4 waves per SIMD */,
          int FNV = 4 /* 16-channel fragments per wave: 4 (64 x 64 wave tiles), or 8 = "wide": 64 x 128 wave tiles, 0.375 instead of 0.5 LDS fragment
                         reads per MFMA and half the weight DMA per flop, one 4-wave workgroup per CU */>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(FNV == 8 ? 1 : (RPW == 16 ? 4 : 2), FNV == 8 ? 1 : (RPW == 16 ? 4 : 2))))
void conv_halo_kernel(HaloArgs g, int NRpad /* halo rows rounded up to 8 * NW */) {
    constexpr int BM = RPW * NW;
    constexpr int FN = FNV;
    constexpr int WAVES_N = BN / (16 * FN), WAVES_M = NW / WAVES_N;
    constexpr int WM = BM / WAVES_M, FM = WM / 16;
    constexpr int QB = BN * 128, QI = BN / (8 * NW);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PBYTES = NRpad * 128;
    unsigned char* pbuf0 = smem;                       // 2 halo stages
    unsigned char* qbuf0 = smem + 2 * PBYTES;          // 2 weight stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int H = g.cH, C = g.C;
    long long* const clk = g_halo_clk;
    if (clk && blockIdx.x == 0 && tid == 0) ocr_clk_enter(clk);
    // co-resident workgroups of a launch start in phase (their prologue DMA, K steps and epilogue stores coincide instead of
    // overlapping — the short-K layers run at half their MFMA-only rate): optional start offset for every other workgroup of a CU
    if (g.stagger > 0 && ((blockIdx.x >> 3) & g.stagger_bit))
        for (int i = 0; i < g.stagger; i += 16) __builtin_amdgcn_s_sleep(16);
    const int NR = BM + 2 * H + 2;                     // halo rows actually needed
    const int PI = NRpad / (8 * NW);                   // halo DMA instructions per wave (8 rows each)

    const int mtiles = (g.M + BM - 1) / BM, ntiles = (g.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int L = blockIdx.x;
    if ((nblk & 7) == 0) L = (L & 7) * (nblk >> 3) + (L >> 3);
    const int m0 = (L / ntiles) * BM, n0 = (L % ntiles) * BN;

    const int rsub = lane >> 3;
    const int csrc = ((lane & 7) ^ rsub) * 8;          // LDS position lane&7 of row r holds source chunk (lane&7)^(r&7)
    const bf16_t* zero = (const bf16_t*)igh_zero_page;
    const long mfirst = (long)m0 - H - 1;              // flat pixel of halo row 0

    const bf16_t* qrow[QI];
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        int n = n0 + (wave * QI + j) * 8 + rsub;
        qrow[j] = (n < g.N) ? g.Q + (long)n * 9 * C + csrc : nullptr;
    }
    auto load_q = [&](int k0, int buf) {               // k0 = tap*C + chunk*64
        unsigned char* sq = qbuf0 + buf * QB;
#pragma unroll
        for (int j = 0; j < QI; ++j) {
            const bf16_t* src = qrow[j] ? qrow[j] + k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sq + (wave * QI + j) * 1024), 16, 0, 0);
        }
    };
    auto load_p = [&](int chunk, int buf) {
        unsigned char* sp = pbuf0 + buf * PBYTES;
        for (int j = 0; j < PI; ++j) {
            const int r = (wave * PI + j) * 8 + rsub;                // halo row (r & 7 == rsub)
            const long m = mfirst + r;
            const bf16_t* src = (r < NR && m >= 0 && m < g.M) ? g.P + m * C + chunk * 64 + csrc : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sp + (wave * PI + j) * 1024), 16, 0, 0);
        }
    };

    // per pixel-fragment validity of the nine taps (bit t set <=> tap t of this lane's pixel is inside the image)
    unsigned vmask[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * WM + b * 16 + (lane & 15);
        unsigned bits = 0;
        if (m < g.M) {
            const int h = m % H, w = (m / H) % g.cW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ww = w + t / 3 - 1, hh = h + t % 3 - 1;
                if ((unsigned)ww < (unsigned)g.cW && (unsigned)hh < (unsigned)H) bits |= 1u << t;
            }
        }
        vmask[b] = bits;
    }

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int nchunks = C / 64;
    const int nsteps = nchunks * 9;
    // Fragment addressing with as little per-step VALU work as possible (PMC: 2.4 VALU instructions per MFMA made the vector
    // ALU as busy as the matrix pipe, and the two do not overlap while all waves move in lock step):
    //   * SAME-padding: a lane whose pixel has no valid tap reads a 128-byte ROW OF ZEROS kept behind the tiles instead of
    //     selecting zeros into the 4 loaded registers — one v_cndmask on the address instead of four on the data;
    //   * the swizzle term ((kk*4 + fq) ^ (row & 7)) only depends on (lane, tap): row = wm*WM + b*16 + (lane&15) + shift,
    //     so it is formed once per step, the second K-half is `^ 64`, and the fragment index b is an immediate offset.
    // (the zero row is row NR of the current halo stage: rows NR .. NRpad-1 exist only as padding of the DMA geometry and
    //  are filled from the zero page with every halo tile; NRpad > NR always holds)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;          // LDS byte address of smem (0 for dynamic LDS, kept general)
    const unsigned qfrag0 = (unsigned)(qbuf0 - smem) + (wn * (16 * FN) + frow) * 128 + ((fq ^ fx) << 4);   // + stage*QB, ^64 for kk = 1
    const unsigned prow0 = (wm * WM + frow) * 128;                                                  // + shift*128 + swizzle
    // prologue: weights of step 0, then halo of chunk 0
    load_q(0, 0);
    load_p(0, 0);
    int tap = 0, chunk = 0;
    for (int s = 0; s < nsteps; ++s) {
        // Q(s) was issued one step ago; at tap 1 the halo of the NEXT chunk was issued right after it and may stay in flight
        if (tap == 1 && chunk + 1 < nchunks) {
            if (PI == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (PI == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (PI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // issue next step's weights, and (at tap 0) the next chunk's halo AFTER them
        {
            int ntap = tap + 1, nchunk = chunk;
            if (ntap == 9) { ntap = 0; ++nchunk; }
            if (ABL != 2 && ABL != 5) {
                if (s + 1 < nsteps) load_q(ntap * C + nchunk * 64, (s + 1) & 1);
                if (tap == 0 && chunk + 1 < nchunks) load_p(chunk + 1, (chunk + 1) & 1);
            }
        }
        const int shift = (H + 1) + (tap / 3 - 1) * H + (tap % 3 - 1);       // halo row of local pixel 0 for this tap
        const unsigned qa = qfrag0 + (s & 1) * QB;
        // the zero rows (NR and NR + 1, both spare: NR is even, NRpad a multiple of 32 above it) are read at the SAME position of
        // their 256 bytes as the real row would be: a redirected lane then uses the 4 of 64 banks the swizzle gave it, instead of
        // piling onto banks 0-3 (PMC r02: 0.15-0.20 conflict cycles per LDS-active
        // cycle at H = 4 / 8, where 1/4 resp. 1/8 of the lanes are redirected for six of the nine taps)
        const unsigned psw = (fq ^ ((frow + shift) & 7)) << 4;
        const unsigned pbase = (chunk & 1) * PBYTES + prow0 + shift * 128 + psw;
        unsigned pa[FM];
        const unsigned zoff = (chunk & 1) * PBYTES + NR * 128 + (((frow + shift) & 1) << 7) + psw;    // rows NR, NR + 1: address bits 0-7 (the bank) unchanged
#pragma unroll
        for (int b = 0; b < FM; ++b) pa[b] = ((vmask[b] >> tap) & 1u) ? pbase + b * 2048 : zoff;
        // All 16 fragment reads of the step are issued back to back as inline asm (the compiler does not count them, so it adds
        // no waits of its own), then two hand-placed counted waits: lgkmcnt(FN + FM) before the first K-half's MFMAs — its
        // fragments were issued first and LDS returns in order — and lgkmcnt(0) before the second.  hipcc's own schedule
        // interleaved small read groups with the MFMAs and drained lgkmcnt(0) five times per step.
        u32x4 afr[2][FN], bfr[2][FM];
        if (ABL == 1 || ABL == 5) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int a = 0; a < FN; ++a) asm volatile("" : "=v"(afr[kk][a]));
#pragma unroll
                for (int b = 0; b < FM; ++b) asm volatile("" : "=v"(bfr[kk][b]));
            }
        } else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned qk = qa ^ (kk * 64);
#pragma unroll
            for (int a = 0; a < FN; ++a)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[kk][a]) : "v"(lds0 + qk), "n"(a * 2048));
#pragma unroll
            for (int b = 0; b < FM; ++b)
                asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[kk][b]) : "v"(lds0 + (pa[b] ^ (kk * 64))));
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FN + FM) : "memory");     // the second half's reads may stay in flight
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (g.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[kk][a]),
                                                                        __builtin_bit_cast(bf16x8, bfr[kk][b]), acc[a][b], 0, 0, 0);
            if (g.prio && kk == 1) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (++tap == 9) { tap = 0; ++chunk; }
    }

    // ---- epilogue (m-block outer, n-fragment inner: the stores completing a 128-B run of a pixel row are adjacent)
    const int flags = g.flags;
    f32x4 bv[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int n = n0 + wn * (16 * FN) + a * 16 + (lane >> 4) * 4;
        bv[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((flags & IGH_BIAS) && n < g.N) bv[a] = *(const f32x4*)(g.bias + n);
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * WM + b * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int n = n0 + wn * (16 * FN) + a * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            f32x4 v = acc[a][b] + bv[a];
            if (flags & IGH_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (flags & IGH_MASK) {
                u32x2 mk = *(const u32x2*)(g.mask + (long)m * g.N + n);
                if (!(bf_lo(mk.x) > 0.f)) v.x = 0.f;
                if (!(bf_hi(mk.x) > 0.f)) v.y = 0.f;
                if (!(bf_lo(mk.y) > 0.f)) v.z = 0.f;
                if (!(bf_hi(mk.y) > 0.f)) v.w = 0.f;
            }
            if (flags & IGH_ACCUM) {
                const u32x2 old = *(const u32x2*)(g.out + (long)m * g.N + n);
                v.x += bf_lo(old.x); v.y += bf_hi(old.x); v.z += bf_lo(old.y); v.w += bf_hi(old.y);
            }
            u32x2 pk;
            pk.x = pack_bf2(v.x, v.y);
            pk.y = pack_bf2(v.z, v.w);
            *(u32x2*)(g.out + (long)m * g.N + n) = pk;
            if (g.pool_kind) {
                // max-pool window of this pixel (max_pool after the ReLU: LSTM_train.py:27-33).  Rounding to bf16 is monotone, so
                // the max of the fp32 values, rounded, is the max of the stored bf16 values.  Feature-axis neighbour = pixel m ^ 1 =
                // lane ^ 1; time-axis neighbour = pixel m +- H: lane ^ H for H = 4, 8, the same lane of fragment b ^ 1 for H = 16.
                // (M is a multiple of 2 H, tiles start at multiples of 32 pixels: a window never straddles tiles or the m < M bound.)
                f32x4 mx = v;
                if (g.pool_kind == 2) {
                    if (H == 16) {
                        f32x4 u = acc[a][FM > 1 ? (b ^ 1) : 0] + bv[a];      // (instances with FM == 1 are not dispatched for this case)
                        u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f);
                        mx.x = fmaxf(mx.x, u.x); mx.y = fmaxf(mx.y, u.y); mx.z = fmaxf(mx.z, u.z); mx.w = fmaxf(mx.w, u.w);
                    } else {
                        mx.x = fmaxf(mx.x, __shfl_xor(mx.x, H, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, H, 64));
                        mx.z = fmaxf(mx.z, __shfl_xor(mx.z, H, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, H, 64));
                    }
                }
                mx.x = fmaxf(mx.x, __shfl_xor(mx.x, 1, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, 1, 64));
                mx.z = fmaxf(mx.z, __shfl_xor(mx.z, 1, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, 1, 64));
                const int h = m % H, col = m / H;
                const bool writer = !(h & 1) && (g.pool_kind == 1 || !(col & 1));
                if (writer) {
                    const long pidx = g.pool_kind == 1 ? (long)(m >> 1) : (long)(col >> 1) * (H >> 1) + (h >> 1);
                    u32x2 pp;
                    pp.x = pack_bf2(mx.x, mx.y);
                    pp.y = pack_bf2(mx.z, mx.w);
                    *(u32x2*)(g.pool + pidx * g.N + n) = pp;
                }
            }
        }
    }
    if (clk && blockIdx.x == 0 && tid == 0) ocr_clk_exit(clk);
}

// (A variant on v_mfma_f32_32x32x16_bf16 — same tiles, 32-row fragments, row swizzle (r & 7) ^ ((r >> 3) & 1) — was written in round 2,
// passed the parity tests on hardware in round 3 and measured 10-15 % SLOWER on every layer (profiles/r03a_conv_32x32x16.log): removed.)

template <int BN, int NW, int ABL = 0, int RPW = 32, int FNV = 4>
static int launch_halo_(const HaloArgs& g, hipStream_t stream) {
    if (!g.P) return 1;                                  // plan query (ocr_conv3x3_kernel_choice): this file's kernels, nothing is launched
    constexpr int BM = RPW * NW;
    const int NR = BM + 2 * g.cH + 2;
    const int NRpad = (NR + 8 * NW) / (8 * NW) * (8 * NW);       // strictly greater than NR: the spare rows are the zero rows
    const int lds = 2 * NRpad * 128 + 2 * BN * 128;              // halo stages, weight stages
    static int lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute((const void*)conv_halo_kernel<BN, NW, ABL, RPW, FNV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return OCR_ERR_EXEC;
        lds_set = lds;
    }
    int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    conv_halo_kernel<BN, NW, ABL, RPW, FNV><<<mt * nt, 64 * NW, lds, stream>>>(g, NRpad);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
template <int BN, int NW>
static int launch_halo(const HaloArgs& g, hipStream_t stream) {
#ifdef OCR_EXPERIMENTS
    static int abl = -1;                        // timing ablations for tools/halo_variants.py (OCR_HALO_ABL; results are wrong)
    if (abl < 0) { const char* e = getenv("OCR_HALO_ABL"); abl = e ? atoi(e) : 0; }
    if (abl == 1) return launch_halo_<BN, NW, 1>(g, stream);
    if (abl == 2) return launch_halo_<BN, NW, 2>(g, stream);
    if (abl == 5) return launch_halo_<BN, NW, 5>(g, stream);
#endif
    return launch_halo_<BN, NW, 0>(g, stream);
}

int k2_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind);
int ws_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind);
#define K2_DEFAULT 1            // on since the weight prefetch distance went from 2 to 3 steps (profiles/r03n: the step 1.415 against 1.440 ms, three
                                // interleaved runs each; with distance 2 it was an opt-in: no faster (tile D) or 1.4 % slower (tiles A + D) in the step,
                                // profiles/r03l_bench_conv_ab.log, although 8 % faster back to back, profiles/r03j_conv_k2_final.log)

// -1 = shape not covered (caller falls back to igemm.hip / gemm.hip)
int halo_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                      const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind) {
    if ((Cin & 63) || (Cout & 3) || M < 1024 || H > 30) return -1;
    if (flags & ~(IGH_BIAS | IGH_RELU | IGH_MASK | IGH_ACCUM)) return -1;
    // seventh generation (conv_ws.hip: weights in registers, workgroups persistent over the pixel tiles) for the short-K layers it covers and
    // chooses (Cin = 64 / 128 at H = 16 / 8 with at least two tiles per workgroup); A/B knob OCR_CONV_WS = 0 / 2 (never / every covered shape)
    if (pool_kind < 3) {
        const int rc = ws_try_dispatch(x, wpack, y, M, W, H, Cin, Cout, bias, mask, flags, stream, pool, pool_kind);
        if (rc >= 0) return rc;
    }
    if (pool_kind >= 3) {    // batch-norm statistics from the epilogue (3: forward, `pool` = float partial rows; 4: backward sums of a masked data
                             // gradient, `pool` = host K3BnBwd): the plane-layout kernels only
        if (pool_kind > 4 || !pool || (flags & IGH_ACCUM) || (M & 255)) return -1;
        if (pool_kind == 3 ? (flags & IGH_MASK) != 0 : !(flags & IGH_MASK)) return -1;
        static int k2s = -1;
        if (k2s < 0) { const char* e = getenv("OCR_CONV_K2"); k2s = e ? atoi(e) : K2_DEFAULT; }
        return k2s ? k2_try_dispatch(x, wpack, y, M, W, H, Cin, Cout, bias, mask, flags, stream, pool, pool_kind) : -1;
    }
    if (pool_kind) {         // fused max-pool: ReLU epilogue without mask, even feature axis, 2 x 2 only for H in {4, 8, 16} and even W
        if (!pool || (flags & (IGH_MASK | IGH_ACCUM)) || !(flags & IGH_RELU) || (H & 1) || (Cout & 3) || (Cout % 64 && Cout % 128)) return -1;
        if (pool_kind == 2 && ((H != 4 && H != 8 && H != 16) || (W & 1))) return -1;
        if (pool_kind != 1 && pool_kind != 2) return -1;
        if (M % (pool_kind == 2 ? 2 * H : 2)) return -1;
    }
    // fifth generation (conv_k2.hip: 128 x 64 wave tiles through an in-workgroup K split, ping-pong K halves) where its tiles fill the
    // chip; A/B knob OCR_CONV_K2 = 0 keeps every shape on this file's kernels
    static int k2 = -1;
    if (k2 < 0) { const char* e = getenv("OCR_CONV_K2"); k2 = e ? atoi(e) : K2_DEFAULT; }
    if (k2) {
        const int rc = k2_try_dispatch(x, wpack, y, M, W, H, Cin, Cout, bias, mask, flags, stream, pool, pool_kind);
        if (rc >= 0) return rc;
    }
    // Tile choice.  128-pixel workgroups of 4 waves leave room for TWO workgroups per CU (80 KiB of LDS each): they drift out
    // of phase, so one's MFMA burst covers the other's barrier / DMA-issue / LDS-read phase (+5-7 % where the grid then still
    // has two workgroups for every CU).  Otherwise 256-pixel workgroups of 8 waves, one per CU.
    static int nw = -1;                                  // A/B knob OCR_HALO_NW: 8 / 4 force one kind, unset = by grid size
    if (nw < 0) { const char* e = ocr_tune_env("OCR_HALO_NW"); nw = e ? atoi(e) : 0; }
    static int stag = -1, stag_bit = 32;                 // experiment knob OCR_HALO_STAGGER=units[,bit] (units of 64 clocks; default off)
    if (stag < 0) { const char* e = ocr_tune_env("OCR_HALO_STAGGER"); stag = e ? atoi(e) : 0; const char* c = e ? strchr(e, ',') : nullptr; if (c) stag_bit = atoi(c + 1); if (stag < 0) stag = 0; }
    static int prio = -1;
    if (prio < 0) { const char* e = ocr_tune_env("OCR_HALO_PRIO"); prio = e ? atoi(e) : 1; }      // measured: 451 against 458 us over the ten launches
    HaloArgs g = {(const bf16_t*)x, (const bf16_t*)wpack, M, Cout, Cin, W, H, (bf16_t*)y, bias, (const bf16_t*)mask, flags, (bf16_t*)pool, pool_kind, stag, stag_bit, prio};
#ifdef OCR_EXPERIMENTS      // measured and rejected in round 2: dense = equal, wide = 27 % slower
    static int dense = -1;                               // A/B knob OCR_HALO_DENSE=1: 8 waves per 128-pixel tile (4 waves per SIMD with two workgroups per CU)
    if (dense < 0) { const char* e = getenv("OCR_HALO_DENSE"); dense = e ? atoi(e) : 0; }
    if (dense && nw != 8) {
        const int NRd = (128 + 2 * H + 2 + 64) / 64 * 64;
        const long mt4 = (M + 127) / 128;
        if (NRd / 64 == 3) {
            if (Cout >= 128 && mt4 * ((Cout + 127) / 128) >= 448) return launch_halo_<128, 8, 0, 16>(g, stream);
            if (!(pool_kind == 2 && H == 16)) return launch_halo_<64, 8, 0, 16>(g, stream);     // one fragment per wave: no partner fragment
        }
    }
    static int wide = -1;                                // A/B knob OCR_HALO_WIDE=1: 64 x 128 wave tiles (128 pixels x 256 channels per 4-wave workgroup, one per CU)
    if (wide < 0) { const char* e = getenv("OCR_HALO_WIDE"); wide = e ? atoi(e) : 0; }
    if (wide && nw != 8 && Cout % 256 == 0) {
        const int NRw = (128 + 2 * H + 2 + 32) / 32 * 32;
        if ((NRw / 32 == 5 || NRw / 32 == 6) && (long)((M + 127) / 128) * (Cout / 256) >= 224)
            return launch_halo_<256, 4, 0, 32, 8>(g, stream);
    }
#endif
    if (nw != 8) {
        const int NR = 128 + 2 * H + 2, NRp = (NR + 32) / 32 * 32;
        const long mt4 = (M + 127) / 128;
        if (NRp / 32 == 5 || NRp / 32 == 6) {                                // counted vmcnt literals exist for 5 and 6
            if (Cout >= 128 && mt4 * ((Cout + 127) / 128) >= 448) return launch_halo<128, 4>(g, stream);
            // otherwise the smallest tile: it has the most workgroups, and every alternative below leaves CUs idle (a 256-channel
            // layer at M = 8192 is 256 workgroups here against 64 of the 8-wave 256 x 128 tile)
            if (nw != 8) return launch_halo<64, 4>(g, stream);
        } else if (nw == 4) return -1;
    }
    const int NRpad = (256 + 2 * H + 2 + 64) / 64 * 64;
    if (NRpad / 64 != 5 && NRpad / 64 != 6) return -1;               // counted vmcnt literals exist for 5 and 6
    const int mt = (M + 255) / 256;
    if (Cout >= 128 && (long)mt * ((Cout + 127) / 128) >= 200) return launch_halo<128, 8>(g, stream);
    if (Cout >= 128 && (long)mt * ((Cout + 63) / 64) < 256) return launch_halo<128, 8>(g, stream);
    return launch_halo<64, 8>(g, stream);
}

int k3_set_clock_debug(void* dbg);       // conv_k3.hip: the plane-layout kernels stamp the same block
int ws_set_clock_debug(void* dbg);       // conv_ws.hip likewise
extern "C" int ocr_conv_halo_clock_debug(void* dbg /* device int64[8] or NULL */) {
    long long* p = (long long*)dbg;
    if (k3_set_clock_debug(dbg) != OCR_OK || ws_set_clock_debug(dbg) != OCR_OK) return OCR_ERR_EXEC;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_halo_clk), &p, sizeof(p)) == hipSuccess ? OCR_OK : OCR_ERR_EXEC;
}

// does the halo kernel take this shape (so that its epilogue-only features — accumulate, fused pool — may be asked for)?
bool halo_covers(long M, int W, int H, int Cin, int Cout) {
    if ((Cin & 63) || (Cout & 3) || M < 1024 || M > 0x7fffffffL || H > 30 || H < 1) return false;
    const int NR4 = (128 + 2 * H + 2 + 32) / 32 * 32, NR8 = (256 + 2 * H + 2 + 64) / 64 * 64;
    return NR4 / 32 == 5 || NR4 / 32 == 6 || NR8 / 64 == 5 || NR8 / 64 == 6;
}
