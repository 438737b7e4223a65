// 3x3 SAME implicit-GEMM convolution for gfx950, fifth generation: 128 x 64 WAVE TILES through an IN-WORKGROUP K SPLIT,
// ping-pong between the two K halves (same contract as ocr_conv3x3_bf16 — reference lib/networks/network.py:160-191 forward, and
// its data gradient; same halo-tile / tap-reuse idea as conv_halo.hip).
//
// Why (round 3): conv_halo's 64 x 64 wave tiles read 0.5 LDS fragments per MFMA and its 128-pixel
// workgroups re-stream a 16 KiB weight tile per 2.1 MFLOP; with two such workgroups per CU the LDS carries 128 ds_read_b128 +
// 36 KiB of DMA per K step against 1024 MFMA cycles per SIMD — the LDS is as busy as the matrix pipe, which is why removing
// either the reads or the DMA returned the same ~20 % and nothing overlapped (r02 ablations).  Larger wave tiles need more
// accumulators than two independent workgroups per CU can hold, and larger workgroup tiles leave half the chip idle at batch 64
// (128 tiles of 256 x 256).  So: ONE workgroup of 8 waves per CU on a 256-pixel x 128-channel tile, waves = 2 (pixels) x 2
// (channels) x 2 (K halves): the waves w and w + 4 own the SAME 128 x 64 output block, w the channels 0-31 of every 64-channel
// K chunk and w + 4 the channels 32-63.  Per K step (tap, chunk) a wave reads 4 weight + 8 pixel fragments for 32 MFMAs (0.375
// per MFMA), the workgroup DMAs 16 KiB of weights + 1/9 of a 40 KiB halo tile for 4.2 MFLOP (half of conv_halo's bytes per flop),
// and 256 tiles still fill the chip for every layer of the headline step but two (those take the 128-pixel form, FM = 4).
// The two K halves meet once, after the last step: each wave hands half of its accumulators to its partner through LDS
// (16 KiB per wave, the stages are dead by then) and finishes the other half (bias / ReLU / mask / accumulate / fused max-pool
// epilogue as in conv_halo).
//
// Schedule (fourth version; what was measured on the way is in DESIGN section 3): the two waves of a SIMD (w, w + 4) run half a step
// apart — waves 0-3 do LOAD(s) = 12 fragment reads + DMA issue for step s + 2, then COMP(s) = 32 MFMAs (s_setprio 1) with the NEXT
// step's fragment addresses computed beside them; waves 4-7 do COMP(s - 1) on the fragments they read in the previous interval, then
// LOAD(s) — with ONE workgroup barrier per step: see the comment at `interval` in the kernel.  Weight tiles have four stages (three
// for the 512-pixel tiles), halo tiles two; DMA pieces are retired with counted vmcnt one interval after their issue.
// DMA addressing: buffer_load ... lds through a buffer descriptor — per piece a loop-invariant 32-bit lane offset and a scalar
// offset that advances per step, no per-step vector arithmetic (the flat-pointer form cost 150-260 issue cycles per piece in
// conv_halo); rows that do not exist (n >= N, pixels outside [0, M), the zero rows behind the halo) carry an out-of-range lane
// offset and arrive as zeros.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

enum { K2_BIAS = 1, K2_RELU = 2, K2_MASK = 16, K2_ACCUM = 64 };

struct K2Args {
    const bf16_t* P; const bf16_t* Q;     // P [M pixels][C] ; Q [N][9*C]
    int M, N, C;                          // C % 64 == 0
    int cW, cH;
    bf16_t* out; const float* bias; const bf16_t* mask; int flags;
    bf16_t* pool; int pool_kind;          // as conv_halo: 0 none, 1 feature pairs (1 x 2), 2 2 x 2
    int abl;                              // timing ablations (wrong results; only in `make EXPERIMENTS=1` builds, OCR_K2_ABL): 1 no DMA after the
};                                        // prologue, 2 no fragment reads, 4 no MFMAs

#ifdef OCR_EXPERIMENTS
// diagnostic (experiments build, OCR_K2_ABL & 8): s_memtime stamps of workgroup 0, kept in spare LDS behind the stages (no global store
// enters the vmcnt queue of the pipeline) and copied out at the end: dbg[(wave * 80 + interval) * 4 + {0 interval start, 1 MFMAs start,
// 2 MFMAs done, 3 in front of the barrier}]
__device__ unsigned* k2_dbg;
extern "C" int ocr_conv_k2_debug(void* dbg) {
    unsigned* q = (unsigned*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(k2_dbg), &q, sizeof(q)) == hipSuccess ? OCR_OK : OCR_ERR_MEMOPS;
}
#define K2_STAMP(slot) do { if ((abl & 8) && blockIdx.x == 0 && s < 80) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); \
        if (lane == 0) ((unsigned*)(smem + 2 * PBYTES + NST * QB))[(wave * 80 + s) * 4 + (slot)] = t_; } } while (0)
#else
#define K2_STAMP(slot) do { } while (0)
#endif
typedef __attribute__((address_space(3))) void* lptr_t;
#define K2_OOB 0x80000000u                // lane offset of a row that does not exist: beyond any descriptor's range

#define K2_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int FM /* 16-pixel fragments per wave: 8 or 4 */, int BN /* channels per tile: 128 (waves 2 pixel x 2 channel x 2 K) or 64
                                                                          (4 pixel x 1 channel x 2 K: 512- / 256-pixel tiles) */,
          int NST = 4 /* weight stages; tile s + NST - 2 is streamed during step s */>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_k2_kernel(K2Args g, int NRpad /* halo rows incl. the zero rows, a multiple of 8 */) {
    constexpr int NW = 8, FN = 4;
    constexpr int WN = BN / 64, WMW = 4 / WN;           // waves along channels / pixels (per K half)
    constexpr int WM = FM * 16, BM = WMW * WM;
    constexpr int PIMAX = ((BM + 38 + 7) / 8 + NW - 1) / NW;      // halo DMA pieces (8 rows) per wave and chunk, at most (9 / 5 / 3)
    constexpr int QI = BN / 64;                         // weight DMA pieces per wave and step (2 / 1)
    constexpr int QB = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PBYTES = NRpad * 128, QOFF = 2 * PBYTES;

#ifdef OCR_EXPERIMENTS
    const int abl = g.abl;
#else
    constexpr int abl = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wm = (wave & 3) / WN, wn = (wave & 3) % WN;
    const int H = g.cH, C = g.C;
    const int NR = BM + 2 * H + 2;                      // halo rows actually needed; rows NR, NR + 1 are the zero rows (NR + 2 <= NRpad)
    const int npieces = NRpad >> 3;

    const int mtiles = (g.M + BM - 1) / BM, ntiles = (g.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int Lb = blockIdx.x;
    if ((nblk & 7) == 0) Lb = (Lb & 7) * (nblk >> 3) + (Lb >> 3);        // XCD-aware: each XCD gets a contiguous run of tiles
    const int m0 = (Lb / ntiles) * BM, n0 = (Lb % ntiles) * BN;

    const int rsub = lane >> 3;
    const int csrc = ((lane & 7) ^ rsub) * 8;           // LDS position lane&7 of row r holds source chunk (lane&7)^(r&7)
    const long mfirst = (long)m0 - H - 1;               // flat pixel of halo row 0

    const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc((void*)g.P, 0, (int)((long)g.M * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t qsrd = __builtin_amdgcn_make_buffer_rsrc((void*)g.Q, 0, (int)((long)g.N * 9 * C * 2), 0x00020000);
    unsigned voffQ[QI], voffP[PIMAX];
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int n = n0 + (wave * QI + j) * 8 + rsub;
        voffQ[j] = n < g.N ? (unsigned)(((long)n * 9 * C + csrc) * 2) : K2_OOB;
    }
#pragma unroll
    for (int j = 0; j < PIMAX; ++j) {                   // this wave's halo pieces are j * 8 + wave (those below npieces)
        const int r = (j * NW + wave) * 8 + rsub;
        const long m = mfirst + r;
        voffP[j] = (r < NR && m >= 0 && m < g.M) ? (unsigned)((m * C + csrc) * 2) : K2_OOB;
    }
    auto load_q = [&](int k0 /* tap*C + chunk*64 */, int stage) {
#pragma unroll
        for (int j = 0; j < QI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qsrd, (lptr_t)(smem + QOFF + stage * QB + (wave * QI + j) * 1024), 16, (int)voffQ[j], k0 * 2, 0, 0);
    };
    auto load_p1 = [&](int chunk, int buf, int j, unsigned voff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(psrd, (lptr_t)(smem + buf * PBYTES + (j * NW + wave) * 1024), 16, (int)voff, chunk * 128, 0, 0);
    };
    // piece j of the next chunk (static register index; pieces beyond npieces do not exist for this wave)
    auto load_pj = [&](int chunk, int buf, int j) {
        if ((j * NW + wave) >= npieces) return 0;
        switch (j) {
            case 0: load_p1(chunk, buf, 0, voffP[0]); break;
            case 1: if (PIMAX > 1) load_p1(chunk, buf, 1, voffP[PIMAX > 1 ? 1 : 0]); break;
            case 2: if (PIMAX > 2) load_p1(chunk, buf, 2, voffP[PIMAX > 2 ? 2 : 0]); break;
            case 3: if (PIMAX > 3) load_p1(chunk, buf, 3, voffP[PIMAX > 3 ? 3 : 0]); break;
            case 4: if (PIMAX > 4) load_p1(chunk, buf, 4, voffP[PIMAX > 4 ? 4 : 0]); break;
            case 5: if (PIMAX > 5) load_p1(chunk, buf, 5, voffP[PIMAX > 5 ? 5 : 0]); break;
            case 6: if (PIMAX > 6) load_p1(chunk, buf, 6, voffP[PIMAX > 6 ? 6 : 0]); break;
            case 7: if (PIMAX > 7) load_p1(chunk, buf, 7, voffP[PIMAX > 7 ? 7 : 0]); break;
            default: if (PIMAX > 8) load_p1(chunk, buf, 8, voffP[PIMAX > 8 ? 8 : 0]); break;
        }
        return 1;
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
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    // this wave's K half: 16-byte chunks kh*4 + fq of a 128-byte row, at position (kh*4 + fq) ^ (row & 7)
    const unsigned qfrag0 = lds0 + QOFF + (wn * 64 + frow) * 128 + ((((kh << 2) | fq) ^ fx) << 4);       // + stage*QB, + a*2048
    const unsigned prow0 = lds0 + (wm * WM + frow) * 128;                                                 // + stage, + shift*128 + swizzle

    // ---- prologue: weight tiles of steps 0 and 1, the whole halo of chunk 0; everything has landed before barrier 0
    constexpr int DEPTH = NST - 2;                      // prefetch distance of the weight tiles, in steps
    load_q(0, 0);
    if (nsteps > 1) load_q(C, 1);
    if (DEPTH > 2 && nsteps > 2) load_q(2 * C, 2);
    if (DEPTH > 3 && nsteps > 3) load_q(3 * C, 3);
#pragma unroll
    for (int j = 0; j < PIMAX; ++j) load_pj(0, 0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // The nine taps are unrolled, so everything that depends on the tap alone is an immediate, a scalar or a loop-invariant lane mask.
    unsigned pa[FM];                                    // pixel-fragment addresses of the step being loaded
    u32x4 afr[FN], bfr[FM];                             // fragments: waves 4-7 carry them across the barrier
    auto gen_addr = [&](auto tapc, int chunk) {
        constexpr int TAP = decltype(tapc)::value;
        // halo row of local pixel 0 for this tap; pixel fragment b: row (wm*WM + b*16 + frow + shift), chunk position
        // ((kh*4 + fq) ^ (row & 7)); lanes whose tap falls outside the image read a zero row (NR / NR + 1: the one with the real row's
        // parity, at the real row's position in its 256 bytes, so the redirected lanes keep the banks the swizzle gave them)
        const int shift = (H + 1) + (TAP / 3 - 1) * H + (TAP % 3 - 1);
        const unsigned psw = ((((unsigned)kh << 2) | (unsigned)fq) ^ (unsigned)((frow + shift) & 7)) << 4;
        const unsigned pbase = prow0 + (chunk & 1) * PBYTES + shift * 128 + psw;
        const unsigned zoff = lds0 + (chunk & 1) * PBYTES + NR * 128 + (((frow + shift) & 1) << 7) + psw;
#pragma unroll
        for (int b = 0; b < FM; ++b) pa[b] = (vmask[b] & (1u << TAP)) ? pbase + b * 2048 : zoff;
    };
    int qs = 0 /* s % NST */, s = 0;
    // LOAD(s): the 12 fragment reads of step s, then the DMA issue; returns the number of pieces issued
    auto load = [&](auto tapc, int chunk) -> int {
        constexpr int TAP = decltype(tapc)::value;
        gen_addr(tapc, chunk);
        const unsigned qa = qfrag0 + qs * QB;
        if (abl & 2) {
#pragma unroll
            for (int a = 0; a < FN; ++a) asm volatile("" : "=v"(afr[a]));
#pragma unroll
            for (int b = 0; b < FM; ++b) asm volatile("" : "=v"(bfr[b]) : "v"(pa[b]));
        } else {
#pragma unroll
            for (int a = 0; a < FN; ++a)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[a]) : "v"(qa), "n"(a * 2048));
#pragma unroll
            for (int b = 0; b < FM; ++b)
                asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[b]) : "v"(pa[b]));
        }
        // DMA: the weight tile of step s + 2, and this step's share of the next chunk's halo pieces
        const bool more_q = s + DEPTH < nsteps && !(abl & 1);
        int issued = more_q ? QI : 0;
        constexpr int T2 = (TAP + DEPTH) % 9;
        const int c2 = chunk + (TAP + DEPTH >= 9 ? 1 : 0);
        int q2 = qs + DEPTH; if (q2 >= NST) q2 -= NST;
        if (more_q) load_q(T2 * C + c2 * 64, q2);
        // PPT pieces per step at taps 1 .. 5: the last piece is issued four steps before tap 0 of the next chunk reads the tile (a piece
        // is only known to have landed one step after its issue — the counted wait covers what was issued BEFORE the latest LOAD)
        constexpr int PPT = (PIMAX + 4) / 5;
        if (TAP >= 1 && TAP <= 5 && chunk + 1 < nchunks && !(abl & 1)) {
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
                constexpr int J0 = TAP >= 1 ? (TAP - 1) * PPT : 0;
                if (J0 + u < PIMAX) issued += load_pj(chunk + 1, (chunk + 1) & 1, J0 + u);
            }
        }
        return issued;
    };
    // COMP: the MFMAs on the fragments in registers — nothing else: address arithmetic interleaved with them (third version) doubled the
    // time of the MFMA burst (stamps: 32 MFMAs 950 instead of 516 clocks; the in-order wave waits on every s_nop / dependency in between)
    auto comp = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        K2_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        if (kh) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);      // the K half that multiplies first drains first (conv_k3.hip, profiles/r03af)
        if (!(abl & 4)) {
#pragma unroll
            for (int b = 0; b < FM; ++b)
#pragma unroll
                for (int a = 0; a < FN; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[a]), __builtin_bit_cast(bf16x8, bfr[b]),
                                                                        acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < FN; ++a) asm volatile("" :: "v"(afr[a]));
#pragma unroll
            for (int b = 0; b < FM; ++b) asm volatile("" :: "v"(bfr[b]));
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        K2_STAMP(2);
    };
    auto vmwait = [&](int younger) {                    // retire everything but the `younger` most recent pieces of this wave
        switch (younger) {
            case 0: K2_VMWAIT(0); break; case 1: K2_VMWAIT(1); break; case 2: K2_VMWAIT(2); break; case 3: K2_VMWAIT(3); break;
            case 4: K2_VMWAIT(4); break; case 5: K2_VMWAIT(5); break; case 6: K2_VMWAIT(6); break; case 7: K2_VMWAIT(7); break;
            case 8: K2_VMWAIT(8); break; case 9: K2_VMWAIT(9); break; case 10: K2_VMWAIT(10); break; case 11: K2_VMWAIT(11); break;
            default: K2_VMWAIT(12); break;
        }
    };
    int prev_issued = 0, prev2_issued = 0;              // DEPTH 3 (4): the pieces of the previous (two) interval(s) may stay in flight too
    // One interval = one K step, closed by ONE workgroup barrier.  Waves 0-3 run LOAD(s) then COMP(s); waves 4-7 run COMP(s - 1) (on
    // the fragments they loaded in the previous interval) then LOAD(s): the two waves of a SIMD use the matrix pipe in opposite halves
    // of the interval, and nothing forces a load segment and an MFMA segment to take equally long (the first versions had a second
    // barrier in the middle: the interval then cost twice the LONGER segment, and the load segment is the longer one).
    //   * tile s + 2 is streamed into the stage of tile s + 2 - NST: with four stages its last readers finished an interval ago; with
    //     three (512-pixel tiles) waves 4-7 read it at the END of the previous interval, so they wait for those reads before the barrier;
    //   * every wave retires, before the barrier of interval s, what it issued before this interval (counted vmcnt) — with five stages
    //     (tile A: prefetch distance 3) before the PREVIOUS interval: a piece then has two intervals (~1.8 us) to arrive, which is what
    //     a first-touch miss to HBM needs under load (with one interval tile A lost 15-25 % behind a cache scrub, profiles/r03m);
    //     the barrier publishes tile s + 1 and any halo piece one (two) intervals after its issue.
    auto interval = [&](auto khc, auto tapc, int chunk) {
        constexpr int KH = decltype(khc)::value, TAP = decltype(tapc)::value;
        int issued;
        K2_STAMP(0);
        if (KH == 0) {
            issued = load(tapc, chunk);
            comp();
        } else {
            if (s > 0) comp();                           // step s - 1
            issued = load(tapc, chunk);
            if (NST < 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        vmwait(DEPTH > 3 ? issued + prev_issued + prev2_issued : DEPTH > 2 ? issued + prev_issued : issued);
        prev2_issued = prev_issued;
        prev_issued = issued;
        if (KH == 0) K2_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ++s;
        if (++qs == NST) qs = 0;
    };
    // the two wave groups run separate copies of the loop (a per-interval branch on kh made the register allocator keep both groups'
    // live ranges: a thousand spills); both copies execute the same number of barriers
    auto run = [&](auto khc) {
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            interval(khc, std::integral_constant<int, 0>{}, chunk); interval(khc, std::integral_constant<int, 1>{}, chunk);
            interval(khc, std::integral_constant<int, 2>{}, chunk); interval(khc, std::integral_constant<int, 3>{}, chunk);
            interval(khc, std::integral_constant<int, 4>{}, chunk); interval(khc, std::integral_constant<int, 5>{}, chunk);
            interval(khc, std::integral_constant<int, 6>{}, chunk); interval(khc, std::integral_constant<int, 7>{}, chunk);
            interval(khc, std::integral_constant<int, 8>{}, chunk);
        }
        if (decltype(khc)::value == 1) comp();          // the last step's MFMAs
    };
    if (kh == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
#ifdef OCR_EXPERIMENTS
    if ((abl & 8) && blockIdx.x == 0 && k2_dbg != nullptr) {
        __syncthreads();
        for (int i = tid; i < 8 * 80 * 4; i += 512) k2_dbg[i] = ((const unsigned*)(smem + 2 * PBYTES + NST * QB))[i];
        __syncthreads();
    }
#endif

    // ---- the two K halves meet: a wave keeps the pixel fragments [kh*FM/2, (kh+1)*FM/2) and hands the others to its partner
    // (every MFMA and every fragment read of the workgroup is complete: all waves have passed the last barrier).
    // kh is wave-uniform but not a compile-time constant; the tail is instantiated for both halves and branched on, so that every
    // accumulator index stays static (a select between two elements of acc[][] sent the whole array to scratch memory).
    constexpr int FH = FM / 2;
    auto tail = [&](auto half_c) {
        constexpr int KH = decltype(half_c)::value;
        {
            f32x4* mine = (f32x4*)smem + (size_t)wave * (FN * FH * 64);
            const f32x4* theirs = (const f32x4*)smem + (size_t)(wave ^ 4) * (FN * FH * 64);
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FH; ++b) mine[(a * FH + b) * 64 + lane] = acc[a][(1 - KH) * FH + b];      // what the partner finishes
            __syncthreads();
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FH; ++b) acc[a][KH * FH + b] += theirs[(a * FH + b) * 64 + lane];
        }
        // epilogue for this wave's FH pixel fragments (m-block outer, n-fragment inner: the stores completing a 128-B run of a pixel
        // row are adjacent)
        const int flags = g.flags;
        f32x4 bv[FN];
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
            bv[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((flags & K2_BIAS) && n < g.N) bv[a] = *(const f32x4*)(g.bias + n);
        }
#pragma unroll
        for (int bb = 0; bb < FH; ++bb) {
            const int m = m0 + wm * WM + (KH * FH + bb) * 16 + (lane & 15);
            if (m >= g.M) continue;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
                if (n >= g.N) continue;
                f32x4 v = acc[a][KH * FH + bb] + bv[a];
                if (flags & K2_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (flags & K2_MASK) {
                    u32x2 mk = *(const u32x2*)(g.mask + (long)m * g.N + n);
                    if (!(bf_lo(mk.x) > 0.f)) v.x = 0.f;
                    if (!(bf_hi(mk.x) > 0.f)) v.y = 0.f;
                    if (!(bf_lo(mk.y) > 0.f)) v.z = 0.f;
                    if (!(bf_hi(mk.y) > 0.f)) v.w = 0.f;
                }
                if (flags & K2_ACCUM) {
                    const u32x2 old = *(const u32x2*)(g.out + (long)m * g.N + n);
                    v.x += bf_lo(old.x); v.y += bf_hi(old.x); v.z += bf_lo(old.y); v.w += bf_hi(old.y);
                }
                u32x2 pk;
                pk.x = pack_bf2(v.x, v.y);
                pk.y = pack_bf2(v.z, v.w);
                *(u32x2*)(g.out + (long)m * g.N + n) = pk;
                if (g.pool_kind) {
                    // max-pool window of this pixel, as in conv_halo: feature-axis neighbour = lane ^ 1; time-axis neighbour = lane ^ H
                    // for H = 4, 8, the same lane of fragment b ^ 1 for H = 16 (b ^ 1 stays inside this wave's half: FH is even)
                    f32x4 mx = v;
                    if (g.pool_kind == 2) {
                        if (H == 16) {
                            f32x4 u = acc[a][KH * FH + (bb ^ 1)] + bv[a];
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
    };
    if (kh) tail(std::integral_constant<int, 1>{}); else tail(std::integral_constant<int, 0>{});
}

// tile configurations: A = 256 pixels x 128 channels (waves 2 x 2 x 2 K), D = 256 x 64 (4 x 1 x 2 K: layers whose 256 x 128 tiles would not
// fill the chip); 512 x 64 and 128 x 128 were built and measured too (profiles/r03e / r03h_conv*.log): never faster than these two
template <int FM, int BN, int NST = 4>
static int launch_k2(const K2Args& g, hipStream_t stream) {
    if (!g.P) return BN == 128 ? 2 : 3;                  // plan query (ocr_conv3x3_kernel_choice): nothing is launched
    constexpr int BM = (4 / (BN / 64)) * FM * 16;
    const int NRpad = (BM + 2 * g.cH + 4 + 7) / 8 * 8;             // needed rows + two zero rows, in 8-row DMA pieces
    int lds = 2 * NRpad * 128 + NST * BN * 128;                     // halo stages, weight stages (the K-half exchange reuses them)
    if (lds > 163840 || lds < 8 * FM * 2048) return -1;
#ifdef OCR_EXPERIMENTS
    if (lds + 10240 <= 163840) lds += 10240; else if (g.abl & 8) return -1;      // room for the stamps
#endif
    static int attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute((const void*)conv_k2_kernel<FM, BN, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC;
        attr = lds;
    }
    const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    conv_k2_kernel<FM, BN, NST><<<mt * nt, 512, lds, stream>>>(g, NRpad);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
// Which tile, if any?  A where its tiles fill the chip (>= 224), else D; layers with fewer than OCR_K2_MINSTEPS K steps (default 18)
// stay on conv_halo: with 9 steps a tile is mostly prologue and epilogue, and conv_halo's two independent workgroups per CU overlap
// those (measured: conv2 33 against 40 us; conv3_1 forward, 18 steps: 26.9 on conv_halo, 27.6 on conv_k2, 24.0 on conv_k3 — profiles/r03q).  OCR_K2_CFG = A / D forces one tile where it covers the shape.
static int k2_choose(long M, int H, int Cin, int Cout, int minsteps_k3 = -1 /* >= 0: the step floor of conv_k3.hip instead of OCR_K2_MINSTEPS */) {
    if ((Cin & 63) || (Cout & 63) || M < 4096 || H > 16 || H < 1) return 0;
    if (M * Cin * 2 >= 0x7fffffffL || (long)Cout * 9 * Cin * 2 >= 0x7fffffffL) return 0;   // 32-bit descriptor offsets
    static int force = -1, minsteps = -1, allow_a = -1, mintiles = -1;
    if (mintiles < 0) { const char* e = ocr_tune_env("OCR_K2_MINTILES"); mintiles = e ? atoi(e) : 128; }
    if (force < 0) { const char* e = getenv("OCR_K2_CFG"); force = (e && (e[0] == 'A' || e[0] == 'D')) ? e[0] : 0; }
    if (minsteps < 0) { const char* e = ocr_tune_env("OCR_K2_MINSTEPS"); minsteps = e ? atoi(e) : 18; }
    // OCR_K2_TILES = D keeps the dispatcher off tile A.  With prefetch distance 2 tile A was SLOWER than conv_halo inside the train step
    // (1.461 against 1.435 ms, profiles/r03k) and 15-25 % slower behind a cache scrub (profiles/r03m) although 6 % faster when the same
    // launch repeats back to back: one 8-wave workgroup per CU in lock step has nothing to run while a piece arrives late from HBM.
    // With distance 3 (five weight stages) it is faster in all three settings (profiles/r03n: cold 474 against 504 us over the ten
    // layers, hot 416 against 446, step 1.415 against 1.440 ms).
    if (allow_a < 0) { const char* e = ocr_tune_env("OCR_K2_TILES"); allow_a = (e && e[0] == 'D') ? 0 : 1; }
    // first a tile whose grid fills the chip (>= 224 workgroups: A, then D), else one that fills at least half of it (OCR_K2_MINTILES, 128:
    // the small-batch layers of configs[4], 6090 -> 6513 images/s; for the headline net preferring A with 128 tiles over D with 256 cost
    // 22 us per step, profiles/r03_final_a_* against r03y)
    const char order[2] = {'A', 'D'};
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < 2; ++i) {
            const char c = order[i];
            if (force && c != force) continue;
            if (!force && c == 'A' && !allow_a) continue;
            const int bn = c == 'A' ? 128 : 64;
            if (Cout % bn) continue;
            if (force) return c;
            if (9 * (Cin / 64) < (minsteps_k3 >= 0 ? minsteps_k3 : minsteps)) return 0;
            if ((M + 255) / 256 * (Cout / bn) >= (pass == 0 ? 224 : mintiles)) return c;
        }
    return 0;
}

int k3_try_dispatch(int tile, const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind);

// -1 = shape not covered (caller falls back to conv_halo.hip)
int k2_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind) {
    if (flags & ~(K2_BIAS | K2_RELU | K2_MASK | K2_ACCUM)) return -1;
    // sixth generation (conv_k3.hip: the same tiles with the halo stored as feature-row planes) where it covers the shape; its own step floor
    // OCR_K3_MINSTEPS (default 18 like conv_k2: the 9-step conv2 forward is no faster on it than on conv_halo's two workgroups per CU —
    // 34.0 against 33.6 us, 41 against 39.7 behind a cache scrub, profiles/r03t); A/B knob OCR_CONV_K3 = 0
    static int k3 = -1, k3min = -1;
    if (k3 < 0) { const char* e = getenv("OCR_CONV_K3"); k3 = e ? atoi(e) : 1; }
    if (k3min < 0) { const char* e = ocr_tune_env("OCR_K3_MINSTEPS"); k3min = e ? atoi(e) : 18; }
    if (k3) {
        const int c3 = k2_choose(M, H, Cin, Cout, k3min);
        if (c3) {
            const int rc = k3_try_dispatch(c3, x, wpack, y, M, W, H, Cin, Cout, bias, mask, flags, stream, pool, pool_kind);
            if (rc >= 0) return rc;
        }
    }
    if (pool_kind >= 3) return -1;                       // batch-norm statistics in the epilogue: conv_k3 only
    const int c = k2_choose(M, H, Cin, Cout);
    if (!c) return -1;
    static int abl = -1;
    if (abl < 0) { const char* e = ocr_tune_env("OCR_K2_ABL"); abl = e ? atoi(e) : 0; }
    K2Args g = {(const bf16_t*)x, (const bf16_t*)wpack, M, Cout, Cin, W, H, (bf16_t*)y, bias, (const bf16_t*)mask, flags, (bf16_t*)pool, pool_kind, abl};
#ifdef OCR_EXPERIMENTS
    static int nst = -1;                                 // A/B knob OCR_K2_NST: weight stages (prefetch distance + 2)
    if (nst < 0) { const char* e = ocr_tune_env("OCR_K2_NST"); nst = e ? atoi(e) : 0; }
    if (c == 'A' && nst == 4) return launch_k2<8, 128, 4>(g, stream);
    if (c == 'D' && nst == 5) return launch_k2<4, 64, 5>(g, stream);
    if (c == 'D' && nst == 6) return launch_k2<4, 64, 6>(g, stream);
#endif
    // tile A: five weight stages = prefetch distance 3 (see k2_choose); tile D: four — five or six change nothing for it (profiles/r03o)
    if (c == 'A') return launch_k2<8, 128, 5>(g, stream);
    return launch_k2<4, 64, 4>(g, stream);
}
