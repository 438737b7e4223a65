// Third-generation 3x3 SAME convolution weight gradient for gfx950 — all nine taps from ONE staged tile, no atomics.
//     dW[tap][ci][co] += sum_m X[m + shift(tap)][ci] * dY[m][co]          (reference: tf.gradients of network.py:166 at train.py:81)
//
// gemm_tn2.hip gives every (tap, 128 ci, 128 co) tile its own workgroup: the same dY rows cross L2 -> LDS nine times and the same
// X rows nine times (shifted), 32 KiB of LDS fill per 2.1 MFLOP — exactly the 64 B/clk/CU a CU can fill at the MFMA peak —
// and the split-M partial sums meet in fp32 atomics (20-28 MB per layer, a quarter of the kernel, and not bit-reproducible).
// Here a workgroup of 8 waves owns ALL NINE taps of a 64 ci x 64 co tile (36 864 fp32 results = 144 accumulator registers per
// lane on four of its waves, the other four hold the same tile for the other half of the pixels):
//   * per step of 128 pixels ONE halo tile of X (128 + 2H + 2 rows x 64 channels: every tap of every pixel of the step lies in
//     it, exactly as in conv_halo.hip) and ONE tile of dY (128 rows x 64 channels) are LDS-DMA'd: 34 KiB per 18.9 MFLOP, 8x
//     less fill per flop; the nine A operands are nine shifted views of the halo image, the B operand is shared by all taps;
//   * operands stay row-major [pixel][channel] (what DMA can write) and are transposed by ds_read_b64_tr_b16 on the way into
//     the MFMA (lane semantics pinned by tests/test_gpu_kernels.py::test_probe_tr16), k permutation as in gemm_tn.hip;
//     128-byte rows, 32-byte slot XOR ((row >> 1) & 3) on the DMA source chunk and on the read: any 8 consecutive rows x 32 B
//     of a 32-lane read half hit 64 distinct banks, for every tap shift;
//   * SAME padding without touching the loop: the feature axis H is 4, 8 or 16 tall and a lane's four transposed elements are
//     four consecutive pixels of ONE image column, so "pixel has no neighbour above / below" is a loop-invariant per-lane AND
//     mask on one register of the fragment; "no neighbour column" (w = 0 / W-1) is decided per 32-pixel block by a scalar
//     test and only then applied (rare: 2 blocks per image);
//   * waves (cb, kh): channel block cb = 16 ci, pixel half kh of the step; the halves are summed through LDS at the end
//     (halves the partial-sum traffic), the tile's partial result goes to a slab [split][9][Cin][Cout] with plain stores;
//   * a second kernel adds the S slabs into dW in a fixed order: deterministic, and the bias gradient (column sums of dY,
//     taken from the dY tile by the workgroups of ci tile 0) rides along.
// Covered: Cin % 64 == 0, Cout % 64 == 0, H in {4, 8, 16}.  Everything else stays on gemm_tn2.hip / gemm_tn.hip.
#include "common.h"
#include <stdlib.h>
#include <utility>
#include <type_traits>

struct W9Args {
    const bf16_t* X; const bf16_t* dY;       // [M][Cin], [M][Cout]
    int M, Cin, Cout, cW, cH;
    int k_per_split;                         // pixels per split, multiple of 128
    int S, T_ci, T_co, map;                  // splits; tiles; workgroup map (0 plain, 1 split = xcd (mod 8), 2 8/S XCDs per split)
    float* part;                             // [S][9][Cin][Cout]
    float* cs_part;                          // [S][Cout] or nullptr
};

__device__ u32x4 w9_zero_page[4];
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int W9_NDMA = 5;                        // LDS-DMA instructions per wave and step (1 KiB each)
constexpr int W9_STAGE = 8 * W9_NDMA * 1024;      // 40 KiB: halo rows | dY rows | spare
constexpr int W9_NST = 3;
constexpr int W9_LDS = 4 * 9 * 4 * 4 * 64 * 4;    // 147 456 B: the epilogue's half-sum exchange (> 3 stages + 8 KiB zero block)

#define W9_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define W9_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

// ---- the LDS read stream of one step, as compile-time tables --------------------------------------------------------------
// Groups in issue (= return) order: B(block 0) | A(0,0) .. A(0,8) | B(block 1) | A(1,0) .. A(1,8); a B group is 8 reads, an A
// group 2.  Tap n = 9 kk + t consumes A(kk,t) and B(kk).  Before tap n the stream is advanced to LA groups past A(n), as far as
// the 4-bit lgkmcnt allows (at most 15 reads in flight); the wait of tap n is the number of reads younger than A(n).
constexpr int w9_gsize(int p) { return (p == 0 || p == 10) ? 8 : 2; }
constexpr int w9_gpos(int n) { return n < 9 ? 1 + n : 2 + n; }
constexpr int w9_younger(int n, int upto) { int o = 0; for (int q = w9_gpos(n) + 1; q <= upto; ++q) o += w9_gsize(q); return o; }
constexpr int w9_upto(int n, int LA) {
    int p = w9_gpos(n) + LA;
    if (p > 19) p = 19;
    const int own = n == 0 ? 10 : 2;                // reads of A(n) itself (and of B(0) at the first tap) may still be in flight
    while (own + w9_younger(n, p) > 15) --p;
    return p;
}

// the same as tables (constant arrays indexed by the unrolled tap number fold to immediates; calls with loops may not)
struct W9Tab { int upto[18]; int wait[18]; };
constexpr W9Tab w9_make_tab(int LA) {
    W9Tab t = {};
    for (int n = 0; n < 18; ++n) { t.upto[n] = w9_upto(n, LA); t.wait[n] = w9_younger(n, t.upto[n]); }
    return t;
}
template <int LA> struct W9T { static constexpr W9Tab tab = w9_make_tab(LA); };

// continuous stream: positions 20.. are the next step's groups; at most 13 reads in flight (two more slots of the 4-bit
// counter are left to the column-sum reads); every read older than A(n) was waited for at tap n - 1
constexpr int w9c_younger(int n, int upto) { int o = 0; for (int q = w9_gpos(n) + 1; q <= upto; ++q) o += w9_gsize(q % 20); return o; }
constexpr int w9c_upto(int n, int LA) {
    int p = w9_gpos(n) + LA;
    if (n == 9) p = p;                                  // (B(1) sits before A(1,0): already counted by the positions)
    int own = 2;
    if (n == 0) own = 10;                               // B(0) and A(0,0) of this step were fetched by the previous step's last taps
    while (own + w9c_younger(n, p) > 13) --p;
    return p;
}
constexpr W9Tab w9c_make_tab(int LA) {
    W9Tab t = {};
    for (int n = 0; n < 18; ++n) { t.upto[n] = w9c_upto(n, LA); t.wait[n] = w9c_younger(n, t.upto[n]); }
    for (int n = 1; n < 18; ++n) if (t.upto[n] < t.upto[n - 1]) { t.upto[n] = t.upto[n - 1]; t.wait[n] = w9c_younger(n, t.upto[n]); }
    return t;
}
template <int LA> struct W9C { static constexpr W9Tab tab = w9c_make_tab(LA); };

__device__ long long* w9_dbg;                       // diagnostic: s_memtime stamps of workgroup 0 (ocr_wgrad9_debug)

template <int LA /* A groups kept in flight ahead of the tap being multiplied */, int DMA_AT /* tap before which the next stage's DMA is issued */,
          bool DBG, int ABL = 0 /* timing ablations (wrong results): 1 no fragment reads, 2 no DMA after the prologue, 3 no MFMA, 4 no masks */,
          int DMA_STEP = 0 /* 0: the five DMA pieces in one block before tap DMA_AT; k: one piece every k taps from DMA_AT on */,
          bool STAG = false /* waves of pixel half 1 issue their DMA block nine taps later than those of half 0 */,
          bool REDIR = false /* SAME padding along the feature axis by reading a zero row (address select, no ALU op on the fragment) */,
          bool SADDR = false /* DMA with a scalar base + 32-bit lane offset (needs REDIR: rows outside the tensor are then never read from LDS) */>
__global__ __launch_bounds__(512) void wgrad9_kernel(W9Args g) {
    constexpr int NSLOT = LA + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3, kh = wave >> 2;
    const int H = g.cH, W = g.cW;
    int split, tile;
    {
        const int b = blockIdx.x, T = g.T_ci * g.T_co;
        if (g.map == 1) { const int x = b & 7, q = b >> 3; split = (q / T) * 8 + x; tile = q % T; }
        else if (g.map == 2) { const int x = b & 7, q = b >> 3, G = 8 / g.S; split = x / G; tile = (x % G) * (T / G) + q; }
        else { split = b / T; tile = b % T; }
    }
    const int ti = tile / g.T_co, tj = tile % g.T_co;
    const int ci0 = ti * 64, co0 = tj * 64;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.M, kbeg + g.k_per_split);
    const int nsteps = kend > kbeg ? (kend - kbeg + 127) >> 7 : 0;
    const int NR = 128 + 2 * H + 2;                 // halo rows a step needs
    const int nhalo = (NR + 7) >> 3;                // DMA instructions covering them (8 rows of 128 B each)
    const int NRp = nhalo << 3;
    const bool do_cs = g.cs_part != nullptr && ti == 0;
    const bf16_t* zero = (const bf16_t*)w9_zero_page;

    // ---- DMA geometry: instruction u = wave + 8 i covers stage bytes [u KiB, (u + 1) KiB): lane -> row 8u' + (lane >> 3),
    //      LDS chunk position lane & 7, which holds SOURCE chunk q (slot XOR (row >> 1) & 3; (8u' + rr) >> 1 & 3 == rr >> 1 & 3)
    const int rr = lane >> 3, pp = lane & 7;
    const int qsrc = ((((pp >> 1) ^ ((rr >> 1) & 3)) << 1) | (pp & 1)) * 8;       // first channel of the source chunk
    const bf16_t* src[W9_NDMA];
    int pix[W9_NDMA];
#pragma unroll
    for (int i = 0; i < W9_NDMA; ++i) {
        const int u = wave + 8 * i;
        src[i] = zero; pix[i] = 0x40000000;            // spare piece: always the zero page
        if (u < nhalo) {
            const int r = 8 * u + rr;
            pix[i] = (r < NR) ? kbeg - (H + 1) + r : 0x40000000;
            src[i] = g.X + (long)(kbeg - (H + 1) + r) * g.Cin + ci0 + qsrc;
        } else if (u < nhalo + 16) {
            const int r = 8 * (u - nhalo) + rr;
            pix[i] = kbeg + r;
            src[i] = g.dY + (long)(kbeg + r) * g.Cout + co0 + qsrc;
        }
    }
    // SADDR form: byte offset of this lane's 16 bytes from the tensor base at step 0 (halo rows before the tensor wrap to a huge
    // unsigned value and are replaced below); 32 bits suffice: the plan admits M * C * 2 < 2^31
    unsigned off0[W9_NDMA];
#pragma unroll
    for (int i = 0; i < W9_NDMA; ++i) {
        const int u = wave + 8 * i;
        off0[i] = 0;
        if (u < nhalo) off0[i] = (unsigned)(((long)(kbeg - (H + 1) + 8 * u + rr) * g.Cin + ci0 + qsrc) * 2);
        else if (u < nhalo + 16) off0[i] = (unsigned)(((long)(kbeg + 8 * (u - nhalo) + rr) * g.Cout + co0 + qsrc) * 2);
    }
    const unsigned safeX = (unsigned)((ci0 + qsrc) * 2);            // a valid address for halo rows outside the tensor (never read: redirected)
    const bool tail = (kend & 127) != 0;                            // the last step of this split has dY rows past kend: zero page, old form
    auto stage_load = [&](int step, int buf) {
        unsigned char* st = smem + buf * W9_STAGE;
#pragma unroll
        for (int i = 0; i < W9_NDMA; ++i) {
            const int u = wave + 8 * i;
            const int px = pix[i] + step * 128;
            if (SADDR && u < nhalo) {
                unsigned vo = off0[i] + (unsigned)step * (unsigned)(128 * 2) * (unsigned)g.Cin;
                if ((unsigned)px >= (unsigned)g.M) vo = safeX;
                dma16_saddr(lds0 + buf * W9_STAGE + u * 1024, vo, g.X);
                continue;
            }
            if (SADDR && u < nhalo + 16 && !(tail && step == nsteps - 1)) {
                dma16_saddr(lds0 + buf * W9_STAGE + u * 1024, off0[i] + (unsigned)step * (unsigned)(128 * 2) * (unsigned)g.Cout, g.dY);
                continue;
            }
            const bf16_t* s = zero;
            if (u < nhalo) { if (px >= 0 && px < g.M) s = src[i] + (long)step * 128 * g.Cin; }
            else if (u < nhalo + 16) { if (px < kend) s = src[i] + (long)step * 128 * g.Cout; }
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(st + u * 1024), 16, 0, 0);
        }
    };

    auto dma_one = [&](int i, int step, int buf) {       // piece i of this wave (i is a constant after unrolling)
        const int u = wave + 8 * i;
        const int px = pix[i] + step * 128;
        const bf16_t* s = zero;
        if (u < nhalo) { if (px >= 0 && px < g.M) s = src[i] + (long)step * 128 * g.Cin; }
        else if (u < nhalo + 16) { if (px < kend) s = src[i] + (long)step * 128 * g.Cout; }
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + buf * W9_STAGE + u * 1024), 16, 0, 0);
    };

    // ---- fragment addressing (loop invariant): lane (g4, L) supplies row 4 g4 + (L >> 2) of a 4 x 16 block, 8-byte piece L & 3
    const int g4 = lane >> 4, L = lane & 15;
    const int rowl = kh * 64 + 4 * g4 + (L >> 2);
    unsigned offA[9], offB[4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int a0 = rowl + (H + 1) + (t / 3 - 1) * H + (t % 3 - 1);          // halo row of this lane's pixel for tap t
        offA[t] = a0 * 128 + ((cb ^ ((a0 >> 1) & 3)) << 5) + (L & 3) * 8;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) offB[c] = NRp * 128 + rowl * 128 + ((c ^ ((rowl >> 1) & 3)) << 5) + (L & 3) * 8;
    // SAME padding along the feature axis: element e of a transposed read is pixel row 4 g4 + e (+16, +32 k: same h, H | 16)
    const unsigned mlo = ((4 * g4) % H == 0) ? 0xffff0000u : 0xffffffffu;         // taps with dh = -1: element 0 has h == 0
    const unsigned mhi = ((4 * g4 + 4) % H == 0) ? 0x0000ffffu : 0xffffffffu;     // taps with dh = +1: element 3 has h == H - 1
    // REDIR: the lane that SUPPLIES a padded element to the transposing read (row L >> 2 of its 4 x 16 block) reads 8 zero bytes
    // instead — an address select folded into the loop-invariant per-tap offsets: offA of such a lane is absolute (zero block
    // behind the stages) and its stage base is 0.  Nothing touches the fragment between the read and the MFMA (the AND / select
    // version made the compiler copy every fragment into one operand tuple: each copy waited for the previous tap's MFMAs —
    // measured 11 us of 75 on conv4_2).
    constexpr unsigned ZOFF = W9_NST * W9_STAGE;                                   // 8 KiB of zeros (offsets up to 6 KiB are added)
    const int hsup = (4 * g4 + (L >> 2)) % H;                                     // feature row of the element this lane supplies (H = 2: a 4-row block spans two columns)
    const bool red_m = hsup == 0;                                                 // taps with dh = -1: no row above
    const bool red_p = hsup == H - 1;                                             // taps with dh = +1: no row below
    const unsigned zabs = lds0 + ZOFF + (L & 3) * 8;                              // this lane's 8 zero bytes
    if (REDIR) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dh = t % 3 - 1;
            if ((dh < 0 && red_m) || (dh > 0 && red_p)) offA[t] = lds0 + ZOFF + (offA[t] & 255);     // same banks as the real row (immediates are multiples of 256)
        }
        *(u32x4*)(smem + ZOFF + tid * 16) = (u32x4){0, 0, 0, 0};                  // 512 threads x 16 B (visible after the first barrier)
    }
    const int hs = H == 2 ? 1 : (H == 4 ? 2 : (H == 8 ? 3 : 4));                  // log2 H
    const int ncol = 32 >> hs;                                                   // image columns per 32-pixel block
    int wc = (kbeg >> hs) % W;                                                   // column (within its image) of the step's first pixel
    const int lc0 = (4 * g4 + (L >> 2)) >> hs, lc1 = (4 * g4 + (L >> 2) + 16) >> hs;   // column (inside a 32-pixel block) of the element this lane supplies, both reads

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // column-sum geometry: thread -> source chunk tid & 7 (8 channels), rows (tid >> 3) and (tid >> 3) + 64 of the dY tile
    const int csq = tid & 7, csr = tid >> 3;
    const unsigned offC = NRp * 128 + csr * 128 + (((((csq >> 1) ^ ((csr >> 1) & 3)) << 1) | (csq & 1)) << 4);   // (csr + 64) >> 1 & 3 same

#pragma unroll
    for (int p = 0; p < W9_NST - 1; ++p)
        if (p < nsteps) stage_load(p, p);
    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;
        if (DBG) st0 = __builtin_amdgcn_s_memtime();
        if (step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W9_NDMA) : "memory");     // the next step may stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DBG) st1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (DBG) st2 = __builtin_amdgcn_s_memtime();
        const unsigned sb = lds0 + cur * W9_STAGE;
        const unsigned sbm = (REDIR && red_m) ? 0u : sb, sbp = (REDIR && red_p) ? 0u : sb;      // stage base as seen by the dh = -1 / +1 taps

        s16x4 alo[NSLOT], ahi[NSLOT], blo[2][4], bhi[2][4];
        bool bnd = false;                    // does the current 32-pixel block touch image column 0 or W - 1 ? (wave-uniform)
        int w0 = 0, w1 = 0;                  // this lane's image column for the two reads (valid when bnd)
        // REDIR: "no neighbour column" (w == 0 for dw = -1, w == W - 1 for dw = +1) as an address select too, per 32-pixel block
        bool zl_m[2], zh_m[2], zl_p[2], zh_p[2];
        if (REDIR) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                int cbase = wc + ((kh * 64 + kb * 32) >> hs);
                while (cbase >= W) cbase -= W;
                const bool bd = cbase == 0 || cbase + ncol >= W;
                int c0 = cbase + lc0, c1 = cbase + lc1;
                if (c0 >= W) c0 -= W;
                if (c1 >= W) c1 -= W;
                zl_m[kb] = bd && c0 == 0; zh_m[kb] = bd && c1 == 0; zl_p[kb] = bd && c0 == W - 1; zh_p[kb] = bd && c1 == W - 1;
            }
        }
#define W9_ISSUE(P_) do { \
            if (ABL == 1 || ABL == 5) break; \
            if ((P_) == 0 || (P_) == 10) { \
                const int kb_ = (P_) == 0 ? 0 : 1; \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) { W9_TR(blo[kb_][c], sb + offB[c], kb_ * 32 * 128); W9_TR(bhi[kb_][c], sb + offB[c], kb_ * 32 * 128 + 16 * 128); } \
            } else { \
                const int m_ = (P_) < 10 ? (P_) - 1 : (P_) - 2, kb_ = m_ / 9, tt_ = m_ % 9; \
                const unsigned sa_ = (tt_ % 3 == 0 ? sbm : (tt_ % 3 == 2 ? sbp : sb)) + offA[tt_]; \
                unsigned sl_ = sa_, sh_ = sa_; \
                if (REDIR && tt_ / 3 == 0) { sl_ = zl_m[kb_] ? zabs : sa_; sh_ = zh_m[kb_] ? zabs : sa_; } \
                if (REDIR && tt_ / 3 == 2) { sl_ = zl_p[kb_] ? zabs : sa_; sh_ = zh_p[kb_] ? zabs : sa_; } \
                W9_TR(alo[m_ % NSLOT], sl_, kb_ * 32 * 128); W9_TR(ahi[m_ % NSLOT], sh_, kb_ * 32 * 128 + 16 * 128); \
            } } while (0)
#pragma unroll
        for (int P = 0; P < 20; ++P)
            if (P <= W9T<LA>::tab.upto[0]) W9_ISSUE(P);
#pragma unroll
        for (int n = 0; n < 18; ++n) {
            const int kk = n / 9, t = n % 9;
            if (DMA_STEP == 0 && (n == DMA_AT || (STAG && n == DMA_AT + 9)) && (!STAG || (n == DMA_AT) == (kh == 0))) {
                // the buffer of step + 2 was last read in step - 1: free since this step's barrier
                if (step + 2 < nsteps && ABL != 2 && ABL != 5) { int nb = cur + 2; if (nb >= W9_NST) nb -= W9_NST; stage_load(step + 2, nb); }
                if (DBG) st3 = __builtin_amdgcn_s_memtime();
            }
            if (DMA_STEP > 0 && n >= DMA_AT && (n - DMA_AT) % DMA_STEP == 0 && (n - DMA_AT) / DMA_STEP < W9_NDMA) {
                if (step + 2 < nsteps && ABL != 2) { int nb = cur + 2; if (nb >= W9_NST) nb -= W9_NST; dma_one((n - DMA_AT) / DMA_STEP, step + 2, nb); }
                if (DBG) st3 = __builtin_amdgcn_s_memtime();
            }
            if (t == 0 && !REDIR) {
                int cbase = wc + ((kh * 64 + kk * 32) >> hs);
                while (cbase >= W) cbase -= W;
                bnd = cbase == 0 || cbase + ncol >= W;
                w0 = cbase + lc0; w1 = cbase + lc1;           // < 2 W (the plan requires W >= 32 / H)
                if (w0 >= W) w0 -= W;
                if (w1 >= W) w1 -= W;
            }
            if (n > 0) {                         // advance the read stream
#pragma unroll
                for (int P = 0; P < 20; ++P)
                    if (P > W9T<LA>::tab.upto[n - 1] && P <= W9T<LA>::tab.upto[n]) W9_ISSUE(P);
            }
            switch (W9T<LA>::tab.wait[n]) {              // folds: n is a constant after unrolling
                case 0: W9_WAIT(0); break;   case 2: W9_WAIT(2); break;   case 4: W9_WAIT(4); break;   case 6: W9_WAIT(6); break;
                case 8: W9_WAIT(8); break;   case 10: W9_WAIT(10); break; case 12: W9_WAIT(12); break; default: W9_WAIT(14); break;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DBG && n == 0) st4 = __builtin_amdgcn_s_memtime();
            u32x2 lo = __builtin_bit_cast(u32x2, alo[n % NSLOT]), hi = __builtin_bit_cast(u32x2, ahi[n % NSLOT]);
            const int dh = t % 3 - 1, dw = t / 3 - 1;
            if (ABL == 1 || ABL == 5) { asm volatile("" : "+v"(lo), "+v"(hi)); }
            if (dh < 0 && ABL != 4 && !REDIR) { lo.x &= mlo; hi.x &= mlo; }
            if (dh > 0 && ABL != 4 && !REDIR) { lo.y &= mhi; hi.y &= mhi; }
            if (dw != 0 && bnd && ABL != 4 && !REDIR) {      // rare (two blocks per image): the neighbour column does not exist
                const int bad = dw < 0 ? 0 : W - 1;
                if (w0 == bad) { lo.x = 0; lo.y = 0; }
                if (w1 == bad) { hi.x = 0; hi.y = 0; }
            }
            const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
            const bf16x8 fa = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x2 bl = __builtin_bit_cast(u32x2, blo[kk][c]), bh = __builtin_bit_cast(u32x2, bhi[kk][c]);
                const u32x4 bv = {bl.x, bl.y, bh.x, bh.y};
                if (ABL == 3) { asm volatile("" :: "v"(av), "v"(bv)); continue; }
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, __builtin_bit_cast(bf16x8, bv), acc[t][c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef W9_ISSUE
        if (do_cs) {            // bias gradient: column sums of the dY tile (rows past kend are zero-filled)
            u32x4 v0, v1;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(sb + offC));
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(v1) : "v"(sb + offC));
            W9_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
            cs[0] += bf_lo(v0.x) + bf_lo(v1.x); cs[1] += bf_hi(v0.x) + bf_hi(v1.x); cs[2] += bf_lo(v0.y) + bf_lo(v1.y); cs[3] += bf_hi(v0.y) + bf_hi(v1.y);
            cs[4] += bf_lo(v0.z) + bf_lo(v1.z); cs[5] += bf_hi(v0.z) + bf_hi(v1.z); cs[6] += bf_lo(v0.w) + bf_lo(v1.w); cs[7] += bf_hi(v0.w) + bf_hi(v1.w);
        }
        if (DBG && blockIdx.x == 0 && lane == 0 && w9_dbg != nullptr && step < 64) {
            long long* d = w9_dbg + (wave * 64 + step) * 8;
            d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = st4; d[5] = __builtin_amdgcn_s_memtime();
        }
        wc += 128 >> hs;
        while (wc >= W) wc -= W;
        cur = (cur + 1 == W9_NST) ? 0 : cur + 1;
    }
    __syncthreads();                                   // every DMA has landed and every tile is dead: LDS is reused below

    if (do_cs) {
        float* red = (float*)smem;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 64) {                                // channel tid = chunk tid >> 3, element tid & 7; 64 row groups
            float s = 0.f;
            for (int u = 0; u < 64; ++u) s += red[(u * 8 + (tid >> 3)) * 8 + (tid & 7)];
            g.cs_part[(long)split * g.Cout + co0 + tid] = s;
        }
        __syncthreads();
    }
    // sum the two pixel halves: waves kh = 1 hand their accumulators over through LDS ([wave][tap][c][r][lane], conflict free).
    // (Tried: meeting in LDS as a [tap][ci][co] tile image — ds_add_f32 from the second half, 16-byte slab stores by all eight
    // waves: +40 us per layer, LDS float atomics are slow; the slab write is bound by its 37.7 MB anyway.)
    float* xch = (float*)smem + cb * (9 * 4 * 4 * 64) + lane;
    if (kh == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[((t * 4 + c) * 4 + r) * 64] = acc[t][c][r];
    }
    __syncthreads();
    if (kh == 0) {
        float* slab = g.part + (long)split * 9 * g.Cin * g.Cout;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + cb * 16 + g4 * 4 + r, co = co0 + c * 16 + L;
                    slab[((long)t * g.Cin + ci) * g.Cout + co] = acc[t][c][r] + xch[((t * 4 + c) * 4 + r) * 64];
                }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// PLANE-LAYOUT form of the kernel above (round 3; same slabs, same reduction, same results up to the summation order inside a step).
// wgrad9_kernel spends ~190 VALU instructions per wave and step beside its 72 MFMAs (per-tap address selects for the SAME padding,
// 64-bit DMA addresses with bounds tests): 4800 cycles per step against 2304 of MFMA, and VALU work does not hide behind the other
// wave's MFMAs.  As in conv_k3.hip, a step's 128 pixels are NC = 128 / H whole image columns, staged as H planes (plane h, row c' <-
// pixel (column c0 - 1 + c', feature row h); PS = NC + 2 rounded up to 8 rows per plane) and the dY rows in the same (h, column) order:
//   * the 32 pixels of an MFMA's K block are 32 columns of one plane (H = 4) or 16 columns of two planes (H = 8): tap (dw, dh) reads the
//     same rows shifted by dw in plane h + dh — a per-dw lane register (the slot XOR depends on ((c' + dw) >> 1) & 3 only) + an immediate;
//   * a plane that does not exist is read from the zero block with no select, and an MFMA whose 32 pixels are all padding is not issued
//     (H = 4: a sixth of them);
//   * image edges are the halo columns c' = 0 / NC + 1 of a step (W % NC == 0): their DMA lanes get bit 31 OR-ed into the lane offset
//     (out of range -> zeros) on the steps that start / end an image;
//   * DMA through buffer descriptors: loop-invariant 32-bit lane offsets, the step advance is a scalar offset.
// Covered: H in {4, 8}, W % (128 / H) == 0, M * C * 2 < 2^31; everything else stays on wgrad9_kernel.
#ifdef OCR_EXPERIMENTS
// diagnostic (experiments build, ocr_wgrad9_debug): wall-clock stamps (100 MHz) of every workgroup's first thread —
// dbg[block * 8 + {0 entry, 1 K loop done, 2 column sums done, 3 slab stores issued, 4 acknowledged}] (tools/w9p_phases.py)
#define W9P_PHASE(slot) do { if (w9_dbg && threadIdx.x == 0) w9_dbg[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define W9P_PHASE(slot) do { } while (0)
#endif
// GENW (round 6, configs[3]): any image width W >= NC — a step's NC columns may cross ONE image boundary: local column b of the step is the first
// column of the next image.  The planes are staged exactly as for whole-image steps (plane row 1 + c = local column c, the two halo rows); what
// differs is two elements of the contraction: the +1 tap of column b - 1 (the last column of image A) and the -1 tap of column b (the first of image
// B) must read zeros — the lanes that SUPPLY those two pixels are redirected, for that tap only, to one of the zero rows every plane carries behind
// its columns (rows NC + 2 .. PS - 1 arrive as zeros with every stage), so a fragment address stays "lane register + immediate" and nothing
// touches the DMA.  Per step: a uniform branch; in the steps that cross a boundary (40 % at W = 80, NC = 32) four compares and four selects.
// (First form, measured and replaced: conv_k3w's zero ROW inserted into the planes — the DMA lanes behind it fetch one column further left — cost
// ~45 VALU instructions per crossing step and ran conv4_2 at W = 80 in 78 us; profiles/r06h_ab_varwidth_first_form.log, r06h2_ab_varwidth.log.)
// Index algebra replayed on the CPU: tools/w9p_plane_model.py (a_fragment_rows_genw), tests/test_w9p_plane_model.py::test_general_width_boundary_redirect.
template <int H, int LA = 2 /* A groups kept in flight ahead of the tap being multiplied */, bool CONT = false /* continuous read stream: see runc */,
          bool GENW = false>
__global__ __launch_bounds__(512) void wgrad9p_kernel(W9Args g) {
    static_assert(!(CONT && GENW), "the continuous-stream schedule exists for whole-image steps only");
    constexpr int DMA_AT = 3, NSLOT = LA + 1;
    constexpr int NC = 128 / H;                         // image columns per step
    constexpr int PS = (NC + 2 + 7) / 8 * 8;            // rows per plane (40 / 24)
    constexpr int XROWS = H * PS, XPIECES = XROWS / 8;  // 160 rows = 20 pieces / 192 rows = 24 pieces; then 128 dY rows = 16 pieces
    static_assert(XPIECES + 16 <= 8 * W9_NDMA, "a stage holds the planes and the dY tile");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    W9P_PHASE(0);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3, kh = wave >> 2;
    const int W = g.cW;
    int split, tile;
    {
        const int b = blockIdx.x, T = g.T_ci * g.T_co;
        if (g.map == 1) { const int x = b & 7, q = b >> 3; split = (q / T) * 8 + x; tile = q % T; }
        else if (g.map == 2) { const int x = b & 7, q = b >> 3, G = 8 / g.S; split = x / G; tile = (x % G) * (T / G) + q; }
        else { split = b / T; tile = b % T; }
    }
    const int ti = tile / g.T_co, tj = tile % g.T_co;
    const int ci0 = ti * 64, co0 = tj * 64;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.M, kbeg + g.k_per_split);
    const int nsteps = kend > kbeg ? (kend - kbeg) >> 7 : 0;       // M % 128 == 0: whole steps
    const bool do_cs = g.cs_part != nullptr && ti == 0;

    // ---- DMA geometry: instruction u = wave + 8 i covers stage bytes [u KiB, (u + 1) KiB): lane -> row 8u + (lane >> 3), LDS chunk
    //      position lane & 7, which holds SOURCE chunk q (slot XOR (row >> 1) & 3; PS % 8 == 0: (row >> 1) & 3 == (c' >> 1) & 3)
    const int rr = lane >> 3, pp = lane & 7;
    const int qsrc = ((((pp >> 1) ^ ((rr >> 1) & 3)) << 1) | (pp & 1)) * 8;       // first channel of the source chunk
    // descriptors: X from H pixels before the split (the first halo column), dY from the split; lane offsets are relative to those
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(g.X + ((long)kbeg - H) * g.Cin), 0,
                                                                          (int)(((long)g.M - kbeg + H) * g.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)(g.dY + (long)kbeg * g.Cout), 0,
                                                                          (int)(((long)g.M - kbeg) * g.Cout * 2), 0x00020000);
    // GENW: zero rows of a plane the redirected lanes read: ZLO for the first read of a tap, ZHI + 16 (H = 4: the second read carries a 16-row immediate)
    constexpr int ZLO = NC + 4, ZHI = H == 4 ? NC + 4 - 16 : NC + 4;
    static_assert(NC + 2 <= ZLO && ZLO < PS, "a zero row exists behind the columns of every plane");
    unsigned voff[W9_NDMA], eL[W9_NDMA], eR[W9_NDMA];
#pragma unroll
    for (int i = 0; i < W9_NDMA; ++i) {
        const int u = wave + 8 * i;
        voff[i] = OOB; eL[i] = 0; eR[i] = 0;           // spare piece: zeros
        if (u < XPIECES) {
            const int r = 8 * u + rr, h = r / PS, cp = r % PS;
            if (cp < NC + 2) voff[i] = (unsigned)(((cp * H + h) * g.Cin + ci0 + qsrc) * 2);
            eL[i] = cp == 0 ? OOB : 0; eR[i] = cp == NC + 1 ? OOB : 0;
        } else if (u < XPIECES + 16) {
            const int r = 8 * (u - XPIECES) + rr, h = r / NC, col = r % NC;
            voff[i] = (unsigned)(((col * H + h) * g.Cout + co0 + qsrc) * 2);
        }
    }
    const int xstep = 128 * g.Cin * 2, ystep = 128 * g.Cout * 2;            // bytes per step
    const int wc0 = (kbeg / H) % W;                     // column (within its image) of the split's first pixel
    auto stage_load = [&](int step, int buf, int wcs /* column of that step's first pixel */) {
        const unsigned selL = wcs == 0 ? OOB : 0u, selR = wcs + NC == W ? OOB : 0u;
#pragma unroll
        for (int i = 0; i < W9_NDMA; ++i) {
            const int u = wave + 8 * i;
            lptr_t dst = (lptr_t)(smem + buf * W9_STAGE + u * 1024);
            if (u < XPIECES) {
                const unsigned vo = voff[i] | (eL[i] & selL) | (eR[i] & selR);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, dst, 16, (int)vo, step * xstep, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, dst, 16, (int)voff[i], step * ystep, 0, 0);
            }
        }
    };

    // ---- fragment addressing (loop invariant): lane (g4, L) supplies row 4 g4 + (L >> 2) of a 4 x 16 block, 8-byte piece L & 3
    const int g4 = lane >> 4, L = lane & 15;
    unsigned baseA[3], zabs[3], offB[4];
    constexpr unsigned ZOFF = W9_NST * W9_STAGE;       // 8 KiB of zeros
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int cp = 1 + 4 * g4 + (L >> 2) + (d - 1);                          // plane row of the element this lane supplies (+16 for the second read at H = 4)
        baseA[d] = cp * 128 + ((cb ^ ((cp >> 1) & 3)) << 5) + (L & 3) * 8;
        zabs[d] = lds0 + ZOFF + baseA[d];
    }
    const int clo = 4 * g4 + (L >> 2);                 // GENW: local column of the pixel this lane supplies (second read at H = 4: + 16)
    // ... and where it reads instead when that pixel's dw = -1 / +1 neighbour belongs to another image: a zero row, at the 8-byte piece it would have read
    const unsigned zlo0 = ZLO * 128 + (baseA[0] & 127), zlo2 = ZLO * 128 + (baseA[2] & 127);
    const unsigned zhi0 = ZHI * 128 + (baseA[0] & 127), zhi2 = ZHI * 128 + (baseA[2] & 127);
    const int rowl = kh * 64 + 4 * g4 + (L >> 2);
#pragma unroll
    for (int c = 0; c < 4; ++c) offB[c] = XROWS * 128 + rowl * 128 + ((c ^ ((rowl >> 1) & 3)) << 5) + (L & 3) * 8;
    *(u32x4*)(smem + ZOFF + tid * 16) = (u32x4){0, 0, 0, 0};                      // 512 threads x 16 B (visible after the first barrier)

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int csq = tid & 7, csr = tid >> 3;
    const unsigned offC = XROWS * 128 + csr * 128 + (((((csq >> 1) ^ ((csr >> 1) & 3)) << 1) | (csq & 1)) << 4);   // (csr + 64) >> 1 & 3 same

    int wcl = wc0;                                      // column of the next step to be loaded
    int wcc = wc0;                                      // GENW: column of the step being multiplied
#pragma unroll
    for (int p = 0; p < W9_NST - 1; ++p)
        if (p < nsteps) { stage_load(p, p, wcl); wcl += NC; if (wcl >= W) wcl -= W; }

    auto run = [&](auto khc) {
        constexpr int KH = decltype(khc)::value;
        int cur = 0;
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W9_NDMA) : "memory");     // the next step may stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned sb = lds0 + cur * W9_STAGE;
            unsigned sbA[3], sbAh[3], sbB[4];
#pragma unroll
            for (int d = 0; d < 3; ++d) { sbA[d] = sb + baseA[d]; sbAh[d] = sbA[d]; }
            if (GENW && W - wcc < NC) {
                // (uniform branch) this step crosses an image boundary at local column bq: the -1 tap of column bq and the +1 tap of column bq - 1 read zeros
                const int bq = W - wcc, chi = clo + (H == 4 ? 16 : 0);
                if (clo == bq) sbA[0] = sb + zlo0;
                if (clo == bq - 1) sbA[2] = sb + zlo2;
                if (chi == bq) sbAh[0] = sb + zhi0;
                if (chi == bq - 1) sbAh[2] = sb + zhi2;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) sbB[c] = sb + offB[c];
            s16x4 alo[NSLOT], ahi[NSLOT], blo[2][4], bhi[2][4];
            // planes of the two 16-pixel halves of K block kk, shifted by dh; -1: the plane does not exist
#define W9P_PLANE(kk_, dh_, half_) (H == 4 ? (KH * 2 + (kk_) + (dh_)) : (KH * 4 + (kk_) * 2 + (half_) + (dh_)))
#define W9P_OK(pl_) ((pl_) >= 0 && (pl_) < H)
#define W9P_ISSUE(P_) do { \
                if ((P_) == 0 || (P_) == 10) { \
                    const int kb_ = (P_) == 0 ? 0 : 1; \
                    _Pragma("unroll") for (int c = 0; c < 4; ++c) { W9_TR(blo[kb_][c], sbB[c], kb_ * 32 * 128); W9_TR(bhi[kb_][c], sbB[c], kb_ * 32 * 128 + 16 * 128); } \
                } else { \
                    constexpr int m_ = (P_) < 10 ? (P_) - 1 : (P_) - 2, kb_ = m_ / 9, tt_ = m_ % 9, dw_ = tt_ / 3, dh_ = tt_ % 3 - 1; \
                    constexpr int pl_ = W9P_PLANE(kb_, dh_, 0), ph_ = W9P_PLANE(kb_, dh_, 1); \
                    constexpr int il_ = W9P_OK(pl_) ? pl_ * PS * 128 : 0, ih_ = W9P_OK(ph_) ? ph_ * PS * 128 + (H == 4 ? 16 * 128 : 0) : (H == 4 ? 16 * 128 : 0); \
                    W9_TR(alo[m_ % NSLOT], W9P_OK(pl_) ? sbA[dw_] : zabs[dw_], il_); \
                    W9_TR(ahi[m_ % NSLOT], W9P_OK(ph_) ? sbAh[dw_] : zabs[dw_], ih_); \
                } } while (0)
            // the read stream as compile-time positions (tables of wgrad9_kernel): issue everything up to tab.upto[n] before tap n
            auto issue_to = [&](auto fromc, auto toc) {
                constexpr int FROM = decltype(fromc)::value, TO = decltype(toc)::value;
                if constexpr (FROM <= 0 && 0 <= TO) W9P_ISSUE(0);
                if constexpr (FROM <= 1 && 1 <= TO) W9P_ISSUE(1);
                if constexpr (FROM <= 2 && 2 <= TO) W9P_ISSUE(2);
                if constexpr (FROM <= 3 && 3 <= TO) W9P_ISSUE(3);
                if constexpr (FROM <= 4 && 4 <= TO) W9P_ISSUE(4);
                if constexpr (FROM <= 5 && 5 <= TO) W9P_ISSUE(5);
                if constexpr (FROM <= 6 && 6 <= TO) W9P_ISSUE(6);
                if constexpr (FROM <= 7 && 7 <= TO) W9P_ISSUE(7);
                if constexpr (FROM <= 8 && 8 <= TO) W9P_ISSUE(8);
                if constexpr (FROM <= 9 && 9 <= TO) W9P_ISSUE(9);
                if constexpr (FROM <= 10 && 10 <= TO) W9P_ISSUE(10);
                if constexpr (FROM <= 11 && 11 <= TO) W9P_ISSUE(11);
                if constexpr (FROM <= 12 && 12 <= TO) W9P_ISSUE(12);
                if constexpr (FROM <= 13 && 13 <= TO) W9P_ISSUE(13);
                if constexpr (FROM <= 14 && 14 <= TO) W9P_ISSUE(14);
                if constexpr (FROM <= 15 && 15 <= TO) W9P_ISSUE(15);
                if constexpr (FROM <= 16 && 16 <= TO) W9P_ISSUE(16);
                if constexpr (FROM <= 17 && 17 <= TO) W9P_ISSUE(17);
                if constexpr (FROM <= 18 && 18 <= TO) W9P_ISSUE(18);
                if constexpr (FROM <= 19 && 19 <= TO) W9P_ISSUE(19);
            };
            auto tap = [&](auto nc) {
                constexpr int n = decltype(nc)::value, kk = n / 9, t = n % 9, dh = t % 3 - 1;
                if (n == DMA_AT) {
                    // the buffer of step + 2 was last read in step - 1: free since this step's barrier
                    if (step + 2 < nsteps) { int nb = cur + 2; if (nb >= W9_NST) nb -= W9_NST; stage_load(step + 2, nb, wcl); wcl += NC; if (wcl >= W) wcl -= W; }
                }
                if constexpr (n == 0) issue_to(std::integral_constant<int, 0>{}, std::integral_constant<int, W9T<LA>::tab.upto[0]>{});
                else issue_to(std::integral_constant<int, W9T<LA>::tab.upto[n > 0 ? n - 1 : 0] + 1>{}, std::integral_constant<int, W9T<LA>::tab.upto[n]>{});
                switch (W9T<LA>::tab.wait[n]) {              // folds: n is a constant
                    case 0: W9_WAIT(0); break;   case 2: W9_WAIT(2); break;   case 4: W9_WAIT(4); break;   case 6: W9_WAIT(6); break;
                    case 8: W9_WAIT(8); break;   case 10: W9_WAIT(10); break; case 12: W9_WAIT(12); break; default: W9_WAIT(14); break;
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr bool live = W9P_OK(W9P_PLANE(kk, dh, 0)) || W9P_OK(W9P_PLANE(kk, dh, 1));
                if constexpr (live) {
                    const u32x2 lo = __builtin_bit_cast(u32x2, alo[n % NSLOT]), hi = __builtin_bit_cast(u32x2, ahi[n % NSLOT]);
                    const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x2 bl = __builtin_bit_cast(u32x2, blo[kk][c]), bh = __builtin_bit_cast(u32x2, bhi[kk][c]);
                        const u32x4 bv = {bl.x, bl.y, bh.x, bh.y};
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, __builtin_bit_cast(bf16x8, bv), acc[t][c], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            tap(std::integral_constant<int, 0>{});  tap(std::integral_constant<int, 1>{});  tap(std::integral_constant<int, 2>{});
            tap(std::integral_constant<int, 3>{});  tap(std::integral_constant<int, 4>{});  tap(std::integral_constant<int, 5>{});
            tap(std::integral_constant<int, 6>{});  tap(std::integral_constant<int, 7>{});  tap(std::integral_constant<int, 8>{});
            tap(std::integral_constant<int, 9>{});  tap(std::integral_constant<int, 10>{}); tap(std::integral_constant<int, 11>{});
            tap(std::integral_constant<int, 12>{}); tap(std::integral_constant<int, 13>{}); tap(std::integral_constant<int, 14>{});
            tap(std::integral_constant<int, 15>{}); tap(std::integral_constant<int, 16>{}); tap(std::integral_constant<int, 17>{});
#undef W9P_ISSUE
#undef W9P_OK
#undef W9P_PLANE
            if (do_cs) {            // bias gradient: column sums of the dY tile
                u32x4 v0, v1;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(sb + offC));
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(v1) : "v"(sb + offC));
                W9_WAIT(0);
                __builtin_amdgcn_sched_barrier(0);
                cs[0] += bf_lo(v0.x) + bf_lo(v1.x); cs[1] += bf_hi(v0.x) + bf_hi(v1.x); cs[2] += bf_lo(v0.y) + bf_lo(v1.y); cs[3] += bf_hi(v0.y) + bf_hi(v1.y);
                cs[4] += bf_lo(v0.z) + bf_lo(v1.z); cs[5] += bf_hi(v0.z) + bf_hi(v1.z); cs[6] += bf_lo(v0.w) + bf_lo(v1.w); cs[7] += bf_hi(v0.w) + bf_hi(v1.w);
            }
            cur = (cur + 1 == W9_NST) ? 0 : cur + 1;
            if (GENW) { wcc += NC; if (wcc >= W) wcc -= W; }
        }
    };
    // Continuous-stream schedule (CONT): the per-step barrier above stops the read stream — after it every wave first fetches B(0), A(0,0)..
    // with the matrix pipe idle, 3400 cycles per step against 1920 of MFMA (profiles/r03w_w9p_phases.log).  Here the barrier sits before tap
    // NB of a step ("stage step + 1 has landed for everyone" and "everyone has left step - 1", a whole step before either matters), the last
    // taps of a step already fetch the next step's first groups from the next stage (positions 20.. of the tables W9C), and the five DMA
    // pieces of stage step + 2 go out one at a time between the taps behind the barrier.
    auto runc = [&](auto khc) {
        constexpr int KH = decltype(khc)::value;
        constexpr int NB = 4, NSL = (LA + 1 <= 3) ? 3 : 6;
        static_assert(18 % NSL == 0 && LA + 1 <= NSL, "slot rotation closes over a step");
        if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W9_NDMA) : "memory");       // stage 0 landed (stage 1 may stay in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s16x4 alo[NSL], ahi[NSL], blo[2][4], bhi[2][4];
        u32x4 cv0 = {0, 0, 0, 0}, cv1 = {0, 0, 0, 0};
        unsigned sbA[3], sbB[4], nbA[3], nbB[4];       // fragment bases in the current and in the next stage
#pragma unroll
        for (int d = 0; d < 3; ++d) nbA[d] = lds0 + baseA[d];
#pragma unroll
        for (int c = 0; c < 4; ++c) nbB[c] = lds0 + offB[c];
#define W9P_PLANE(kk_, dh_, half_) (H == 4 ? (KH * 2 + (kk_) + (dh_)) : (KH * 4 + (kk_) * 2 + (half_) + (dh_)))
#define W9P_OK(pl_) ((pl_) >= 0 && (pl_) < H)
#define W9PC_ISSUE(P_, A_, B_) do { \
            constexpr int q_ = (P_) % 20; \
            if (q_ == 0 || q_ == 10) { \
                constexpr int kb_ = q_ == 0 ? 0 : 1; \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) { W9_TR(blo[kb_][c], B_[c], kb_ * 32 * 128); W9_TR(bhi[kb_][c], B_[c], kb_ * 32 * 128 + 16 * 128); } \
            } else { \
                constexpr int m_ = q_ < 10 ? q_ - 1 : q_ - 2, kb_ = m_ / 9, tt_ = m_ % 9, dw_ = tt_ / 3, dh_ = tt_ % 3 - 1; \
                constexpr int pl_ = W9P_PLANE(kb_, dh_, 0), ph_ = W9P_PLANE(kb_, dh_, 1); \
                constexpr int il_ = W9P_OK(pl_) ? pl_ * PS * 128 : 0, ih_ = W9P_OK(ph_) ? ph_ * PS * 128 + (H == 4 ? 16 * 128 : 0) : (H == 4 ? 16 * 128 : 0); \
                W9_TR(alo[m_ % NSL], W9P_OK(pl_) ? A_[dw_] : zabs[dw_], il_); \
                W9_TR(ahi[m_ % NSL], W9P_OK(ph_) ? A_[dw_] : zabs[dw_], ih_); \
            } } while (0)
        // positions FROM .. TO of the stream: 0..19 this step's groups (current stage), 20..39 the next step's (next stage; behind the last
        // step they are read all the same — from the same stage, unused — so that the counted waits stay exact)
        auto issue_to = [&](auto fromc, auto toc) {
            constexpr int FROM = decltype(fromc)::value, TO = decltype(toc)::value;
#define W9PC_AT(P_) if constexpr (FROM <= (P_) && (P_) <= TO) { if constexpr ((P_) < 20) W9PC_ISSUE(P_, sbA, sbB); else W9PC_ISSUE(P_, nbA, nbB); }
            W9PC_AT(0) W9PC_AT(1) W9PC_AT(2) W9PC_AT(3) W9PC_AT(4) W9PC_AT(5) W9PC_AT(6) W9PC_AT(7) W9PC_AT(8) W9PC_AT(9)
            W9PC_AT(10) W9PC_AT(11) W9PC_AT(12) W9PC_AT(13) W9PC_AT(14) W9PC_AT(15) W9PC_AT(16) W9PC_AT(17) W9PC_AT(18) W9PC_AT(19)
            W9PC_AT(20) W9PC_AT(21) W9PC_AT(22) W9PC_AT(23) W9PC_AT(24) W9PC_AT(25) W9PC_AT(26) W9PC_AT(27) W9PC_AT(28) W9PC_AT(29)
            W9PC_AT(30) W9PC_AT(31) W9PC_AT(32) W9PC_AT(33) W9PC_AT(34) W9PC_AT(35) W9PC_AT(36) W9PC_AT(37) W9PC_AT(38) W9PC_AT(39)
#undef W9PC_AT
        };
        issue_to(std::integral_constant<int, 20>{}, std::integral_constant<int, W9C<LA>::tab.upto[17]>{});     // read-stream prologue of step 0 (from stage 0)
        int cur = 0;
        for (int step = 0; step < nsteps; ++step) {
            const int nxt = (cur + 1 == W9_NST) ? 0 : cur + 1, nn = (nxt + 1 == W9_NST) ? 0 : nxt + 1;
            const bool more = step + 1 < nsteps, more2 = step + 2 < nsteps;
            const unsigned sb = lds0 + cur * W9_STAGE, sbn = lds0 + (more ? nxt : cur) * W9_STAGE;
#pragma unroll
            for (int d = 0; d < 3; ++d) { sbA[d] = nbA[d]; nbA[d] = sbn + baseA[d]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) { sbB[c] = nbB[c]; nbB[c] = sbn + offB[c]; }
            const unsigned selL = wcl == 0 ? OOB : 0u, selR = wcl + NC == W ? OOB : 0u;      // edges of the step whose stage is streamed during this one
            auto tap = [&](auto nc) {
                constexpr int n = decltype(nc)::value, kk = n / 9, t = n % 9, dh = t % 3 - 1;
                if constexpr (n == NB) {
                    // own DMA pieces of stage step + 1 (issued a step ago) have landed; behind the barrier that holds for every wave's pieces,
                    // and every wave has left step - 1: its buffer (the one of step + 2) may be refilled
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (do_cs) {                 // bias gradient: this step's dY tile, two 16-byte reads riding in the stream
                        asm volatile("ds_read_b128 %0, %1" : "=v"(cv0) : "v"(sb + offC));
                        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(cv1) : "v"(sb + offC));
                    }
                }
                if constexpr (n > NB && ((n - NB) & 1) && (n - NB) / 2 < W9_NDMA) {
                    if (more2) {
                        constexpr int i = (n - NB) / 2;
                        const int u = wave + 8 * i;
                        lptr_t dst = (lptr_t)(smem + nn * W9_STAGE + u * 1024);
                        if (u < XPIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, dst, 16, (int)(voff[i] | (eL[i] & selL) | (eR[i] & selR)), (step + 2) * xstep, 0, 0);
                        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, dst, 16, (int)voff[i], (step + 2) * ystep, 0, 0);
                    }
                }
                constexpr int prev = n == 0 ? W9C<LA>::tab.upto[17] - 20 : W9C<LA>::tab.upto[n > 0 ? n - 1 : 0];
                issue_to(std::integral_constant<int, prev + 1>{}, std::integral_constant<int, W9C<LA>::tab.upto[n]>{});
                switch (W9C<LA>::tab.wait[n]) {              // folds: n is a constant
                    case 0: W9_WAIT(0); break;   case 2: W9_WAIT(2); break;   case 4: W9_WAIT(4); break;   case 6: W9_WAIT(6); break;
                    case 8: W9_WAIT(8); break;   case 10: W9_WAIT(10); break; case 12: W9_WAIT(12); break; default: W9_WAIT(13); break;
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr bool live = W9P_OK(W9P_PLANE(kk, dh, 0)) || W9P_OK(W9P_PLANE(kk, dh, 1));
                if constexpr (live) {
                    const u32x2 lo = __builtin_bit_cast(u32x2, alo[n % NSL]), hi = __builtin_bit_cast(u32x2, ahi[n % NSL]);
                    const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x2 bl = __builtin_bit_cast(u32x2, blo[kk][c]), bh = __builtin_bit_cast(u32x2, bhi[kk][c]);
                        const u32x4 bv = {bl.x, bl.y, bh.x, bh.y};
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, __builtin_bit_cast(bf16x8, bv), acc[t][c], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            tap(std::integral_constant<int, 0>{});  tap(std::integral_constant<int, 1>{});  tap(std::integral_constant<int, 2>{});
            tap(std::integral_constant<int, 3>{});  tap(std::integral_constant<int, 4>{});  tap(std::integral_constant<int, 5>{});
            tap(std::integral_constant<int, 6>{});  tap(std::integral_constant<int, 7>{});  tap(std::integral_constant<int, 8>{});
            tap(std::integral_constant<int, 9>{});  tap(std::integral_constant<int, 10>{}); tap(std::integral_constant<int, 11>{});
            tap(std::integral_constant<int, 12>{}); tap(std::integral_constant<int, 13>{}); tap(std::integral_constant<int, 14>{});
            tap(std::integral_constant<int, 15>{}); tap(std::integral_constant<int, 16>{}); tap(std::integral_constant<int, 17>{});
            if (do_cs) {            // the two reads were issued 13 taps (>= 26 younger reads, all waited for in order) ago
                asm volatile("" : "+v"(cv0), "+v"(cv1));
                cs[0] += bf_lo(cv0.x) + bf_lo(cv1.x); cs[1] += bf_hi(cv0.x) + bf_hi(cv1.x); cs[2] += bf_lo(cv0.y) + bf_lo(cv1.y); cs[3] += bf_hi(cv0.y) + bf_hi(cv1.y);
                cs[4] += bf_lo(cv0.z) + bf_lo(cv1.z); cs[5] += bf_hi(cv0.z) + bf_hi(cv1.z); cs[6] += bf_lo(cv0.w) + bf_lo(cv1.w); cs[7] += bf_hi(cv0.w) + bf_hi(cv1.w);
            }
            if (more2) { wcl += NC; if (wcl >= W) wcl -= W; }
            cur = nxt;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef W9PC_ISSUE
#undef W9P_OK
#undef W9P_PLANE
    };
    if (CONT) { if (kh == 0) runc(std::integral_constant<int, 0>{}); else runc(std::integral_constant<int, 1>{}); }
    else { if (kh == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{}); }
    __syncthreads();                                   // every DMA has landed and every tile is dead: LDS is reused below
    W9P_PHASE(1);

    if (do_cs) {
        float* red = (float*)smem;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 64) {                                // channel tid = chunk tid >> 3, element tid & 7; 64 row groups
            float s = 0.f;
            for (int u = 0; u < 64; ++u) s += red[(u * 8 + (tid >> 3)) * 8 + (tid & 7)];
            g.cs_part[(long)split * g.Cout + co0 + tid] = s;
        }
        __syncthreads();
    }
    W9P_PHASE(2);
    // sum the two pixel halves through LDS and store the slab tile (as wgrad9_kernel)
    float* xch = (float*)smem + cb * (9 * 4 * 4 * 64) + lane;
    if (kh == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[((t * 4 + c) * 4 + r) * 64] = acc[t][c][r];
    }
    __syncthreads();
    if (kh == 0) {
        float* slab = g.part + (long)split * 9 * g.Cin * g.Cout;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + cb * 16 + g4 * 4 + r, co = co0 + c * 16 + L;
                    slab[((long)t * g.Cin + ci) * g.Cout + co] = acc[t][c][r] + xch[((t * 4 + c) * 4 + r) * 64];
                }
    }
#ifdef OCR_EXPERIMENTS
    W9P_PHASE(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W9P_PHASE(4);
#endif
}

// Continuous-stream form of the kernel above (measured with s_memtime stamps: after the per-step barrier every wave spent
// 750-1100 cycles issuing its five LDS-DMA pieces and 400-900 more waiting for the first fragments, with the matrix pipe
// idle; 4800 cycles per step against 2304 of MFMA).  Here
//   * the workgroup barrier sits in the MIDDLE of a step (before tap NB): it orders "stage step+1 has landed for everyone" and
//     "everyone has left step-1" (so its buffer may be refilled) — a whole step before anybody needs either;
//   * hence the LDS read stream never stops: the last taps of a step already fetch B(0), A(0,0).. of the next step from the
//     next stage, there is no read prologue behind a barrier;
//   * the five DMA pieces of stage step+2 are issued one at a time between the MFMA groups of taps NB+1, NB+3, ...
template <int LA /* A groups kept in flight */, int NSLOT /* A fragment slots: LA + 1 <= NSLOT, 18 % NSLOT == 0 */, int NB /* tap of the barrier */, bool DBG>
__global__ __launch_bounds__(512) void wgrad9c_kernel(W9Args g) {
    static_assert(18 % NSLOT == 0 && LA + 1 <= NSLOT && NB + 9 < 18, "slot rotation must close over a step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3, kh = wave >> 2;
    const int H = g.cH, W = g.cW;
    int split, tile;
    {
        const int b = blockIdx.x, T = g.T_ci * g.T_co;
        if (g.map == 1) { const int x = b & 7, q = b >> 3; split = (q / T) * 8 + x; tile = q % T; }
        else if (g.map == 2) { const int x = b & 7, q = b >> 3, G = 8 / g.S; split = x / G; tile = (x % G) * (T / G) + q; }
        else { split = b / T; tile = b % T; }
    }
    const int ti = tile / g.T_co, tj = tile % g.T_co;
    const int ci0 = ti * 64, co0 = tj * 64;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.M, kbeg + g.k_per_split);
    const int nsteps = kend > kbeg ? (kend - kbeg + 127) >> 7 : 0;
    const int NR = 128 + 2 * H + 2;                 // halo rows a step needs
    const int nhalo = (NR + 7) >> 3;                // DMA instructions covering them (8 rows of 128 B each)
    const int NRp = nhalo << 3;
    const bool do_cs = g.cs_part != nullptr && ti == 0;
    const bf16_t* zero = (const bf16_t*)w9_zero_page;

    // ---- DMA geometry: instruction u = wave + 8 i covers stage bytes [u KiB, (u + 1) KiB): lane -> row 8u' + (lane >> 3),
    //      LDS chunk position lane & 7, which holds SOURCE chunk q (slot XOR (row >> 1) & 3; (8u' + rr) >> 1 & 3 == rr >> 1 & 3)
    const int rr = lane >> 3, pp = lane & 7;
    const int qsrc = ((((pp >> 1) ^ ((rr >> 1) & 3)) << 1) | (pp & 1)) * 8;       // first channel of the source chunk
    const bf16_t* src[W9_NDMA];
    int pix[W9_NDMA];
#pragma unroll
    for (int i = 0; i < W9_NDMA; ++i) {
        const int u = wave + 8 * i;
        src[i] = zero; pix[i] = 0x40000000;            // spare piece: always the zero page
        if (u < nhalo) {
            const int r = 8 * u + rr;
            pix[i] = (r < NR) ? kbeg - (H + 1) + r : 0x40000000;
            src[i] = g.X + (long)(kbeg - (H + 1) + r) * g.Cin + ci0 + qsrc;
        } else if (u < nhalo + 16) {
            const int r = 8 * (u - nhalo) + rr;
            pix[i] = kbeg + r;
            src[i] = g.dY + (long)(kbeg + r) * g.Cout + co0 + qsrc;
        }
    }
    auto stage_load = [&](int step, int buf) {
        unsigned char* st = smem + buf * W9_STAGE;
#pragma unroll
        for (int i = 0; i < W9_NDMA; ++i) {
            const int u = wave + 8 * i;
            const int px = pix[i] + step * 128;
            const bf16_t* s = zero;
            if (u < nhalo) { if (px >= 0 && px < g.M) s = src[i] + (long)step * 128 * g.Cin; }
            else if (u < nhalo + 16) { if (px < kend) s = src[i] + (long)step * 128 * g.Cout; }
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(st + u * 1024), 16, 0, 0);
        }
    };

    auto dma_one = [&](int i, int step, int buf) {       // piece i of this wave (compile-time i after unrolling)
        const int u = wave + 8 * i;
        const int px = pix[i] + step * 128;
        const bf16_t* s = zero;
        if (u < nhalo) { if (px >= 0 && px < g.M) s = src[i] + (long)step * 128 * g.Cin; }
        else if (u < nhalo + 16) { if (px < kend) s = src[i] + (long)step * 128 * g.Cout; }
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + buf * W9_STAGE + u * 1024), 16, 0, 0);
    };

    // ---- fragment addressing (loop invariant): lane (g4, L) supplies row 4 g4 + (L >> 2) of a 4 x 16 block, 8-byte piece L & 3
    const int g4 = lane >> 4, L = lane & 15;
    const int rowl = kh * 64 + 4 * g4 + (L >> 2);
    unsigned offA[9], offB[4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int a0 = rowl + (H + 1) + (t / 3 - 1) * H + (t % 3 - 1);          // halo row of this lane's pixel for tap t
        offA[t] = a0 * 128 + ((cb ^ ((a0 >> 1) & 3)) << 5) + (L & 3) * 8;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) offB[c] = NRp * 128 + rowl * 128 + ((c ^ ((rowl >> 1) & 3)) << 5) + (L & 3) * 8;
    // SAME padding along the feature axis: element e of a transposed read is pixel row 4 g4 + e (+16, +32 k: same h, H | 16)
    const unsigned mlo = ((4 * g4) % H == 0) ? 0xffff0000u : 0xffffffffu;         // taps with dh = -1: element 0 has h == 0
    const unsigned mhi = ((4 * g4 + 4) % H == 0) ? 0x0000ffffu : 0xffffffffu;     // taps with dh = +1: element 3 has h == H - 1
    const int hs = H == 2 ? 1 : (H == 4 ? 2 : (H == 8 ? 3 : 4));                  // log2 H
    const int ncol = 32 >> hs;                                                   // image columns per 32-pixel block
    int wc = (kbeg >> hs) % W;                                                   // column (within its image) of the step's first pixel
    const int lc0 = (4 * g4 + (L >> 2)) >> hs, lc1 = (4 * g4 + (L >> 2) + 16) >> hs;   // column (inside a 32-pixel block) of the element this lane supplies, both reads

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    // column-sum geometry: thread -> source chunk tid & 7 (8 channels), rows (tid >> 3) and (tid >> 3) + 64 of the dY tile
    const int csq = tid & 7, csr = tid >> 3;
    const unsigned offC = NRp * 128 + csr * 128 + (((((csq >> 1) ^ ((csr >> 1) & 3)) << 1) | (csq & 1)) << 4);   // (csr + 64) >> 1 & 3 same

#pragma unroll
    for (int p = 0; p < W9_NST - 1; ++p)
        if (p < nsteps) stage_load(p, p);
    if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W9_NDMA) : "memory");       // stage 0 landed (stage 1 may stay in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    s16x4 alo[NSLOT], ahi[NSLOT], blo[2][4], bhi[2][4];
    u32x4 cv0 = {0, 0, 0, 0}, cv1 = {0, 0, 0, 0};
#define W9_ISSUE(P_, SB_) do { \
        const int q_ = (P_) % 20; \
        if (q_ == 0 || q_ == 10) { \
            const int kb_ = q_ == 0 ? 0 : 1; \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) { W9_TR(blo[kb_][c], (SB_) + offB[c], kb_ * 32 * 128); W9_TR(bhi[kb_][c], (SB_) + offB[c], kb_ * 32 * 128 + 16 * 128); } \
        } else { \
            const int m_ = q_ < 10 ? q_ - 1 : q_ - 2, kb_ = m_ / 9, tt_ = m_ % 9; \
            W9_TR(alo[m_ % NSLOT], (SB_) + offA[tt_], kb_ * 32 * 128); W9_TR(ahi[m_ % NSLOT], (SB_) + offA[tt_], kb_ * 32 * 128 + 16 * 128); \
        } } while (0)
    if (nsteps > 0) {                        // read-stream prologue of step 0: what the last tap of a step fetches for its successor
#pragma unroll
        for (int P = 20; P < 40; ++P)
            if (P <= W9C<LA>::tab.upto[17]) W9_ISSUE(P, lds0);
    }
    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;
        if (DBG) st0 = __builtin_amdgcn_s_memtime();
        const int nxt = (cur + 1 == W9_NST) ? 0 : cur + 1, nn = (nxt + 1 == W9_NST) ? 0 : nxt + 1;
        const unsigned sb = lds0 + cur * W9_STAGE, sbn = lds0 + nxt * W9_STAGE;
        const bool more = step + 1 < nsteps, more2 = step + 2 < nsteps;
        bool bnd = false;                    // does the current 32-pixel block touch image column 0 or W - 1 ? (wave-uniform)
        int w0 = 0, w1 = 0;                  // this lane's image column for the two reads (valid when bnd)
#pragma unroll
        for (int n = 0; n < 18; ++n) {
            const int kk = n / 9, t = n % 9;
            if (n == NB) {
                // own DMA pieces of stage step+1 (issued a step ago) have landed; after the barrier that holds for every wave's
                // pieces, and every wave has left step-1: its buffer (= the one of step+2) may be refilled
                if (DBG) st1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (DBG) st2 = __builtin_amdgcn_s_memtime();
                if (do_cs) {                 // bias gradient: this step's dY tile, two 16-byte reads riding in the stream
                    asm volatile("ds_read_b128 %0, %1" : "=v"(cv0) : "v"(sb + offC));
                    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(cv1) : "v"(sb + offC));
                }
            }
            if (n > NB && ((n - NB) & 1) && (n - NB) / 2 < W9_NDMA && more2) dma_one((n - NB) / 2, step + 2, nn);
            if (DBG && n == NB + 2 * W9_NDMA) st3 = __builtin_amdgcn_s_memtime();
            if (t == 0) {
                int cbase = wc + ((kh * 64 + kk * 32) >> hs);
                while (cbase >= W) cbase -= W;
                bnd = cbase == 0 || cbase + ncol >= W;
                w0 = cbase + lc0; w1 = cbase + lc1;           // < 2 W (the plan requires W >= 32 / H)
                if (w0 >= W) w0 -= W;
                if (w1 >= W) w1 -= W;
            }
            // advance the read stream: positions 0..19 are this step's groups, 20.. the next step's (from the next stage)
#pragma unroll
            for (int P = 0; P < 40; ++P) {
                const int prev = n == 0 ? W9C<LA>::tab.upto[17] - 20 : W9C<LA>::tab.upto[n - 1];
                if (P > prev && P <= W9C<LA>::tab.upto[n]) {
                    if (P < 20) W9_ISSUE(P, sb);
                    else if (more) W9_ISSUE(P, sbn);
                }
            }
            switch (W9C<LA>::tab.wait[n]) {              // folds: n is a constant after unrolling
                case 0: W9_WAIT(0); break;   case 2: W9_WAIT(2); break;   case 4: W9_WAIT(4); break;   case 6: W9_WAIT(6); break;
                case 8: W9_WAIT(8); break;   case 10: W9_WAIT(10); break; case 12: W9_WAIT(12); break; default: W9_WAIT(13); break;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DBG && n == 0) st4 = __builtin_amdgcn_s_memtime();
            u32x2 lo = __builtin_bit_cast(u32x2, alo[n % NSLOT]), hi = __builtin_bit_cast(u32x2, ahi[n % NSLOT]);
            const int dh = t % 3 - 1, dw = t / 3 - 1;
            if (dh < 0) { lo.x &= mlo; hi.x &= mlo; }
            if (dh > 0) { lo.y &= mhi; hi.y &= mhi; }
            if (dw != 0 && bnd) {                // rare (two blocks per image): the neighbour column does not exist
                const int bad = dw < 0 ? 0 : W - 1;
                if (w0 == bad) { lo.x = 0; lo.y = 0; }
                if (w1 == bad) { hi.x = 0; hi.y = 0; }
            }
            const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
            const bf16x8 fa = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x2 bl = __builtin_bit_cast(u32x2, blo[kk][c]), bh = __builtin_bit_cast(u32x2, bhi[kk][c]);
                const u32x4 bv = {bl.x, bl.y, bh.x, bh.y};
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, __builtin_bit_cast(bf16x8, bv), acc[t][c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_cs) {            // the two reads were issued 12 taps (>= 24 younger reads, all waited for in order) ago
            asm volatile("" : "+v"(cv0), "+v"(cv1));
            cs[0] += bf_lo(cv0.x) + bf_lo(cv1.x); cs[1] += bf_hi(cv0.x) + bf_hi(cv1.x); cs[2] += bf_lo(cv0.y) + bf_lo(cv1.y); cs[3] += bf_hi(cv0.y) + bf_hi(cv1.y);
            cs[4] += bf_lo(cv0.z) + bf_lo(cv1.z); cs[5] += bf_hi(cv0.z) + bf_hi(cv1.z); cs[6] += bf_lo(cv0.w) + bf_lo(cv1.w); cs[7] += bf_hi(cv0.w) + bf_hi(cv1.w);
        }
        if (DBG && blockIdx.x == 0 && lane == 0 && w9_dbg != nullptr && step < 64) {
            long long* d = w9_dbg + (wave * 64 + step) * 8;
            d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = st4; d[5] = __builtin_amdgcn_s_memtime();
        }
        wc += 128 >> hs;
        while (wc >= W) wc -= W;
        cur = nxt;
    }
#undef W9_ISSUE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                   // every DMA has landed and every tile is dead: LDS is reused below

    if (do_cs) {
        float* red = (float*)smem;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 64) {                                // channel tid = chunk tid >> 3, element tid & 7; 64 row groups
            float s = 0.f;
            for (int u = 0; u < 64; ++u) s += red[(u * 8 + (tid >> 3)) * 8 + (tid & 7)];
            g.cs_part[(long)split * g.Cout + co0 + tid] = s;
        }
        __syncthreads();
    }
    // sum the two pixel halves: waves kh = 1 hand their accumulators over through LDS ([wave][tap][c][r][lane], conflict free).
    // (Tried: meeting in LDS as a [tap][ci][co] tile image — ds_add_f32 from the second half, 16-byte slab stores by all eight
    // waves: +40 us per layer, LDS float atomics are slow; the slab write is bound by its 37.7 MB anyway.)
    float* xch = (float*)smem + cb * (9 * 4 * 4 * 64) + lane;
    if (kh == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[((t * 4 + c) * 4 + r) * 64] = acc[t][c][r];
    }
    __syncthreads();
    if (kh == 0) {
        float* slab = g.part + (long)split * 9 * g.Cin * g.Cout;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + cb * 16 + g4 * 4 + r, co = co0 + c * 16 + L;
                    slab[((long)t * g.Cin + ci) * g.Cout + co] = acc[t][c][r] + xch[((t * 4 + c) * 4 + r) * 64];
                }
    }
}

// dw[i] += sum_s part[s][i];  dbias[co] += sum_s cs_part[s][co] — fixed summation order: deterministic.
// 256 threads = `rows` thread rows x (256 / rows) float4 columns, rows = S / 8 clamped to [1, 8]: every thread adds up to eight slabs
// (s = row, row + rows, ...) with all its loads in flight, then the rows meet in LDS.  (One load per thread — eight rows for any S —
// ran the S = 4 .. 16 layers at 2.8 TB/s; a plain loop over all slabs per column left conv2's S = 64 to a handful of CUs.)
__device__ __forceinline__ void w9_reduce_body(float* __restrict__ dw, const float* __restrict__ part, long n4, long slab4, int S, int rows,
                                               float* __restrict__ dbias, const float* __restrict__ cs_part, int Cout, int blk) {
    __shared__ f32x4 red[256];
    const int cols = 256 / rows;
    const int col = threadIdx.x % cols, row = threadIdx.x / cols;
    const long i = (long)blk * cols + col;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, t = a;
    if (i < n4) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, c = b, d = b;
        const f32x4* p = (const f32x4*)part + i;
        // (round 5: the slabs are read exactly once — non-temporal loads keep them from displacing what the next kernels re-read from L2 — and
        //  the running gradient is fetched BEFORE the rows meet, not behind the barrier; same summation order, bit-identical results)
        if (row == 0) t = ((const f32x4*)dw)[i];
        int s = row;
        for (; s + 3 * rows < S; s += 4 * rows) {        // four slabs per round, the rounds independent of each other
            a += __builtin_nontemporal_load(p + (long)s * slab4);              b += __builtin_nontemporal_load(p + (long)(s + rows) * slab4);
            c += __builtin_nontemporal_load(p + (long)(s + 2 * rows) * slab4); d += __builtin_nontemporal_load(p + (long)(s + 3 * rows) * slab4);
        }
        for (; s < S; s += rows) a += __builtin_nontemporal_load(p + (long)s * slab4);
        a = (a + b) + (c + d);
    }
    red[row * cols + col] = a;
    __syncthreads();
    if (row == 0 && i < n4) {
        for (int r = 0; r < rows; ++r) t += red[r * cols + col];
        ((f32x4*)dw)[i] = t;
    }
    if (dbias != nullptr && blk == 0)
        for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
            float t = dbias[c];
            for (int s = 0; s < S; ++s) t += cs_part[(long)s * Cout + c];
            dbias[c] = t;
        }
}
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(float* __restrict__ dw, const float* __restrict__ part, long n4, long slab4,
                                                            int S, int rows, float* __restrict__ dbias, const float* __restrict__ cs_part, int Cout) {
    w9_reduce_body(dw, part, n4, slab4, S, rows, dbias, cs_part, Cout, blockIdx.x);
}
// The reductions of SEVERAL layers in one launch (ocr_wgrad9_reduce_jobs): the per-layer reduce kernels are short (10 - 19 us, the
// smallest ones bound by their launch, not by their 19 MB) and each ends a dependent-kernel boundary; a backward pass that
// keeps every layer's slabs until its end pays for one launch instead of five.  Same per-element summation order as the
// per-layer kernel, hence bit-identical results.
struct W9ReduceJob {        // 64 bytes, mirrored by lstm_ctc_ocr_amd/engine.py (numpy structured dtype)
    float* dw; const float* part; float* dbias; const float* cs_part;
    long n4, slab4;
    int S, rows, Cout, block_start;
};
__global__ __launch_bounds__(256) void wgrad9_reduce_jobs_kernel(const W9ReduceJob* __restrict__ jobs, int njobs) {
    // the block's job = the last one whose block_start <= blockIdx.x, looked up by all threads at once (round 6: the serial scan was one dependent L2
    // round trip per job in front of every block; nn_ops.hip::pack_jobs_kernel)
    __shared__ int jsel[4];
    int j = 0;
    for (int k = threadIdx.x; k < njobs; k += 256)
        if (jobs[k].block_start <= (int)blockIdx.x) j = k;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) j = max(j, __shfl_xor(j, off));
    if ((threadIdx.x & 63) == 0) jsel[threadIdx.x >> 6] = j;
    __syncthreads();
    j = max(max(jsel[0], jsel[1]), max(jsel[2], jsel[3]));
    const W9ReduceJob jb = jobs[j];
    w9_reduce_body(jb.dw, jb.part, jb.n4, jb.slab4, jb.S, jb.rows, jb.dbias, jb.cs_part, jb.Cout, (int)blockIdx.x - jb.block_start);
}

struct W9Plan { int S, k_per_split, map, T_ci, T_co; size_t bytes; };

static bool w9_plan(int M, int W, int H, int Cin, int Cout, W9Plan* p) {
    if ((Cin & 63) || (Cout & 63) || (H != 2 && H != 4 && H != 8 && H != 16) || M < 512 || (M % H) || W * H < 32) return false;
    const int T_ci = Cin / 64, T_co = Cout / 64, T = T_ci * T_co;
    static int smax = -1;                       // A/B knob OCR_W9_SMAX: cap on the split count (partial slabs cost 2 x 147 KB x workgroups of HBM traffic)
    if (smax < 0) { const char* e = ocr_tune_env("OCR_W9_SMAX"); smax = e ? atoi(e) : 64; if (smax < 1) smax = 1; }
    int S = 1;
    while ((long)S * 2 * T <= 256 && S * 2 <= smax && M / (S * 2) >= 512) S *= 2;   // <= one workgroup per CU, >= four steps each
    p->S = S; p->T_ci = T_ci; p->T_co = T_co;
    p->k_per_split = ceil_div(ceil_div(M, S), 128) * 128;
    p->map = S >= 8 ? 1 : ((T % (8 / S)) == 0 ? 2 : 0);
    p->bytes = (size_t)S * ((size_t)9 * Cin * Cout + Cout) * sizeof(float);
    return true;
}

extern "C" int ocr_wgrad9_debug(void* dbg /* device int64[8 waves][64 steps][8] or NULL */) {
    long long* q = (long long*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(w9_dbg), &q, sizeof(q)) == hipSuccess ? OCR_OK : OCR_ERR_MEMOPS;
}

extern "C" int ocr_conv3x3_wgrad_workspace_size(int Nb, int W, int H, int Cin, int Cout, size_t* bytes) {
    if (!bytes || Nb <= 0 || W <= 0 || H <= 0 || Cin <= 0 || Cout <= 0) return OCR_ERR_INVALID;
    W9Plan p;
    *bytes = w9_plan(Nb * W * H, W, H, Cin, Cout, &p) ? p.bytes : 0;
    return OCR_OK;
}

// -1: shape not covered or workspace too small (caller falls back to the atomics kernels)
int wgrad9_try_dispatch(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin, int Cout,
                        void* workspace, size_t ws_bytes, hipStream_t stream, void* defer_job, int* defer_blocks) {
    W9Plan p;
    const int M = Nb * W * H;
    if (!workspace || !w9_plan(M, W, H, Cin, Cout, &p) || ws_bytes < p.bytes) return -1;
    static int variant = -1;                    // A/B knob OCR_W9_VARIANT (see the table below); default 0
    if (variant < 0) { const char* e = ocr_tune_env("OCR_W9_VARIANT"); variant = e ? atoi(e) : 0; }
    W9Args g = {};
    g.X = (const bf16_t*)x; g.dY = (const bf16_t*)dy; g.M = M; g.Cin = Cin; g.Cout = Cout; g.cW = W; g.cH = H;
    g.k_per_split = p.k_per_split; g.S = p.S; g.T_ci = p.T_ci; g.T_co = p.T_co; g.map = p.map;
    g.part = (float*)workspace;
    g.cs_part = dbias ? g.part + (size_t)p.S * 9 * Cin * Cout : nullptr;
    const int grid = p.S * p.T_ci * p.T_co;
#define W9_LAUNCH(LA_, AT_, DBG_, ...) do { \
        static bool attr = false; \
        if (!attr) { if (hipFuncSetAttribute((const void*)wgrad9_kernel<LA_, AT_, DBG_, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, W9_LDS) != hipSuccess) return OCR_ERR_EXEC; attr = true; } \
        wgrad9_kernel<LA_, AT_, DBG_, ##__VA_ARGS__><<<grid, 512, W9_LDS, stream>>>(g); } while (0)
#define W9C_LAUNCH(LA_, NS_, NB_, DBG_) do { \
        static bool attr = false; \
        if (!attr) { if (hipFuncSetAttribute((const void*)wgrad9c_kernel<LA_, NS_, NB_, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, W9_LDS) != hipSuccess) return OCR_ERR_EXEC; attr = true; } \
        wgrad9c_kernel<LA_, NS_, NB_, DBG_><<<grid, 512, W9_LDS, stream>>>(g); } while (0)
    // plane-layout kernel where it covers the shape (A/B knob OCR_W9_PLANES = 0: wgrad9_kernel everywhere)
    static int planes = -1;
    if (planes < 0) { const char* e = getenv("OCR_W9_PLANES"); planes = e ? atoi(e) : 1; }
    // whole-image steps (W % NC == 0) or, round 6, any W >= NC through the zero-row instances (GENW)
    static int genw = -1;                       // A/B knob OCR_W9P_GENW = 0: general widths stay on wgrad9_kernel; 2: the zero-row instances on whole-image shapes too (tests, timing)
    if (genw < 0) { const char* e = getenv("OCR_W9P_GENW"); genw = e ? atoi(e) : 1; }
    const bool whole = (H == 4 || H == 8) && W % (128 / H) == 0;
    const bool use_p = planes && variant == 0 && (H == 4 || H == 8) && (whole || (genw && W >= 128 / H)) && M % 128 == 0 &&
                       (long)M * (Cin > Cout ? Cin : Cout) * 2 < 0x7fffffffL;
    if (use_p) {
        // (look-ahead 3 and 4 of the fragment read stream measured equal to 2: profiles/r03s_wgrad9p.log)
#define W9P_LAUNCH(H_, LA_, C_, ...) do { \
            static bool attr = false; \
            if (!attr) { if (hipFuncSetAttribute((const void*)wgrad9p_kernel<H_, LA_, C_, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, W9_LDS) != hipSuccess) return OCR_ERR_EXEC; attr = true; } \
            wgrad9p_kernel<H_, LA_, C_, ##__VA_ARGS__><<<grid, 512, W9_LDS, stream>>>(g); } while (0)
        // (the continuous-read-stream schedule `runc` — barrier in the middle of a step, the next step's first fragments fetched by the
        // last taps — measured equal or slower, look-ahead 2 and 5: profiles/r03x_wgrad9p_cont.log; only `make EXPERIMENTS=1` builds it)
#ifdef OCR_EXPERIMENTS
        static int cont = -1;                   // A/B knob OCR_W9P_CONT: 0 barrier at the top of a step, 1 continuous stream (look-ahead 2), 2 (look-ahead 5)
        if (cont < 0) { const char* e = getenv("OCR_W9P_CONT"); cont = e ? atoi(e) : 0; }
        if (cont == 1 || cont == 2) {
            if (H == 4) { if (cont == 1) W9P_LAUNCH(4, 2, true); else W9P_LAUNCH(4, 5, true); }
            else { if (cont == 1) W9P_LAUNCH(8, 2, true); else W9P_LAUNCH(8, 5, true); }
        } else
#endif
        if (!whole || (genw == 2 && W >= 128 / H)) { if (H == 4) W9P_LAUNCH(4, 2, false, true); else W9P_LAUNCH(8, 2, false, true); }
        else if (H == 4) W9P_LAUNCH(4, 2, false); else W9P_LAUNCH(8, 2, false);
#undef W9P_LAUNCH
    } else
    switch (H == 2 ? 0 : variant) {             // H = 2 (a 4-row read block spans two image columns): only the redirecting default handles it
#ifdef OCR_EXPERIMENTS      // timing variants / ablations of tools/w9_variants.py (round 2, all measured slower or equal)
        case 1: W9_LAUNCH(3, 0, false); break;
        case 2: W9_LAUNCH(4, 0, false); break;
        case 3: W9_LAUNCH(2, 3, false); break;
        case 4: W9_LAUNCH(4, 3, false); break;
        case 5: W9_LAUNCH(6, 3, false); break;
        case 6: W9C_LAUNCH(2, 3, 4, false); break;
        case 7: W9C_LAUNCH(5, 6, 4, false); break;
        case 8: W9C_LAUNCH(3, 6, 4, false); break;
        case 9: W9C_LAUNCH(2, 3, 2, false); break;
        case 16: W9C_LAUNCH(2, 3, 4, true); break;
        case 17: W9C_LAUNCH(5, 6, 4, true); break;
        case 10: W9_LAUNCH(2, 0, true); break;
        case 30: W9_LAUNCH(2, 3, false, 0, 0, false, true); break;      // redirect masks
        case 31: W9_LAUNCH(2, 3, false, 0, 0, true, true); break;       // + DMA block staggered between the pixel halves
        case 32: W9_LAUNCH(3, 3, false, 0, 0, false, true); break;      // LA 3
        case 33: W9_LAUNCH(3, 3, false, 0, 0, true, true); break;
        case 34: W9_LAUNCH(2, 1, false, 0, 0, true, true); break;
        case 35: W9_LAUNCH(2, 3, false, 5, 0, false, true); break;      // ablation: MFMA only (no reads, no DMA)
        case 36: W9_LAUNCH(2, 3, false, 1, 0, false, true); break;      // ablation: no reads
        case 37: W9_LAUNCH(2, 3, false, 2, 0, false, true); break;      // ablation: no DMA
        case 21: W9_LAUNCH(2, 3, false, 1); break;
        case 22: W9_LAUNCH(2, 3, false, 2); break;
        case 23: W9_LAUNCH(2, 3, false, 3); break;
        case 24: W9_LAUNCH(2, 3, false, 4); break;
        case 14: W9_LAUNCH(4, 3, true); break;
        case 40: W9_LAUNCH(2, 0, false); break;                         // first version: AND / select masks, DMA right behind the barrier
        case 42:                                                        // scalar-base (SADDR) DMA form: measured 4 % slower (284 vs 272 us over the five layers)
            if ((long)M * (Cin > Cout ? Cin : Cout) * 2 < 0x7fffffffL) W9_LAUNCH(2, 3, false, 0, 0, false, true, true);
            else W9_LAUNCH(2, 3, false, 0, 0, false, true);
            break;
#endif
        default: W9_LAUNCH(2, 3, false, 0, 0, false, true); break;      // measured best (r2): look-ahead 2, DMA block before tap 3, zero-row padding
    }
#undef W9_LAUNCH
#undef W9C_LAUNCH
    OCR_CHECK_LAUNCH();
    const long n4 = (long)9 * Cin * Cout / 4;
    int rows = p.S / 8;
    rows = rows < 1 ? 1 : (rows > 8 ? 8 : rows);
    while (rows & (rows - 1)) rows &= rows - 1;                       // power of two
    const int cols = 256 / rows;
    const int blocks = (int)((n4 + cols - 1) / cols);
    if (defer_job != nullptr) {                                       // the caller runs this reduction later, with others, in one launch
        W9ReduceJob jb = {dw, g.part, dbias, g.cs_part, n4, n4, p.S, rows, Cout, 0};
        *(W9ReduceJob*)defer_job = jb;
        *defer_blocks = blocks;
        return OCR_OK;
    }
    wgrad9_reduce_kernel<<<blocks, 256, 0, stream>>>(dw, g.part, n4, n4, p.S, rows, dbias, g.cs_part, Cout);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" int ocr_wgrad9_reduce_jobs(const void* jobs, int njobs, int total_blocks, void* stream) {
    static_assert(sizeof(W9ReduceJob) == 64, "W9ReduceJob is mirrored by engine.py");
    if (!jobs || njobs <= 0 || total_blocks <= 0) return OCR_ERR_INVALID;
    wgrad9_reduce_jobs_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const W9ReduceJob*)jobs, njobs);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
