// 3x3 SAME implicit-GEMM convolution for gfx950, sixth generation: conv_k2.hip's tiles and schedule with the halo image stored as
// FEATURE-ROW PLANES, so that the inner loop has no vector address arithmetic and multiplies no padding
// (same contract as ocr_conv3x3_bf16 — reference lib/networks/network.py:160-191 forward, and its data gradient).
//
// Why (round 3, profiles/r03p_k2_stamps.log and the per-interval instruction mix of conv_k2): with DMA, fragment reads AND MFMAs removed
// a K step of conv_k2 still took 1100 of its 2030 cycles.  Every step re-derived eight fragment addresses per wave — "is tap t of my
// pixel inside the image" is a per-lane predicate when 16 consecutive pixels (several image columns) form a fragment, so each address
// was a mask test + select, the compiler parked the 72 lane masks in spilled SGPR pairs (two v_readlane per fragment), and the counted
// DMA waits were a branch tree on a runtime piece count.  VALU work does not hide behind the other wave's MFMAs on this SIMD.
//
// Here a 16-pixel fragment is 16 CONSECUTIVE IMAGE COLUMNS AT ONE FEATURE ROW h, and the halo tile of a chunk is laid out in LDS as
// H planes of PS rows (PS = columns of the tile + 2, rounded up to 8): plane h, row c' <- pixel (column cbase - 1 + c', feature row h).
//   * tap (dw, dh) of fragment (h, column block) is the same 16 rows shifted by dw rows in plane h + dh: the address is a
//     loop-invariant lane register (one per dw: the XOR swizzle depends on (c' + dw) & 7 only, PS % 8 == 0) plus an IMMEDIATE;
//   * SAME padding along the feature axis: plane h + dh does not exist -> the whole fragment contributes nothing for that tap -> its
//     reads and MFMAs are not issued (H = 4: a sixth of all MFMAs, H = 8: a twelfth);
//   * SAME padding along the time axis: a tile is NC = 256 / H whole columns of ONE image (W % NC == 0), image edges can only be the
//     halo columns c' = 0 / NC + 1 — their DMA lanes carry an out-of-range offset and arrive as zeros;
//   * every wave issues the same, compile-time number of DMA pieces per step (P buffers padded to 40 pieces, the pieces of a chunk that
//     does not exist go through a zero-length descriptor): the counted vmcnt waits are immediates, the step is straight-line code.
// Tiles, wave roles (2 x 2 x 2 K or 4 x 1 x 2 K), the ping-pong of the K halves with one barrier per step, weight stages / prefetch
// distance and the K-half exchange are conv_k2.hip's.  Two more things were measured here first (and copied to conv_k2 where they apply):
//   * the wave group that multiplies FIRST in an interval runs its MFMAs at s_setprio 2, the other at 1: with equal priorities the two
//     bursts of a SIMD interleave, both finish late and the first group's load segment runs behind them (1260 clocks per step against 853
//     of MFMA; skewed ~1040: profiles/r03af_conv_k3_prio.log);
//   * the output tile is staged through LDS as a bf16 [256 pixels][BN] image and leaves as 16-byte row stores; mask and fused max-pool work
//     on that image (store phase 4.8 -> 2.8 us per launch, 8.9 -> 4.2 with the mask: profiles/r03u / r03v_k3_phases_*.log).
// Covered: H in {4, 8, 16} with W % (256 / H) == 0; H in {2, 4, 8} with any W whose boundary rows fit the plane (GENW, see the template
// parameter); M % 256 == 0, Cin % 64 == 0, Cout % 64 == 0; everything else stays on conv_k2 / conv_halo.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

enum { K3_BIAS = 1, K3_RELU = 2, K3_MASK = 16, K3_ACCUM = 64 };

struct K3Args {
    const bf16_t* P; const bf16_t* Q;     // P [M pixels][C] ; Q [N][9*C]
    int M, N, C;                          // C % 64 == 0
    int cW, cH;
    bf16_t* out; const float* bias; const bf16_t* mask; int flags;
    bf16_t* pool; int pool_kind;          // as conv_halo: 0 none, 1 feature pairs (1 x 2), 2 2 x 2; 3 = `pool` is float [M / 256][2][N]: per-tile
                                          // partial sums / sums of squares of the stored bf16 outputs (batch-norm statistics, nn_ops.hip)
    int prio;                             // MFMA priority of waves 4-7 (the K half that multiplies FIRST in an interval); waves 0-3 use 1
    // pool_kind 4 (a data gradient whose producer is a batch-norm + ReLU layer): the stored values are g = mask(y > 0) * result, and `pool` receives
    // the partial sums of the batch-norm BACKWARD pass per tile: [M / 256][2][N] = (sum g, sum g * xhat), xhat = (bnz - bn_mean) * bn_rstd
    const bf16_t* bnz; const float* bn_mean; const float* bn_rstd;
};
// what `pool` points to (HOST memory) when k3_try_dispatch is called with pool_kind 4
struct K3BnBwd { float* partials; const void* z; const float* mean; const float* rstd; };

#ifdef OCR_EXPERIMENTS
// diagnostic (experiments build): wall-clock stamps (100 MHz) of every workgroup's first thread — dbg[block * 8 + {0 entry, 1 prologue landed,
// 2 K loop done, 3 K halves exchanged, 4 stores issued, 5 stores acknowledged}] (tools/k3_phases.py)
__device__ long long* k3_dbg;
extern "C" int ocr_conv_k3_debug(void* dbg) {
    long long* q = (long long*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(k3_dbg), &q, sizeof(q)) == hipSuccess ? OCR_OK : OCR_ERR_MEMOPS;
}
#define K3_PHASE(slot) do { if (k3_dbg && threadIdx.x == 0) k3_dbg[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define K3_PHASE(slot) do { } while (0)
#endif
// diagnostic (ocr_conv_halo_clock_debug sets it together with conv_halo's): workgroup 0 stamps {shader-clock counter, 100 MHz wall clock} at
// entry and exit, so bench.py reports the clock the plane-layout kernels really ran at (round 3 measured it inside conv_halo launches only)
__device__ long long* g_k3_clk;
int k3_set_clock_debug(void* dbg) {
    long long* p = (long long*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k3_clk), &p, sizeof(p)) == hipSuccess ? OCR_OK : OCR_ERR_EXEC;
}
// packed bf16 max (exact: keeps the raw bits of the larger half)
__device__ __forceinline__ uint32_t k3_max2(uint32_t a, uint32_t b) {
    const uint32_t lo = (bf_lo(b) > bf_lo(a)) ? (b & 0xffffu) : (a & 0xffffu);
    const uint32_t hi = (bf_hi(b) > bf_hi(a)) ? (b & 0xffff0000u) : (a & 0xffff0000u);
    return lo | hi;
}
__device__ __forceinline__ u32x4 k3_max8(u32x4 a, u32x4 b) {
    u32x4 r = {k3_max2(a.x, b.x), k3_max2(a.y, b.y), k3_max2(a.z, b.z), k3_max2(a.w, b.w)};
    return r;
}
typedef __attribute__((address_space(3))) void* lptr_t;
#define K3_OOB 0x80000000u                // lane offset of a row that does not exist: beyond any descriptor's range
#define K3_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int FM /* 16-pixel fragments per wave: 8 or 4 */, int BN /* channels per tile: 128 (waves 2 pixel x 2 channel x 2 K) or 64 (4 x 1 x 2 K) */,
          int NST /* weight stages; tile s + NST - 2 is streamed during step s */, int H /* feature rows: 4, 8 or 16; 2 in the general-width form */,
          bool SINGLE = false /* C == 64: one chunk, one halo buffer, no halo pieces inside the loop */,
          bool GENW = false /* any image width: a tile's columns may cross image boundaries — a zero row sits in every plane in front of each
                               interior boundary column, local column c at plane row 1 + c + k(c), k(c) = boundaries in [1, c]; the fragment
                               base is then a lane register per (dw, column block) */,
          bool BNB = false /* pool_kind 4 (batch-norm backward sums in the write-out): its own instances — compiled into the common epilogue its
                              48 extra live registers spilled in the 256 x 128 kernels */>
__device__ __forceinline__ void k3_body(const K3Args& g) {
    constexpr int NW = 8, FN = 4, BM = 256;
    constexpr int WN = BN / 64, WMW = 4 / WN;           // waves along channels / pixels (per K half)
    static_assert(FM * WMW == 16, "256-pixel tiles");
    constexpr int NC = BM / H, NCB = NC / 16;           // image columns per tile; 16-column blocks
    constexpr int PS = (NC + 2 + 7) / 8 * 8;            // rows per plane
    constexpr int PPIECES = (H * PS / 8 + NW - 1) / NW * NW;                    // 40 (H = 4, 8) / 48 (H = 16) pieces of 8 rows: every wave
    constexpr int PBYTES = PPIECES * 1024, PI = PPIECES / NW;                   // streams the same PI = 5 / 6 pieces per chunk
    static_assert(H * PS <= PPIECES * 8 && PI <= 8, "planes fit the padded buffer; one piece per tap");
    constexpr int CBW = NCB / WMW > 0 ? NCB / WMW : 1;  // column blocks per wave group
    constexpr int HW = FM / CBW;                        // planes per wave group
    constexpr int WGC = NCB / CBW;                      // wave groups along the columns
    constexpr bool STATIC_H = HW == H;                  // else (256 x 64 tiles at H = 8) a wave group owns the upper or the lower four planes
    constexpr int QI = BN / 64;                         // weight DMA pieces per wave and step (2 / 1)
    constexpr int QB = BN * 128, QTOT = NST * QB;
    constexpr int DEPTH = NST - 2;                      // prefetch distance of the weight tiles, in steps
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // [NST weight stages][2 halo buffers]

    const int tid = threadIdx.x, lane = tid & 63;
    K3_PHASE(0);
    long long* const clk = g_k3_clk;
    if (clk && blockIdx.x == 0 && tid == 0) ocr_clk_enter(clk);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wm = (wave & 3) / WN, wn = (wave & 3) % WN;
    const int C = g.C;
    const int cb0 = (wm % WGC) * CBW, hbase = (wm / WGC) * HW;
    const bool top = STATIC_H || hbase == 0, bot = STATIC_H || hbase + HW == H;

    const int mtiles = g.M / BM, ntiles = (g.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int Lb = blockIdx.x;
    if ((nblk & 7) == 0) Lb = (Lb & 7) * (nblk >> 3) + (Lb >> 3);        // XCD-aware: each XCD gets a contiguous run of tiles
    const int m0 = (Lb / ntiles) * BM, n0 = (Lb % ntiles) * BN;
    const int col0 = m0 / H;                            // first column of the tile, counted over the whole batch (image = col / W)
    const bool edge_l = col0 % g.cW == 0, edge_r = (col0 + NC) % g.cW == 0;

    const int rsub = lane >> 3;
    const int csrc = ((lane & 7) ^ rsub) * 8;           // LDS position lane&7 of row r holds source chunk (lane&7)^(r&7)

    const int pbytes_all = (int)((long)g.M * C * 2);
    const __amdgpu_buffer_rsrc_t qsrd = __builtin_amdgcn_make_buffer_rsrc((void*)g.Q, 0, (int)((long)g.N * 9 * C * 2), 0x00020000);
    unsigned voffQ[QI], voffP[PI];
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int n = n0 + (wave * QI + j) * 8 + rsub;
        voffQ[j] = n < g.N ? (unsigned)(((long)n * 9 * C + csrc) * 2) : K3_OOB;
    }
    // GENW: interior boundary columns of this tile t_i = b0 + i W (b0 = W - col0 % W in 1 .. W); k(c) = #{i : c >= t_i}
    constexpr int KMAX = 8;
    const int b0w = g.cW - col0 % g.cW;
    auto kof = [&](int c) { int k = 0;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) k += (c >= b0w + i * g.cW) ? 1 : 0;
        return k; };
#pragma unroll
    for (int j = 0; j < PI; ++j) {                      // this wave's halo pieces are j * 8 + wave
        const int r = (j * NW + wave) * 8 + rsub;       // buffer row: plane r / PS, plane row r % PS (r & 7 == rsub == plane row & 7)
        const int h = r / PS, cp = r % PS;
        if (!GENW) {
            const bool ok = h < H && cp < NC + 2 && !(cp == 0 && edge_l) && !(cp == NC + 1 && edge_r);
            const long m = (long)(col0 - 1 + cp) * H + h;
            voffP[j] = ok ? (unsigned)((m * C + csrc) * 2) : K3_OOB;
        } else {
            // plane row cp -> local column: q = cp - 1 counts columns AND the zero rows in front of it; zero row i sits at q = t_i + i
            const int q = cp - 1, last = NC + kof(NC - 1);           // q of the right halo column
            int n = 0; bool pad = false;
#pragma unroll
            for (int i = 0; i < KMAX; ++i) { const int z = b0w + i * g.cW + i; n += (q >= z) ? 1 : 0; pad = pad || q == z; }
            int col = col0 + q - n;                                  // q in [0, last): a column of the tile unless a zero row
            bool ok = h < H && q >= 0 && q < last && !pad;
            if (h < H && q == -1) { ok = !edge_l; col = col0 - 1; }
            if (h < H && q == last) { ok = !edge_r; col = col0 + NC; }
            const long m = (long)col * H + h;
            voffP[j] = ok ? (unsigned)((m * C + csrc) * 2) : K3_OOB;
        }
    }
    auto load_q = [&](int k0 /* tap*C + chunk*64 */, int stage) {
#pragma unroll
        for (int j = 0; j < QI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qsrd, (lptr_t)(smem + stage * QB + (wave * QI + j) * 1024), 16, (int)voffQ[j], k0 * 2, 0, 0);
    };
    auto load_p = [&](const __amdgpu_buffer_rsrc_t srd, int chunk, int buf, int j, unsigned voff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(smem + QTOT + buf * PBYTES + (j * NW + wave) * 1024), 16, (int)voff, chunk * 128, 0, 0);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int kq = (kh << 2) | fq;                      // this lane's 16-byte chunk of a 128-byte row (the wave's K half)
    const int nchunks = C / 64;
    const int nsteps = nchunks * 9;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned qfrag0 = lds0 + (wn * 64 + frow) * 128 + ((kq ^ fx) << 4);                             // + stage*QB, + a*2048
    // pixel fragments: lane register per dw (row c' = cb0*16 + frow + 1 + dw of plane hbase - 1), immediate per (fragment, dh)
    constexpr int NPB = GENW ? CBW : 1;                 // GENW: the zero rows shift the column blocks of a wave group differently
    unsigned pbase[3][NPB];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int cbl = 0; cbl < NPB; ++cbl) {
            const int c = (cb0 + cbl) * 16 + frow;      // local column of this lane's pixel
            const int cp = c + d + (GENW ? kof(c) : 0); // = plane row of the column + dw
            pbase[d][cbl] = lds0 + QTOT + ((hbase - 1) * PS + cp) * 128 + ((kq ^ (cp & 7)) << 4);         // QTOT >= PS * 128: never below lds0
        }

    // ---- prologue: weight tiles of steps 0 .. DEPTH - 1, the whole halo of chunk 0; everything has landed before barrier 0
    {
        const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc((void*)g.P, 0, pbytes_all, 0x00020000);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load_q(d * C, d);                    // (a step that does not exist streams rows nobody reads)
#pragma unroll
        for (int j = 0; j < PI; ++j) load_p(psrd, 0, 0, j, voffP[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    K3_PHASE(1);

    u32x4 afr[FN], bfr[FM];                             // fragments: waves 4-7 carry them across the barrier
    int qs = 0 /* s % NST */, s = 0;
    // is fragment b multiplied at feature shift dh?  (compile time where a wave group owns all H planes)
    auto live = [&](int b, int dh) -> bool {
        const int hl = b / CBW + dh;
        return !((hl < 0 && top) || (hl >= HW && bot));
    };
    // LOAD(s): the fragment reads of step s, then the DMA issue — a fixed number of pieces per tap: QI weights + one halo piece at taps 1 .. 5
    auto load = [&](auto tapc, int chunk) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int DW = TAP / 3, DH = TAP % 3 - 1;
        const unsigned qa = qfrag0 + qs * QB;
        unsigned pa[NPB];
#pragma unroll
        for (int cbl = 0; cbl < NPB; ++cbl) pa[cbl] = pbase[DW][cbl] + (chunk & 1) * PBYTES;
#pragma unroll
        for (int a = 0; a < FN; ++a)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afr[a]) : "v"(qa), "n"(a * 2048));
#pragma unroll
        for (int b = 0; b < FM; ++b)
            if (live(b, DH)) {
                if (GENW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[b]) : "v"(pa[b % NPB]), "n"(((b / CBW + DH + 1) * PS) * 128));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[b]) : "v"(pa[0]), "n"(((b / CBW + DH + 1) * PS + (b % CBW) * 16) * 128));
            }
        constexpr int T2 = (TAP + DEPTH) % 9;
        const int c2 = chunk + (TAP + DEPTH >= 9 ? 1 : 0);
        int q2 = qs + DEPTH; if (q2 >= NST) q2 -= NST;
        load_q(T2 * C + c2 * 64, q2);                   // past the last step: rows nobody reads into a stage nobody reads (or zeros)
        if (!SINGLE && TAP >= 1 && TAP <= PI) {         // the next chunk's halo: a zero-length descriptor when there is no next chunk
            const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc((void*)g.P, 0, chunk + 1 < nchunks ? pbytes_all : 0, 0x00020000);
            constexpr int J = TAP >= 1 && TAP <= PI ? TAP - 1 : 0;
            load_p(psrd, chunk + 1, (chunk + 1) & 1, J, voffP[J]);
        }
    };
    // COMP: the MFMAs on the fragments in registers — nothing else
    auto comp = [&](auto tapc, auto khc) {
        constexpr int DH = decltype(tapc)::value % 3 - 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // waves 4-7 multiply first in an interval and load second: with a HIGHER priority than waves 0-3 their 32 MFMAs drain before the
        // partner's start, so their load segment runs beside the partner's MFMAs instead of behind both interleaved MFMA bursts
        // (equal priorities: 1260 clocks per step against 853 of MFMA; skewed: ~1040 — profiles/r03af_conv_k3_prio.log)
        if (decltype(khc)::value == 1 && g.prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (decltype(khc)::value == 1 && g.prio == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < FM; ++b)
            if (live(b, DH)) {
#pragma unroll
                for (int a = 0; a < FN; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[a]), __builtin_bit_cast(bf16x8, bfr[b]),
                                                                        acc[a][b], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // One interval = one K step, closed by ONE workgroup barrier (conv_k2.hip: waves 0-3 LOAD(s) then COMP(s); waves 4-7 COMP(s - 1) then
    // LOAD(s)).  Before the barrier every wave retires the pieces it issued before this interval (DEPTH 2) / before the previous interval
    // (DEPTH 3): pieces per tap are constants, so the wait is an immediate.
    auto interval = [&](auto khc, auto tapc, int chunk) {
        constexpr int KH = decltype(khc)::value, TAP = decltype(tapc)::value;
        constexpr int PREV = (TAP + 8) % 9, PREV2 = (TAP + 7) % 9;
        auto npc = [](int t) constexpr { return QI + ((!SINGLE && t >= 1 && t <= PI) ? 1 : 0); };
        constexpr int YOUNG = npc(TAP) + (DEPTH > 2 ? npc(PREV) : 0) + (DEPTH > 3 ? npc(PREV2) : 0);
        if (KH == 0) {
            load(tapc, chunk);
            comp(tapc, khc);
        } else {
            if (s > 0) comp(std::integral_constant<int, PREV>{}, khc);           // step s - 1
            load(tapc, chunk);
        }
        K3_VMWAIT(YOUNG);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ++s;
        if (++qs == NST) qs = 0;
    };
    auto run = [&](auto khc) {
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            interval(khc, std::integral_constant<int, 0>{}, chunk); interval(khc, std::integral_constant<int, 1>{}, chunk);
            interval(khc, std::integral_constant<int, 2>{}, chunk); interval(khc, std::integral_constant<int, 3>{}, chunk);
            interval(khc, std::integral_constant<int, 4>{}, chunk); interval(khc, std::integral_constant<int, 5>{}, chunk);
            interval(khc, std::integral_constant<int, 6>{}, chunk); interval(khc, std::integral_constant<int, 7>{}, chunk);
            interval(khc, std::integral_constant<int, 8>{}, chunk);
        }
        if (decltype(khc)::value == 1) comp(std::integral_constant<int, 8>{}, khc);          // the last step's MFMAs
    };
    if (kh == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the pieces streamed past the last step: LDS is reused below
    __syncthreads();
    K3_PHASE(2);

    // ---- the two K halves meet: a wave keeps the pixel fragments [kh*FM/2, (kh+1)*FM/2) and hands the others to its partner (conv_k2.hip)
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
        K3_PHASE(3);
        const int flags = g.flags;
        f32x4 bv[FN];
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
            bv[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((flags & K3_BIAS) && n < g.N) bv[a] = *(const f32x4*)(g.bias + n);
        }
        // Staged write-out (everything but the bf16 accumulate form): the tile goes through LDS as a bf16 [256 pixels][BN channels] image — local
        // pixel lp = column * H + h is global row m0 + lp — and leaves as 16-byte lane stores, four (eight) whole rows per wave instruction;
        // the ReLU mask of the layer below arrives the same way.  (The direct form below writes 8 bytes per lane, 16 pixel rows x 32 B per
        // instruction: 4.8 us of the 57 us of conv4_2, 8.9 with the mask loads in between — profiles/r03u_k3_phases.log.)
        if (!(flags & K3_ACCUM)) {
            constexpr int ROWB = BN * 2, U = ROWB / 16, SWM = BN == 128 ? 7 : 3;      // row bytes; 16-byte units per row; swizzle bits of the column
            __syncthreads();                            // the exchange region has been read
#pragma unroll
            for (int bb = 0; bb < FH; ++bb) {
                const int b = KH * FH + bb;
                const int lp = ((cb0 + b % CBW) * 16 + frow) * H + hbase + b / CBW;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    f32x4 v = acc[a][b] + bv[a];
                    if (flags & K3_RELU) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    u32x2 pk;
                    pk.x = pack_bf2(v.x, v.y);
                    pk.y = pack_bf2(v.z, v.w);
                    const int slot = wn * 16 + a * 4 + (lane >> 4);                      // 8-byte slot of the row; the column's low bits permute the 32-byte groups
                    *(u32x2*)(smem + lp * ROWB + ((slot ^ ((frow & SWM) << 2)) << 3)) = pk;
                }
            }
            __syncthreads();
            constexpr int NIT = 256 * U / 512;
            u32x4 val[NIT], mk[NIT], zt[BNB ? NIT : 1];
            float st_s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, st_q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float bmu[8], brs[8];
            if (BNB) {                                    // this thread's 8 channels are the same in every iteration (512 % U == 0)
#pragma unroll
                for (int c = 0; c < 8; ++c) { bmu[c] = g.bn_mean[n0 + (tid % U) * 8 + c]; brs[c] = g.bn_rstd[n0 + (tid % U) * 8 + c]; }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 512 + tid, lp = idx / U, u = idx % U;
                if (flags & K3_MASK) mk[it] = *(const u32x4*)(g.mask + ((long)m0 + lp) * g.N + n0 + u * 8);
                if (BNB) zt[it] = *(const u32x4*)(g.bnz + ((long)m0 + lp) * g.N + n0 + u * 8);
                val[it] = *(const u32x4*)(smem + lp * ROWB + ((u ^ (((lp / H) & SWM) << 1)) << 4));
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 512 + tid, lp = idx / U, u = idx % U;
                u32x4 v = val[it];
                if (flags & K3_MASK) {
                    const u32x4 q = mk[it];
                    if (!(bf_lo(q.x) > 0.f)) v.x &= 0xffff0000u;
                    if (!(bf_hi(q.x) > 0.f)) v.x &= 0x0000ffffu;
                    if (!(bf_lo(q.y) > 0.f)) v.y &= 0xffff0000u;
                    if (!(bf_hi(q.y) > 0.f)) v.y &= 0x0000ffffu;
                    if (!(bf_lo(q.z) > 0.f)) v.z &= 0xffff0000u;
                    if (!(bf_hi(q.z) > 0.f)) v.z &= 0x0000ffffu;
                    if (!(bf_lo(q.w) > 0.f)) v.w &= 0xffff0000u;
                    if (!(bf_hi(q.w) > 0.f)) v.w &= 0x0000ffffu;
                }
                *(u32x4*)(g.out + ((long)m0 + lp) * g.N + n0 + u * 8) = v;
                if (!BNB && g.pool_kind == 3) {          // statistics of what is stored: the bf16 values, as a separate pass over the tensor would see them
                    const float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
                    for (int c = 0; c < 8; ++c) { st_s[c] += f[c]; st_q[c] = fmaf(f[c], f[c], st_q[c]); }
                } else if (BNB) {                        // batch-norm backward sums of the masked gradient that is stored (bn_bwd_stats_kernel's terms)
                    const u32x4 z = zt[it];
                    const float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
                    const float zf[8] = {bf_lo(z.x), bf_hi(z.x), bf_lo(z.y), bf_hi(z.y), bf_lo(z.z), bf_hi(z.z), bf_lo(z.w), bf_hi(z.w)};
#pragma unroll
                    for (int c = 0; c < 8; ++c) { st_s[c] += f[c]; st_q[c] = fmaf(f[c], (zf[c] - bmu[c]) * brs[c], st_q[c]); }
                }
            }
            if (BNB || g.pool_kind == 3) {
                // Batch-norm statistics from the producing convolution (VERDICT r3 item 3 iv / 6): this tile's per-channel sum and sum of
                // squares over its 256 pixel rows -> partial row m0 / 256 of [M / 256][2][N] (fp32; bn_finalize_kernel adds the rows in
                // double, fixed order: deterministic).  A thread's 16-byte unit u = tid % U is the same in every iteration (512 % U == 0),
                // so it holds the sums of 8 channels over NIT rows; the 512 / U threads of a unit meet in LDS behind the staged image.
                constexpr int RL = 512 / U;
                float* red = (float*)(smem + 256 * ROWB);
                const int u = tid % U, rl = tid / U;
#pragma unroll
                for (int c = 0; c < 8; ++c) { red[rl * BN + u * 8 + c] = st_s[c]; red[(RL + rl) * BN + u * 8 + c] = st_q[c]; }
                __syncthreads();
                if (tid < 2 * BN) {
                    const int which = tid / BN, ch = tid % BN;
                    float t = 0.f;
                    for (int r = 0; r < RL; ++r) t += red[(which * RL + r) * BN + ch];
                    if (n0 + ch < g.N) ((float*)g.pool)[((long)(m0 / 256) * 2 + which) * g.N + n0 + ch] = t;
                }
            } else if (g.pool_kind) {
                // fused max-pool from the staged image (post-ReLU values; bf16 max is exact, so this equals max-pooling the stored tensor):
                // kind 1 pairs the feature rows (h, h + 1) of a column — local pixels 2q, 2q + 1 -> pooled row m0 / 2 + q; kind 2 the 2 x 2
                // window of columns (2c, 2c + 1) -> pooled row (col0 / 2) * (H / 2) + c * (H / 2) + h / 2
                const int kind = g.pool_kind;
                const int nq = kind == 1 ? 128 : 64;
                const long pbase = kind == 1 ? ((long)m0 >> 1) : (long)(col0 >> 1) * (H >> 1);
                for (int idx = tid; idx < nq * U; idx += 512) {
                    const int q = idx / U, u = idx % U;
                    int lp0, cl;
                    if (kind == 1) { lp0 = 2 * q; cl = lp0 / H; }
                    else { cl = 2 * (q / (H >> 1)); lp0 = cl * H + 2 * (q % (H >> 1)); }
                    const unsigned char* r0 = smem + lp0 * ROWB;
                    const int u0 = (u ^ ((cl & SWM) << 1)) << 4;
                    u32x4 m = k3_max8(*(const u32x4*)(r0 + u0), *(const u32x4*)(r0 + ROWB + u0));
                    if (kind == 2) {
                        const int u1 = (u ^ (((cl + 1) & SWM) << 1)) << 4;
                        m = k3_max8(m, k3_max8(*(const u32x4*)(r0 + H * ROWB + u1), *(const u32x4*)(r0 + (H + 1) * ROWB + u1)));
                    }
                    *(u32x4*)(g.pool + (pbase + q) * g.N + n0 + u * 8) = m;
                }
            }
            return;
        }
        // accumulate form (y += result: one fp32 sum, one rounding — a residual gradient with several consumers; never with a fused pool):
        // direct 8-byte stores from the accumulator layout
#pragma unroll
        for (int bb = 0; bb < FH; ++bb) {
            const int b = KH * FH + bb;
            const int h = hbase + b / CBW, col = col0 + (cb0 + b % CBW) * 16 + frow;       // this lane's pixel of fragment b
            const long m = (long)col * H + h;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
                if (n >= g.N) continue;
                f32x4 v = acc[a][b] + bv[a];
                if (flags & K3_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (flags & K3_MASK) {
                    u32x2 mk = *(const u32x2*)(g.mask + m * g.N + n);
                    if (!(bf_lo(mk.x) > 0.f)) v.x = 0.f;
                    if (!(bf_hi(mk.x) > 0.f)) v.y = 0.f;
                    if (!(bf_lo(mk.y) > 0.f)) v.z = 0.f;
                    if (!(bf_hi(mk.y) > 0.f)) v.w = 0.f;
                }
                if (flags & K3_ACCUM) {
                    const u32x2 old = *(const u32x2*)(g.out + m * g.N + n);
                    v.x += bf_lo(old.x); v.y += bf_hi(old.x); v.z += bf_lo(old.y); v.w += bf_hi(old.y);
                }
                u32x2 pk;
                pk.x = pack_bf2(v.x, v.y);
                pk.y = pack_bf2(v.z, v.w);
                *(u32x2*)(g.out + m * g.N + n) = pk;
            }
        }
    };
    if (kh) tail(std::integral_constant<int, 1>{}); else tail(std::integral_constant<int, 0>{});
    if (clk && blockIdx.x == 0 && tid == 0) ocr_clk_exit(clk);
#ifdef OCR_EXPERIMENTS
    K3_PHASE(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K3_PHASE(5);
#endif
}

// the kernels: aligned widths (the names the round's profiles carry) and the general-width form
template <int FM, int BN, int NST, int H, bool SINGLE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_k3_kernel(K3Args g) { k3_body<FM, BN, NST, H, SINGLE, false>(g); }
template <int FM, int BN, int NST, int H>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_k3w_kernel(K3Args g) { k3_body<FM, BN, NST, H, false, true>(g); }
// ... and with the batch-norm backward sums in the write-out (pool_kind 4)
template <int FM, int BN, int NST, int H, bool GENW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_k3b_kernel(K3Args g) { k3_body<FM, BN, NST, H, false, GENW, true>(g); }

template <int FM, int BN, int NST, int H, bool GENW>
static int launch_k3b(const K3Args& g, hipStream_t stream) {
    if (!g.P) return (GENW ? 6 : 4) + (BN == 128 ? 0 : 1);
    constexpr int PS = (256 / H + 2 + 7) / 8 * 8, PPIECES = (H * PS / 8 + 7) / 8 * 8;
    constexpr int need = NST * BN * 128 + 2 * PPIECES * 1024, xch = 8 * (FM / 2) * 4 * 1024, lds = need > xch ? need : xch;
    static_assert(lds <= 163840 && 256 * BN * 2 + 2 * (512 / (BN / 8)) * BN * 4 <= lds, "LDS");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_k3b_kernel<FM, BN, NST, H, GENW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC;
        attr = true;
    }
    conv_k3b_kernel<FM, BN, NST, H, GENW><<<(g.M / 256) * ((g.N + BN - 1) / BN), 512, lds, stream>>>(g);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

template <int FM, int BN, int NST, int H, bool SINGLE = false, bool GENW = false>
static int launch_k3(const K3Args& g, hipStream_t stream) {
    if (!g.P) return (GENW ? 6 : 4) + (BN == 128 ? 0 : 1);       // plan query (ocr_conv3x3_kernel_choice): nothing is launched
    constexpr int PS = (256 / H + 2 + 7) / 8 * 8, PPIECES = (H * PS / 8 + 7) / 8 * 8;
    constexpr int need = NST * BN * 128 + (SINGLE ? 1 : 2) * PPIECES * 1024;     // weight stages, padded halo buffer(s)
    constexpr int xch = 8 * (FM / 2) * 4 * 1024;                                  // the K-half exchange reuses them
    constexpr int lds = need > xch ? need : xch;
    static_assert(lds <= 163840, "LDS");
    static_assert(256 * BN * 2 + 2 * (512 / (BN / 8)) * BN * 4 <= lds, "staged image + the statistics scratch behind it");
    static bool attr = false;
    const int mt = g.M / 256, nt = (g.N + BN - 1) / BN;
    if constexpr (GENW) {
        if (!attr) {
            if (hipFuncSetAttribute((const void*)conv_k3w_kernel<FM, BN, NST, H>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC;
            attr = true;
        }
        conv_k3w_kernel<FM, BN, NST, H><<<mt * nt, 512, lds, stream>>>(g);
    } else {
        if (!attr) {
            if (hipFuncSetAttribute((const void*)conv_k3_kernel<FM, BN, NST, H, SINGLE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC;
            attr = true;
        }
        conv_k3_kernel<FM, BN, NST, H, SINGLE><<<mt * nt, 512, lds, stream>>>(g);
    }
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

// -1 = shape not covered.  tile: 'A' = 256 pixels x 128 channels, 'D' = 256 x 64 (chosen by conv_k2.hip's k2_choose)
int k3_try_dispatch(int tile, const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind) {
    if (H != 2 && H != 4 && H != 8 && H != 16) return -1;
    if (M % 256 || (Cin & 63) || (Cout & 63)) return -1;
    const int NC = 256 / H, PS = (NC + 2 + 7) / 8 * 8;
    const bool genw = W % NC != 0;                       // tiles cross image boundaries: the general-width form (one zero row per boundary)
    if (genw && (H == 16 || NC + 2 + (NC + W - 1) / W > PS)) return -1;       // (H = 16: 16-column tiles, W >= 16 there — not instantiated)
    if (H == 2 && (!genw || (pool_kind && pool_kind < 3))) return -1;       // H = 2 exists in the general-width form only (128-column tiles)
    if (pool_kind >= 3 && (flags & K3_ACCUM)) return -1; // statistics are taken in the staged write-out
    if (pool_kind == 4 && !(flags & K3_MASK)) return -1;  // batch-norm backward sums: of the ReLU-masked gradient
    static int genw_on = -1;                             // A/B knob OCR_K3_GENW = 0: general-width shapes stay on conv_k2 / conv_halo
    if (genw_on < 0) { const char* e = ocr_tune_env("OCR_K3_GENW"); genw_on = e ? atoi(e) : 1; }
    if (genw && !genw_on) return -1;
    static int prio = -1;
    if (prio < 0) { const char* e = ocr_tune_env("OCR_K3_PRIO"); prio = e ? atoi(e) : 2; }      // measured: 1 (equal) 349 us, 2 322, 3 324 over the ten layers (profiles/r03af)
    K3Args g = {(const bf16_t*)x, (const bf16_t*)wpack, M, Cout, Cin, W, H, (bf16_t*)y, bias, (const bf16_t*)mask, flags, (bf16_t*)pool, pool_kind, prio,
                nullptr, nullptr, nullptr};
    if (pool_kind == 4 && x) {                           // (plan queries pass no operands)
        const K3BnBwd* e = (const K3BnBwd*)pool;
        g.pool = (bf16_t*)e->partials; g.bnz = (const bf16_t*)e->z; g.bn_mean = e->mean; g.bn_rstd = e->rstd;
    }
    const bool single = Cin == 64;
    if (pool_kind == 4) {                                // instances exist for the shapes batch-norm layers have here: H in {4, 8}, several chunks
        if (single || (H != 4 && H != 8)) return -1;
        if (tile == 'A') {
            if (Cout % 128) return -1;
            if (genw) return H == 4 ? launch_k3b<8, 128, 5, 4, true>(g, stream) : launch_k3b<8, 128, 5, 8, true>(g, stream);
            return H == 4 ? launch_k3b<8, 128, 5, 4, false>(g, stream) : launch_k3b<8, 128, 5, 8, false>(g, stream);
        }
        if (genw) return H == 4 ? launch_k3b<4, 64, 4, 4, true>(g, stream) : launch_k3b<4, 64, 4, 8, true>(g, stream);
        return H == 4 ? launch_k3b<4, 64, 4, 4, false>(g, stream) : launch_k3b<4, 64, 4, 8, false>(g, stream);
    }
    if (genw) {
        if (tile == 'A') {
            if (Cout % 128) return -1;
            return H == 2 ? launch_k3<8, 128, 5, 2, false, true>(g, stream) : H == 4 ? launch_k3<8, 128, 5, 4, false, true>(g, stream)
                                                                                      : launch_k3<8, 128, 5, 8, false, true>(g, stream);
        }
        return H == 2 ? launch_k3<4, 64, 4, 2, false, true>(g, stream) : H == 4 ? launch_k3<4, 64, 4, 4, false, true>(g, stream)
                                                                                  : launch_k3<4, 64, 4, 8, false, true>(g, stream);
    }
    if (tile == 'A') {
        if (Cout % 128) return -1;
        if (H == 16) return single ? launch_k3<8, 128, 5, 16, true>(g, stream) : launch_k3<8, 128, 4, 16>(g, stream);     // 2 x 48 KiB of halo: four stages
        return H == 4 ? launch_k3<8, 128, 5, 4>(g, stream) : launch_k3<8, 128, 5, 8>(g, stream);
    }
    if (H == 16) return single ? launch_k3<4, 64, 4, 16, true>(g, stream) : launch_k3<4, 64, 4, 16>(g, stream);
    return H == 4 ? launch_k3<4, 64, 4, 4>(g, stream) : launch_k3<4, 64, 4, 8>(g, stream);
}
