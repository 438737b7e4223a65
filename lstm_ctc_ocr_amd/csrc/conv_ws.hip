// 3x3 SAME implicit-GEMM convolution for gfx950, seventh generation: WEIGHT-STATIONARY and PERSISTENT over the pixel tiles, for the short-K
// layers (Cin = 64 / 128: K = 576 / 1152) whose tiles are too short to pay for a launch of their own
// (same contract as ocr_conv3x3_bf16 — reference lib/networks/network.py:160-191 forward, and its data gradient; LSTM_train.py:26-28).
//
// Why (profiles/r04_final3_kernel_stats.md): conv2 forward ran 39.4 us for 19.3 GFLOP (0.20 of the MFMA peak) in every
// kernel form measured, conv2's data gradient 26.9 us (0.29), conv3_1 forward 22.2 us (0.35).  The plane-layout kernels (conv_k3.hip) stream
// a weight tile per K step through LDS and pay ~9 us per launch around the K loop (prologue 2.1, K-half exchange 1.0, staged stores 2.8-4.2,
// dispatch 2.8): with 9 or 18 K steps per tile that is as long as the loop itself, and the loop is LDS-bound (96 KB of fragment reads +
// 21 KB of DMA writes per 1040-clock step = 88 % of the LDS port: 1040 clocks against 853 of MFMA).
//
// Here a workgroup is FOUR waves, one per SIMD (512 registers each):
//   * every wave keeps the weights of its 64 output channels x ONE 64-channel input chunk x all nine taps in REGISTERS for the whole launch
//     (72 A fragments = 288 registers, loaded once from L2): no weight tile is ever written to or read from LDS — fragment reads per MFMA
//     drop from 0.375 to 0.25 and the DMA stream is the pixels only;
//   * the workgroup walks several pixel tiles of one channel tile (grid = CUs; conv2 forward: four 256-pixel tiles per workgroup): the halo
//     planes of tile t + 1 stream into the other LDS buffer while tile t multiplies, the write-out of tile t is wave-local (own staging
//     rows, no workgroup barrier) — ONE barrier per tile, no prologue and no drained pipeline between tiles;
//   * Cin = 128: the two input chunks are the K halves of two waves (K split inside the workgroup, 128-pixel tiles), which meet once per
//     tile through LDS — each wave finishes and writes out half of the pair's pixels.
// Halo layout: conv_k3's feature-row planes with TWO ZERO PLANES (h = -1 and h = H) that are written once and never touched by the DMA, so
// SAME padding along the feature axis needs neither address selects nor per-wave code (cost: 2 / (3 H) of the MFMAs multiply zeros), rows
// per plane = columns + 2 exactly (the XOR swizzle key is the plane row's low bits, carried by the DMA lane that fetches it).  A 16-pixel
// fragment is CF = min(16, columns) consecutive image columns x 16 / CF feature rows.
// Covered: (H, Cin) in {(16, 64), (16, 128), (8, 128)}, Cout % 64 == 0, W % columns-per-tile == 0, bias / ReLU / ReLU-mask write-outs and the
// fused 1 x 2 / 2 x 2 max-pool; everything else stays on conv_k3 / conv_k2 / conv_halo.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

enum { WS_BIAS = 1, WS_RELU = 2, WS_MASK = 16, WS_ACCUM = 64 };

struct WsArgs {
    const bf16_t* P; const bf16_t* Q;     // P [M pixels][C] ; Q [N][9 * C]  (k = tap * C + c, tap = 3 * dw + dh: the pack of conv_k3)
    int M, N, C, cW, cH;
    bf16_t* out; const float* bias; const bf16_t* mask; int flags;
    bf16_t* pool; int pool_kind;          // 0 none, 1 feature pairs (1 x 2), 2 2 x 2 (post-ReLU values of the stored tensor)
    int ntiles, slots, per_slot, xcd_map; // channel tiles; pixel-tile slots (workgroups per channel tile); pixel tiles per slot; XCD-aware id map
};

typedef __attribute__((address_space(3))) void* ws_lptr_t;
#define WS_OOB 0x80000000u                // lane offset of a row that does not exist: beyond any descriptor's range

// diagnostic (ocr_conv_ws_debug; tools/ws_phases.py): 100 MHz wall-clock stamps of every workgroup's first thread — dbg[block * 64 + {0 entry,
// 1 weights + first halo landed, then per tile i < 10: 2 + 6 i + {0 barrier passed, 1 next halo issued, 2 K loop issued, 3 pieces landed / old
// stores acknowledged, 4 K halves exchanged, 5 write-out issued}}]
__device__ long long* g_ws_dbg;
extern "C" int ocr_conv_ws_debug(void* dbg) {
    long long* q = (long long*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_dbg), &q, sizeof(q)) == hipSuccess ? OCR_OK : OCR_ERR_MEMOPS;
}
#define WS_PHASE(slot) do { if (wsdbg && tid == 0 && (slot) < 64) wsdbg[blockIdx.x * 64 + (slot)] = ocr_wall_clock(); } while (0)
__device__ long long* g_ws_clk;          // ocr_conv_halo_clock_debug: workgroup 0 stamps the clocks (common.h)
int ws_set_clock_debug(void* dbg) {
    long long* p = (long long*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_clk), &p, sizeof(p)) == hipSuccess ? OCR_OK : OCR_ERR_EXEC;
}

// packed bf16 max / ReLU as SIGNED 16-bit integer max (v_pk_max_i16, one instruction per pair): non-negative bf16 values order like their bit
// patterns, every negative value (and -0) is a negative integer — so max(x, 0) is the ReLU of a bf16 pair and, on post-ReLU values, the integer
// max is the float max (the single-wave write-out is VALU-issue-bound: the float-compare form of conv_k3 took ~10 instructions per pair here).
// NaN: a NaN with the sign bit set (0xFFxx) is a negative integer and becomes 0, a positive one (0x7Fxx) is kept — where conv_k3's float
// compares and tf.nn.relu keep every NaN.  A diverged run therefore shows its NaNs one layer later at the latest (conv2's data gradient, conv3_1 and
// the loss all run on conv_k3 / float arithmetic; the step report's loss is what the training loop checks), never as a silently finite loss.
typedef short ws_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ws_max2(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ws_s16x2, a), __builtin_bit_cast(ws_s16x2, b)));
}
__device__ __forceinline__ u32x4 ws_max8(u32x4 a, u32x4 b) {
    u32x4 r = {ws_max2(a.x, b.x), ws_max2(a.y, b.y), ws_max2(a.z, b.z), ws_max2(a.w, b.w)};
    return r;
}

// geometry of an instance — mirrored by tools/ws_plane_model.py (tests/test_ws_plane_model.py replays the index algebra on the CPU)
template <int H, int NC, int KSPLIT>
struct WsCfg {
    static constexpr int BM = NC * H;                 // pixels per tile (256 / KSPLIT)
    static constexpr int PG = 4 / KSPLIT;             // waves along the pixels
    static constexpr int CF = NC < 16 ? NC : 16;      // image columns per fragment
    static constexpr int HF = 16 / CF;                // feature rows per fragment
    static constexpr int HW = 4 * HF;                 // feature rows of a wave's four fragments (a wave owns all NC columns x HW rows)
    static constexpr int PS = NC + 2;                 // rows per plane
    static constexpr int CHB = (H + 2) * PS * 128;    // bytes of one chunk's planes (incl. the two zero planes)
    static constexpr int BUFB = KSPLIT * CHB;         // one halo buffer
    static constexpr int PPC = H * PS / 8;            // DMA pieces (8 rows x 128 B) per chunk
    static constexpr int PI = KSPLIT * PPC / 4;       // ... per wave and tile
    static constexpr int NFE = 4 / KSPLIT;            // fragments a wave writes out
    static constexpr int HWE = NFE * HF;              // ... = NC columns x HWE feature rows
    static constexpr int NPE = NC * HWE;              // pixels a wave writes out
    static constexpr int XOFF = 2 * BUFB;             // K-half exchange: 4 waves x 8 KB
    static constexpr int SOFF = XOFF + (KSPLIT == 2 ? 4 * 8192 : 0);
    static constexpr int STW = NPE * 128;             // staging bytes per wave
    static constexpr int TOFF = SOFF + 4 * STW;       // DMA lane table: [wave][piece][lane] source offsets as u16 (registers are for the weights)
    static constexpr int BOFF = TOFF + 4 * PI * 128;  // the channel tile's 64 biases
    static constexpr int MOFF = BOFF + 256;           // MASK instances: the ReLU-mask rows of the pixels a wave writes out (4 x STW)
    static constexpr int LDS_PLAIN = MOFF, LDS_MASK = MOFF + 4 * STW;
    static_assert(((NC + 1) * H + H) * 128 * 2 / 16 * 4 < 65536, "table entries fit 16 bits");
    static_assert(NC == CF && PG * HW == H && (H * PS) % 8 == 0 && (KSPLIT * PPC) % 4 == 0 && LDS_MASK <= 163840, "geometry");
    static_assert(((H + 1) * PS) * 128 < 65536, "ds_read immediates");
};

template <int H, int NC, int KSPLIT, bool MASK /* the ReLU mask of the layer below is applied in the write-out (data gradients) */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_ws_kernel(WsArgs g) {
    using G = WsCfg<H, NC, KSPLIT>;
    constexpr int CF = G::CF, HF = G::HF, PS = G::PS, PI = G::PI, HWE = G::HWE, NPE = G::NPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, lane_id = lane;
    long long* const clk = g_ws_clk;
    long long* const wsdbg = g_ws_dbg;
    if (clk && blockIdx.x == 0 && tid == 0) ocr_clk_enter(clk);
    WS_PHASE(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = KSPLIT == 1 ? wave : (wave & 1), kh = KSPLIT == 1 ? 0 : (wave >> 1);      // pixel group; input chunk (= K half)
    const int C = g.C, N = g.N;

    // ---- which channel tile, which run of pixel tiles
    int id = blockIdx.x, nt, slot;
    if (g.xcd_map) {                                    // ids with equal (id & 7) share an XCD's L2: the channel tiles of one pixel run sit together
        const int per_xcd = (int)gridDim.x >> 3, j = id >> 3;
        nt = j % g.ntiles; slot = (id & 7) * (per_xcd / g.ntiles) + j / g.ntiles;
    } else { nt = id % g.ntiles; slot = id / g.ntiles; }
    const int n0 = nt * 64;
    const int mtiles = g.M / G::BM;
    const int t_begin = slot * g.per_slot;
    int t_end = t_begin + g.per_slot; if (t_end > mtiles) t_end = mtiles;

    const int frow = lane & 15, fq = lane >> 4;
    const int rsub = lane >> 3;
    const unsigned lds0 = (unsigned)(size_t)(ws_lptr_t)smem;

    // ---- DMA: this wave's pieces are q = j * 4 + wave; lane -> (plane row, 16-byte position); the swizzle key is the plane row's low bits.
    // The per-lane source offsets live in LDS as (offset / 16) << 2 | flags (bit 0: left halo column, bit 1: right halo column — masked where
    // the tile touches an image edge).
    unsigned short* const dtab = (unsigned short*)(smem + G::TOFF) + wave * PI * 64 + lane;
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int q = j * 4 + wave;
        const int c = q / G::PPC, pc = q % G::PPC;
        const int r = pc * 8 + rsub;                    // row of the chunk's real planes
        const int h = r / PS, cp = r % PS;
        const unsigned off = (unsigned)((((cp * H + h) * C + c * 64) + (((lane & 7) ^ (cp & 7)) << 3)) * 2);       // a multiple of 16
        dtab[j * 64] = (unsigned short)((off >> 2) | (cp == 0 ? 1u : 0u) | (cp == NC + 1 ? 2u : 0u));
    }
    if (tid < 64) ((float*)(smem + G::BOFF))[tid] = (g.flags & WS_BIAS) ? g.bias[n0 + tid] : 0.f;
    const long pbytes = (long)g.M * C * 2;
    // issue_halo(t, buf, part): part < 0 all pieces now; else piece `part` only.  (Issuing the pieces of tile t + 1 one per K sub-step of
    // tile t, "under" its MFMAs, was measured and rejected: the K loop went from 2.9 to 4.1 us for nine pieces that cost 0.4 us at the top of
    // the tile — one wave per SIMD has no partner to issue vector-memory instructions beside its MFMAs: profiles/r05e_ws_phases.log)
    unsigned e_[PI];
    auto read_table = [&]() {
#pragma unroll
        for (int j = 0; j < PI; ++j) e_[j] = dtab[j * 64];
    };
    auto issue_halo = [&](int t, int buf, auto partc) {
        constexpr int PART = decltype(partc)::value;
        const int col0 = t * NC;                        // first column of the tile, counted over the whole batch
        const bool edge_l = col0 % g.cW == 0, edge_r = (col0 + NC) % g.cW == 0;
        const long base = (long)(col0 - 1) * H * C * 2; // (the left halo column of the very first tile lies in front of the tensor: masked)
        long left = pbytes - base; if (left > 0x7fffffffL) left = 0x7fffffffL;
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)g.P + base), 0, (int)left, 0x00020000);
        const unsigned em = (edge_l ? 1u : 0u) | (edge_r ? 2u : 0u);
#pragma unroll
        for (int j = (PART < 0 ? 0 : PART); j < (PART < 0 ? PI : PART + 1); ++j) {
            const int q = j * 4 + wave;
            const unsigned e = e_[j];
            const unsigned v = (e & em) ? WS_OOB : ((e & ~3u) << 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (ws_lptr_t)(smem + buf * G::BUFB + (q / G::PPC) * G::CHB + PS * 128 + (q % G::PPC) * 1024), 16, (int)v, 0, 0, 0);
        }
    };

    // ---- pixel fragments: lane register per (dw, k half j) — it points into the buffer of the running tile and flips by BUFB per tile —,
    //      immediate per (fragment, dh); the plane offset of the pixel group (a run-time wave index) is part of the register
    unsigned pb[3][2];
    {
        const int colf = frow % CF, hsub = frow / CF;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cp = colf + d;
                pb[d][j] = lds0 + kh * G::CHB + ((pg * 4 * HF + hsub) * PS + cp) * 128 + (((j * 4 + fq) ^ (cp & 7)) << 4);
            }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's table rows
    read_table();
    if (t_begin < t_end) issue_halo(t_begin, 0, std::integral_constant<int, -1>{});        // lands in buffer 0 while the weights arrive

    // ---- this wave's weights: 64 output channels x chunk kh x nine taps = 72 A fragments (lane: channel row frow, k group fq), kept in
    // registers for the whole launch.  Loaded ONCE per workgroup and chunk: the NSH waves that share a chunk fetch 72 / NSH fragments each
    // from global memory and hand them round through LDS (buffer 1's space, R rounds of 36 KB).  (Every wave fetching all 72 itself — the
    // same 147 KB from every CU of the chip, 64-byte row segments — took 9.5 us of a 33 us launch: profiles/r05c_ws_phases.log.)
    u32x4 wq[4][18];
    {
        constexpr int NSH = 4 / KSPLIT, R = KSPLIT == 1 ? 2 : 4, FR = 72 / R, OWN = 72 / NSH;
        static_assert(KSPLIT * FR * 1024 <= G::BUFB && FR % NSH == 0, "a round fits buffer 1");
        const int rank = KSPLIT == 1 ? wave : pg;
        const bf16_t* q = g.Q + ((long)(n0 + frow) * 9 * C + kh * 64 + fq * 8);
        u32x4 own[OWN];
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            const int f = i * NSH + rank, fa = f / 18, fs = f % 18;             // fragment f = a * 18 + s
            own[i] = *(const u32x4*)(q + (long)fa * 16 * 9 * C + (fs >> 1) * C + (fs & 1) * 32);
        }
        u32x4* const wx = (u32x4*)(smem + G::BUFB) + kh * FR * 64 + lane;
#pragma unroll
        for (int rd = 0; rd < R; ++rd) {
#pragma unroll
            for (int i = rd * FR / NSH; i < (rd + 1) * FR / NSH; ++i) wx[(i * NSH + rank - rd * FR) * 64] = own[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int f = rd * FR; f < (rd + 1) * FR; ++f) wq[f / 18][f % 18] = wx[(f - rd * FR) * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // ---- the two zero planes of every chunk image of both buffers (never written again; after the weights, which used buffer 1's space)
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        constexpr int ZU = PS * 8;                      // 16-byte units per plane
        for (int i = tid; i < 2 * KSPLIT * 2 * ZU; i += 256) {
            const int img = i / (2 * ZU), r = i % (2 * ZU);
            const int off = img * G::CHB + (r < ZU ? 0 : (H + 1) * PS * 128) + (r % ZU) * 16;
            *(u32x4*)(smem + off) = z;
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the first halo
    WS_PHASE(1);

    constexpr bool has_mask = MASK;
    constexpr int NIT = NPE / 8;                        // 16-byte row units per lane in the write-out

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        // halo(t) has landed (every wave waited for its own pieces after its K loop); buffer buf ^ 1 and the exchange block are free.  A RAW
        // barrier: __syncthreads() would also drain the write-out stores of tile t - 1 (its release fence waits for vmcnt(0)) in front of every tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int ph0 = 2 + 6 * (t - t_begin);
        WS_PHASE(ph0);
        if (t + 1 < t_end) { read_table(); issue_halo(t + 1, buf ^ 1, std::integral_constant<int, -1>{}); }
        const int col0 = t * NC;
        // rows this wave writes out: NC columns x HWE feature rows from h_lo
        const int h_lo = pg * G::HW + (KSPLIT == 2 ? kh * HWE : 0);
        if (MASK) {                                     // the mask rows of the pixels this wave writes out: straight into LDS, in write-out order
            const long mbase = ((long)col0 * H + h_lo) * N + n0;
            long left = (long)g.M * N - mbase; if (left > 0x3fffffffL) left = 0x3fffffffL;
            const __amdgpu_buffer_rsrc_t msrd = __builtin_amdgcn_make_buffer_rsrc((void*)(g.mask + mbase), 0, (int)(left * 2), 0x00020000);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * 8 + rsub;
                const unsigned v = (unsigned)((((row / HWE) * H + row % HWE) * N + (lane & 7) * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(msrd, (ws_lptr_t)(smem + G::MOFF + wave * G::STW + it * 1024), 16, (int)v, 0, 0, 0);
            }
        }
        WS_PHASE(ph0 + 1);
        // ---- K loop: 18 sub-steps (tap, k half), the fragments of sub-step s + 1 are read while s multiplies; the accumulators start from the
        //      bias (the first MFMA's C operand; the second K half of a split starts from zero)
        f32x4 binit[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            binit[a] = *(const f32x4*)(smem + G::BOFF + (a * 16 + fq * 4) * 4);
            if (KSPLIT == 2 && kh) binit[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(binit[0]), "+v"(binit[1]), "+v"(binit[2]), "+v"(binit[3]));      // (not inside the counted loop)
        f32x4 acc[4][4];
        u32x4 bfr[2][4];
#define WS_RD(S_, DST_) do { \
            constexpr int TAP_ = (S_) >> 1, J_ = (S_) & 1, DW_ = TAP_ / 3, DH_ = TAP_ % 3 - 1; \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_[0]) : "v"(pb[DW_][J_]), "n"(((0 * HF + 1 + DH_) * PS) * 128)); \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_[1]) : "v"(pb[DW_][J_]), "n"(((1 * HF + 1 + DH_) * PS) * 128)); \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_[2]) : "v"(pb[DW_][J_]), "n"(((2 * HF + 1 + DH_) * PS) * 128)); \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_[3]) : "v"(pb[DW_][J_]), "n"(((3 * HF + 1 + DH_) * PS) * 128)); \
        } while (0)
#define WS_MM(S_, SRC_, FIRST_) do { \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[a][S_]), __builtin_bit_cast(bf16x8, SRC_[b]), \
                                                                    (FIRST_) ? binit[a] : acc[a][b], 0, 0, 0); \
        } while (0)
#define WS_WAIT(N_, SRC_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(SRC_[0]), "+v"(SRC_[1]), "+v"(SRC_[2]), "+v"(SRC_[3]))
#define WS_STEP(S_) do { \
            WS_RD((S_) + 1, bfr[((S_) + 1) & 1]); \
            WS_WAIT(4, bfr[(S_) & 1]);              /* the four reads of sub-step S_ have returned (LDS returns in order) */ \
            WS_MM(S_, bfr[(S_) & 1], (S_) == 0); \
            __builtin_amdgcn_sched_barrier(0); \
        } while (0)
        WS_RD(0, bfr[0]);
        WS_STEP(0); WS_STEP(1); WS_STEP(2); WS_STEP(3); WS_STEP(4); WS_STEP(5); WS_STEP(6); WS_STEP(7); WS_STEP(8);
        WS_STEP(9); WS_STEP(10); WS_STEP(11); WS_STEP(12); WS_STEP(13); WS_STEP(14); WS_STEP(15); WS_STEP(16);
        WS_WAIT(0, bfr[1]);
        WS_MM(17, bfr[1], false);
        __builtin_amdgcn_sched_barrier(0);
#undef WS_STEP
#undef WS_WAIT
#undef WS_MM
#undef WS_RD
        WS_PHASE(ph0 + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of halo(t + 1) (issued a K loop ago), the mask rows, the stores of tile t - 1
        WS_PHASE(ph0 + 3);

        // ---- write-out of this wave's NC x HWE pixels (KH selects the fragments a K-split wave keeps)
        auto tail = [&](auto khc) {
            constexpr int KH = decltype(khc)::value;
            constexpr int B0 = KSPLIT == 2 ? 2 * KH : 0;          // first fragment this wave finishes
            if (KSPLIT == 2) {
                f32x4* mine = (f32x4*)(smem + G::XOFF) + (size_t)wave * 512;
                const f32x4* theirs = (const f32x4*)(smem + G::XOFF) + (size_t)(wave ^ 2) * 512;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int i = 0; i < 2; ++i) mine[(a * 2 + i) * 64 + lane_id] = acc[a][2 * (1 - KH) + i];      // what the partner finishes
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[a][B0 + i] += theirs[(a * 2 + i) * 64 + lane_id];
            }
            WS_PHASE(ph0 + 4);
            // (the write-out's lane arithmetic is re-derived per tile from an opaque copy of the lane id: hoisted out of the tile loop its ~20
            //  loop-invariant address registers do not fit beside the weights and came back as scratch reloads inside the loop)
            int lane = lane_id;
            asm volatile("" : "+v"(lane));
            const int frow = lane & 15, fq = lane >> 4;
            unsigned char* const S = smem + G::SOFF + wave * G::STW;
            const int colf = frow % CF, hsub = frow / CF;
            const bool relu = (g.flags & WS_RELU) != 0;
#pragma unroll
            for (int i = 0; i < G::NFE; ++i) {
                const int row = colf * HWE + i * HF + hsub;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 v = acc[a][B0 + i];
                    u32x2 pk;
                    pk.x = pack_bf2(v.x, v.y);
                    pk.y = pack_bf2(v.z, v.w);
                    if (relu) { pk.x = ws_max2(pk.x, 0u); pk.y = ws_max2(pk.y, 0u); }      // ReLU after the rounding: the same values
                    const int slot = a * 4 + fq;                    // 8-byte slot of the 128-byte row; the column's low bits permute the 32-byte groups
                    *(u32x2*)(S + row * 128 + ((slot ^ ((colf & 3) << 2)) << 3)) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its own LDS writes are in order; nothing to wait for but the compiler's view)
            u32x4 val[NIT], mkv[MASK ? NIT : 1];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 64 + lane, row = idx >> 3, u = idx & 7;
                val[it] = *(const u32x4*)(S + row * 128 + ((u ^ (((row / HWE) & 3) << 1)) << 4));
                if (has_mask) mkv[it] = *(const u32x4*)(smem + G::MOFF + wave * G::STW + idx * 16);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 64 + lane, row = idx >> 3, u = idx & 7;
                u32x4 v = val[it];
                if (has_mask) {
                    const u32x4 q = mkv[it];
                    if (!(bf_lo(q.x) > 0.f)) v.x &= 0xffff0000u;
                    if (!(bf_hi(q.x) > 0.f)) v.x &= 0x0000ffffu;
                    if (!(bf_lo(q.y) > 0.f)) v.y &= 0xffff0000u;
                    if (!(bf_hi(q.y) > 0.f)) v.y &= 0x0000ffffu;
                    if (!(bf_lo(q.z) > 0.f)) v.z &= 0xffff0000u;
                    if (!(bf_hi(q.z) > 0.f)) v.z &= 0x0000ffffu;
                    if (!(bf_lo(q.w) > 0.f)) v.w &= 0xffff0000u;
                    if (!(bf_hi(q.w) > 0.f)) v.w &= 0x0000ffffu;
                }
                const long m = (long)(col0 + row / HWE) * H + h_lo + row % HWE;
                *(u32x4*)(g.out + m * N + n0 + u * 8) = v;
            }
            if (g.pool_kind == 1) {             // feature pairs (h, h + 1) of a column -> pooled row m / 2
                constexpr int NQ = NC * (HWE / 2), PIT = NQ * 8 / 64;
                static_assert(NQ * 8 % 64 == 0, "whole wave iterations");
                u32x4 p0[PIT], p1[PIT];
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int idx = it * 64 + lane, q = idx >> 3, u = idx & 7, col = q / (HWE / 2), ph = q % (HWE / 2);
                    const unsigned char* r0 = S + (col * HWE + 2 * ph) * 128 + ((u ^ ((col & 3) << 1)) << 4);
                    p0[it] = *(const u32x4*)r0; p1[it] = *(const u32x4*)(r0 + 128);
                }
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int idx = it * 64 + lane, q = idx >> 3, u = idx & 7, col = q / (HWE / 2), ph = q % (HWE / 2);
                    const long pm = (long)(col0 + col) * (H / 2) + (h_lo >> 1) + ph;
                    *(u32x4*)(g.pool + pm * N + n0 + u * 8) = ws_max8(p0[it], p1[it]);
                }
            } else if (g.pool_kind == 2) {      // 2 x 2 window of columns (2c, 2c + 1) x rows (2p, 2p + 1)
                constexpr int NQ = (NC / 2) * (HWE / 2), PIT = NQ * 8 / 64;
                static_assert(NQ * 8 % 64 == 0, "whole wave iterations");
                u32x4 p0[PIT], p1[PIT], p2[PIT], p3[PIT];
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int idx = it * 64 + lane, q = idx >> 3, u = idx & 7, pc = q / (HWE / 2), ph = q % (HWE / 2);
                    const int c0 = 2 * pc, c1 = c0 + 1;
                    const unsigned char* r0 = S + (c0 * HWE + 2 * ph) * 128 + ((u ^ ((c0 & 3) << 1)) << 4);
                    const unsigned char* r1 = S + (c1 * HWE + 2 * ph) * 128 + ((u ^ ((c1 & 3) << 1)) << 4);
                    p0[it] = *(const u32x4*)r0; p1[it] = *(const u32x4*)(r0 + 128); p2[it] = *(const u32x4*)r1; p3[it] = *(const u32x4*)(r1 + 128);
                }
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int idx = it * 64 + lane, q = idx >> 3, u = idx & 7, pc = q / (HWE / 2), ph = q % (HWE / 2);
                    const long pm = (long)((col0 >> 1) + pc) * (H / 2) + (h_lo >> 1) + ph;
                    *(u32x4*)(g.pool + pm * N + n0 + u * 8) = ws_max8(ws_max8(p0[it], p1[it]), ws_max8(p2[it], p3[it]));
                }
            }
        };
        if (KSPLIT == 2 && kh) tail(std::integral_constant<int, 1>{}); else tail(std::integral_constant<int, 0>{});
        WS_PHASE(ph0 + 5);
        const unsigned flip = buf ? (unsigned)-G::BUFB : (unsigned)G::BUFB;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j < 2; ++j) pb[d][j] += flip;
    }
    if (clk && blockIdx.x == 0 && tid == 0) ocr_clk_exit(clk);
}

template <int H, int NC, int KSPLIT, bool MASK>
static int launch_ws_(WsArgs& g, int grid, hipStream_t stream) {
    using G = WsCfg<H, NC, KSPLIT>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_ws_kernel<H, NC, KSPLIT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, MASK ? G::LDS_MASK : G::LDS_PLAIN) != hipSuccess) return OCR_ERR_EXEC;
        attr = true;
    }
    conv_ws_kernel<H, NC, KSPLIT, MASK><<<grid, 256, MASK ? G::LDS_MASK : G::LDS_PLAIN, stream>>>(g);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
template <int H, int NC, int KSPLIT>
static int launch_ws(WsArgs& g, int cus, hipStream_t stream) {
    using G = WsCfg<H, NC, KSPLIT>;
    if (!g.P) return 8;                                  // plan query (ocr_conv3x3_kernel_choice): nothing is launched
    const int mtiles = g.M / G::BM;
    g.ntiles = g.N / 64;
    int slots = cus / g.ntiles;
    if (slots < 1) slots = 1;
    if (slots > mtiles) slots = mtiles;
    g.per_slot = (mtiles + slots - 1) / slots;
    slots = (mtiles + g.per_slot - 1) / g.per_slot;      // no idle workgroups behind the last tile
    g.slots = slots;
    const int grid = slots * g.ntiles;
    g.xcd_map = (grid % 8 == 0 && (grid / 8) % g.ntiles == 0) ? 1 : 0;
    return (g.flags & WS_MASK) ? launch_ws_<H, NC, KSPLIT, true>(g, grid, stream) : launch_ws_<H, NC, KSPLIT, false>(g, grid, stream);
}

// -1 = shape not covered / not chosen (the caller goes on to conv_k3 / conv_k2 / conv_halo).
// OCR_CONV_WS: 0 never, 1 (default) where a workgroup gets at least two tiles, 2 every covered shape (parity tests at small sizes)
int ws_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream, void* pool, int pool_kind) {
    static int mode = -1, cus = 0;
    if (mode < 0) { const char* e = getenv("OCR_CONV_WS"); mode = e ? atoi(e) : 1; }
    if (!mode) return -1;
    if (flags & ~(WS_BIAS | WS_RELU | WS_MASK)) return -1;
    if (pool_kind < 0 || pool_kind > 2 || (pool_kind && (!(flags & WS_RELU) || (flags & WS_MASK)))) return -1;
    if ((Cout & 63) || (long)M * Cin * 2 > 0x7fffffffL || (long)M * Cout * 2 > 0x7fffffffL) return -1;
    int nc, bm;
    if (H == 16 && Cin == 64) { nc = 16; bm = 256; }
    else if (H == 16 && Cin == 128) { nc = 8; bm = 128; }
    else if (H == 8 && Cin == 128) { nc = 16; bm = 128; }
    else return -1;
    if (W % nc || M % bm) return -1;
    if (pool_kind == 2 && (W & 1)) return -1;
    // CU count of the device: latched only once the query SUCCEEDED (the host-only plan queries of the lowering — x == nullptr, possibly
    // no device at all — came first and fixed the 256 fallback for the rest of the process); until then every call asks again
    if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    const int ncu = cus > 0 ? cus : 256;                 // no device visible (CPU-only policy tests): the MI355X's count
    // default policy (measured, profiles/r05e_ws_bench.log): the single-chunk instance (Cin = 64: conv2 forward 40.8 -> 32.3 us) from two tiles
    // per workgroup on; the K-split instances (Cin = 128) are correct but lose to conv_k3 (conv2 data gradient 35 against 28 us, conv3_1 forward
    // 32 against 21: their per-workgroup weight hand-round costs 7-8 us of a 4-tile run) — OCR_CONV_WS=2 runs them for the parity tests
    if (mode == 1 && (Cin != 64 || (long)(M / bm) * (Cout / 64) < 2L * ncu)) return -1;
    WsArgs g = {(const bf16_t*)x, (const bf16_t*)wpack, M, Cout, Cin, W, H, (bf16_t*)y, bias, (const bf16_t*)mask, flags, (bf16_t*)pool, pool_kind, 0, 0, 0, 0};
    if (H == 16 && Cin == 64) return launch_ws<16, 16, 1>(g, ncu, stream);
    if (H == 16) return launch_ws<16, 8, 2>(g, ncu, stream);
    return launch_ws<8, 16, 2>(g, ncu, stream);
}
