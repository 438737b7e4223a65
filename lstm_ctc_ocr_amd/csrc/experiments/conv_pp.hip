// 3x3 SAME implicit-GEMM convolution, PING-PONG schedule (fourth generation; same contract as ocr_conv3x3_bf16 — reference
// lib/networks/network.py:160-191 forward, and its data gradient).
//
// conv_halo.hip keeps all eight waves of a workgroup in lock step: barrier -> DMA issue -> 16 fragment reads -> 32 MFMAs.
// Timing ablations on MI355X (tools/halo_variants.py, round 2): the five forward layers take 225 us; without the fragment
// reads 178, without the DMA 183, with neither 144 — the matrix pipe idles while both waves of a SIMD issue DMA pieces
// (~100-150 cycles each) and wait for LDS, and nothing else overlaps them because both are in the same phase.
// Here the two waves of a SIMD (w and w + 4 of the workgroup) run half a step apart, separated by workgroup barriers:
//     segment 2s   : waves 0-3  LOAD(s)   = DMA issue for step s+2 + the 16 fragment reads of step s into registers
//                    waves 4-7  COMP(s-1) = 32 MFMAs on the fragments they loaded in segment 2s-1
//     segment 2s+1 : waves 0-3  COMP(s),  waves 4-7  LOAD(s)
// so every SIMD always has one wave on the matrix pipe and one on the memory pipes (the regime MI355X_MICROARCH.md describes
// for two waves per SIMD).  Same tiles as conv_halo (256 pixels x BN channels, 64 x 64 or 32 x 64 per wave,
// v_mfma_f32_16x16x32_bf16, halo tile per 64-channel chunk shared by the nine taps, XOR source swizzle, SAME padding by
// reading a zero row), but THREE weight stages: the tile of step s is read in segments 2s (waves 0-3) and 2s+1 (waves 4-7) and
// the tile of step s+2 is streamed in during exactly those two segments into the buffer that step s-1 left in segment 2s-1.
// Halo pieces of the next chunk are issued one per step (taps 0 .. PI-1), never as a block.
#include "../common.h"
#include <stdlib.h>

enum { PPF_BIAS = 1, PPF_RELU = 2, PPF_MASK = 16 };

struct PpArgs {
    const bf16_t* P; const bf16_t* Q;     // P [M pixels][C] ; Q [N][9*C]
    int M, N, C;                          // C % 64 == 0
    int cW, cH;
    bf16_t* out; const float* bias; const bf16_t* mask; int flags;
};

__device__ u32x4 pp_zero_page[4];
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PP_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__device__ long long* pp_dbg;                      // diagnostic: s_memtime stamps of workgroup 0 (ocr_conv_pp_debug)

template <int BN, bool DBG = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_pp_kernel(PpArgs g) {
    constexpr int NW = 8, BM = 256;
    constexpr int WAVES_N = BN / 64, WAVES_M = NW / WAVES_N;
    constexpr int WM = BM / WAVES_M, FM = WM / 16, FN = 4;
    constexpr int QB = BN * 128, QI = BN / (8 * NW);
    constexpr int NRpad = 320, PI = NRpad / (8 * NW);          // halo rows incl. the zero rows; DMA pieces per wave
    constexpr int PBYTES = NRpad * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pbuf0 = smem;                       // 2 halo stages
    unsigned char* qbuf0 = smem + 2 * PBYTES;          // 3 weight stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;                        // 0: loads in even segments, 1: loads in odd segments
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int H = g.cH, C = g.C;
    const int NR = BM + 2 * H + 2;                     // halo rows actually needed (< NRpad: the rest are the zero rows)

    const int mtiles = (g.M + BM - 1) / BM, ntiles = (g.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int Lb = blockIdx.x;
    if ((nblk & 7) == 0) Lb = (Lb & 7) * (nblk >> 3) + (Lb >> 3);
    const int m0 = (Lb / ntiles) * BM, n0 = (Lb % ntiles) * BN;

    const int rsub = lane >> 3;
    const int csrc = ((lane & 7) ^ rsub) * 8;          // LDS position lane&7 of row r holds source chunk (lane&7)^(r&7)
    const long mfirst = (long)m0 - H - 1;              // flat pixel of halo row 0

    // DMA geometry, all loop invariant: per piece a 32-bit byte offset per lane; the step only moves a scalar base.  Rows that do
    // not exist (n >= N, pixels outside [0, M), the padding rows of the halo stage) are clamped to a real row: what lands in LDS
    // there is never multiplied (outputs n >= N are not stored; SAME-padding taps read the dedicated zero row, see below).
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned voffQ[QI], voffP[PI];
#pragma unroll
    for (int j = 0; j < QI; ++j) {
        const int n = min(n0 + (wave * QI + j) * 8 + rsub, g.N - 1);
        voffQ[j] = (unsigned)(((long)n * 9 * C + csrc) * 2);
    }
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        long m = mfirst + (wave * PI + j) * 8 + rsub;
        m = m < 0 ? 0 : (m >= g.M ? g.M - 1 : m);
        voffP[j] = (unsigned)((m * C + csrc) * 2);
    }
    auto load_q = [&](int k0, int buf) {               // k0 = tap*C + chunk*64
#pragma unroll
        for (int j = 0; j < QI; ++j) dma16_saddr(lds0 + 2 * PBYTES + buf * QB + (wave * QI + j) * 1024, voffQ[j], g.Q + k0);
    };
    auto load_p1 = [&](int chunk, int buf, int j) {    // piece j (0 .. PI-1) of this wave (j is a constant or wave-uniform)
        dma16_saddr(lds0 + buf * PBYTES + (wave * PI + j) * 1024, voffP[j], g.P + chunk * 64);
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
    const unsigned qfrag0 = (unsigned)(qbuf0 - smem) + (wn * 64 + frow) * 128 + ((fq ^ fx) << 4);   // + stage*QB, ^64 for kk = 1
    const unsigned prow0 = (wm * WM + frow) * 128;                                                  // + shift*128 + swizzle

    constexpr unsigned ZROW = 2 * PBYTES + 3 * QB;       // 128 zero bytes behind the stages (written once, visible after barrier 0)
    if (tid < 8) *(u32x4*)(smem + ZROW + tid * 16) = (u32x4){0, 0, 0, 0};
    // ---- prologue: weight tiles of steps 0 = (tap 0, chunk 0) and 1 = (tap 1, chunk 0), the whole halo of chunk 0; everything
    //      has landed before barrier 0
    load_q(0, 0);
    load_q(C, 1);
#pragma unroll
    for (int j = 0; j < PI; ++j) load_p1(0, 0, j);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (role == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // segment 0: waves 4-7 have nothing to multiply yet

    int tap = 0, chunk = 0, qs = 0;                    // qs = s % 3: weight stage of this step
    for (int s = 0; s < nsteps; ++s) {
        long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (DBG) st[0] = __builtin_amdgcn_s_memtime();
        // ================================================================ LOAD(s)
        {   // weight tile of step s+2 into the stage that step s-1 left one segment (waves 0-3) / two segments (waves 4-7) ago
            int t2 = tap + 2, c2 = chunk;
            if (t2 >= 9) { t2 -= 9; ++c2; }
            int q2 = qs + 2; if (q2 >= 3) q2 -= 3;
            if (s + 2 < nsteps) load_q(t2 * C + c2 * 64, q2);
            if (tap < PI && chunk + 1 < nchunks) {                                               // one halo piece per step
                switch (tap) {                  // static register index
                    case 0: load_p1(chunk + 1, (chunk + 1) & 1, 0); break; case 1: load_p1(chunk + 1, (chunk + 1) & 1, 1); break;
                    case 2: load_p1(chunk + 1, (chunk + 1) & 1, 2); break; case 3: load_p1(chunk + 1, (chunk + 1) & 1, 3); break;
                    default: load_p1(chunk + 1, (chunk + 1) & 1, 4); break;
                }
            }
        }
        if (DBG) st[1] = __builtin_amdgcn_s_memtime();
        const int shift = (H + 1) + (tap / 3 - 1) * H + (tap % 3 - 1);       // halo row of local pixel 0 for this tap
        const unsigned qa = qfrag0 + qs * QB;
        const unsigned pbase = (chunk & 1) * PBYTES + prow0 + shift * 128 + ((fq ^ ((frow + shift) & 7)) << 4);
        unsigned pa[FM];
        const unsigned zoff = ZROW;                  // the dedicated zero row
#pragma unroll
        for (int b = 0; b < FM; ++b) pa[b] = ((vmask[b] >> tap) & 1u) ? pbase + b * 2048 : zoff;
        u32x4 afr[2][FN], bfr[2][FM];
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
        // Pieces of this wave younger than the tile of step s+1: the halo piece of LOAD(s-1), the tile of step s+2, the halo
        // piece of LOAD(s).  Everyone's pieces of tile s+1 must have landed before barrier 2s+2: waves 4-7 are in front of it
        // now (end of their LOAD), waves 0-3 at the end of COMP(s).
        const int hprev = (tap >= 1 && tap - 1 < PI && chunk + 1 < nchunks) ? 1 : 0, hcur = (tap < PI && chunk + 1 < nchunks) ? 1 : 0;
        const int younger = (s + 2 < nsteps ? QI : 0) + hprev + hcur;
        auto vmwait = [&]() {
            switch (younger) {
                case 0: PP_VMWAIT(0); break; case 1: PP_VMWAIT(1); break; case 2: PP_VMWAIT(2); break;
                case 3: PP_VMWAIT(3); break; default: PP_VMWAIT(4); break;
            }
        };
        if (DBG) st[2] = __builtin_amdgcn_s_memtime();
        if (role == 1) vmwait();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (DBG) st[3] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (DBG) st[4] = __builtin_amdgcn_s_memtime();
        // ================================================================ COMP(s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[kk][a]),
                                                                        __builtin_bit_cast(bf16x8, bfr[kk][b]), acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (DBG) st[5] = __builtin_amdgcn_s_memtime();
        if (role == 0) vmwait();
        if (DBG) st[6] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (DBG && blockIdx.x == 0 && lane == 0 && pp_dbg != nullptr && s < 80) {
            st[7] = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 8; ++i) pp_dbg[(wave * 80 + s) * 8 + i] = st[i];
        }
        if (++tap == 9) { tap = 0; ++chunk; }
        if (++qs == 3) qs = 0;
    }
    if (role == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // matches the extra barrier waves 4-7 took at the start

    // ---- epilogue (m-block outer, n-fragment inner: the stores completing a 128-B run of a pixel row are adjacent)
    const int flags = g.flags;
    f32x4 bv[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
        bv[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((flags & PPF_BIAS) && n < g.N) bv[a] = *(const f32x4*)(g.bias + n);
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * WM + b * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            f32x4 v = acc[a][b] + bv[a];
            if (flags & PPF_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (flags & PPF_MASK) {
                u32x2 mk = *(const u32x2*)(g.mask + (long)m * g.N + n);
                if (!(bf_lo(mk.x) > 0.f)) v.x = 0.f;
                if (!(bf_hi(mk.x) > 0.f)) v.y = 0.f;
                if (!(bf_lo(mk.y) > 0.f)) v.z = 0.f;
                if (!(bf_hi(mk.y) > 0.f)) v.w = 0.f;
            }
            u32x2 pk;
            pk.x = pack_bf2(v.x, v.y);
            pk.y = pack_bf2(v.z, v.w);
            *(u32x2*)(g.out + (long)m * g.N + n) = pk;
        }
    }
}

static bool g_pp_dbg = false;
extern "C" int ocr_conv_pp_debug(void* dbg /* device int64[8 waves][80 steps][8] or NULL */) {
    long long* q = (long long*)dbg;
    g_pp_dbg = q != nullptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(pp_dbg), &q, sizeof(q)) == hipSuccess ? OCR_OK : OCR_ERR_MEMOPS;
}

template <int BN, bool DBG>
static int launch_pp_(const PpArgs& g, hipStream_t stream) {
    const int lds = 2 * 320 * 128 + 3 * BN * 128 + 128;          // halo stages, weight stages, zero row
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_pp_kernel<BN, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC;
        attr = true;
    }
    const int mt = (g.M + 255) / 256, nt = (g.N + BN - 1) / BN;
    conv_pp_kernel<BN, DBG><<<mt * nt, 512, lds, stream>>>(g);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
template <int BN>
static int launch_pp(const PpArgs& g, hipStream_t stream) { return g_pp_dbg ? launch_pp_<BN, true>(g, stream) : launch_pp_<BN, false>(g, stream); }

// -1 = shape not covered (caller falls back to conv_halo.hip / igemm.hip / gemm.hip)
int pp_try_dispatch(const void* x, const void* wpack, void* y, int M, int W, int H, int Cin, int Cout, const float* bias,
                    const void* mask, int flags, hipStream_t stream) {
    if ((Cin & 63) || (Cout & 3) || M < 1024 || H > 30) return -1;
    if (flags & ~(PPF_BIAS | PPF_RELU | PPF_MASK)) return -1;
    if (256 + 2 * H + 2 >= 320) return -1;                       // at least one spare (zero) row in the 320-row halo stage
    PpArgs g = {(const bf16_t*)x, (const bf16_t*)wpack, M, Cout, Cin, W, H, (bf16_t*)y, bias, (const bf16_t*)mask, flags};
    const int mt = (M + 255) / 256;
    if (Cout >= 128 && (long)mt * ((Cout + 127) / 128) >= 200) return launch_pp<128>(g, stream);
    if (Cout >= 128 && (long)mt * ((Cout + 63) / 64) < 256) return launch_pp<128>(g, stream);
    return launch_pp<64>(g, stream);
}
