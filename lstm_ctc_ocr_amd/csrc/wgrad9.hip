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
constexpr int W9_LDS = 4 * 9 * 4 * 4 * 64 * 4;    // 147 456 B: the epilogue's half-sum exchange (> 3 stages)

#define W9_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define W9_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

__global__ __launch_bounds__(512) void wgrad9_kernel(W9Args g) {
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
    const int hs = H == 4 ? 2 : (H == 8 ? 3 : 4);                                // log2 H
    const int ncol = 32 >> hs;                                                   // image columns per 32-pixel block
    int wc = (kbeg >> hs) % W;                                                   // column (within its image) of the step's first pixel
    const int lc0 = (4 * g4) >> hs, lc1 = (4 * g4 + 16) >> hs;                   // this lane's column inside a 32-pixel block (both reads)

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
    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W9_NDMA) : "memory");     // the next step may stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + 2 < nsteps) { int nb = cur + 2; if (nb >= W9_NST) nb -= W9_NST; stage_load(step + 2, nb); }
        const unsigned sb = lds0 + cur * W9_STAGE;

        s16x4 alo[4], ahi[4], blo[2][4], bhi[2][4];
        // prologue of the read stream: B of block 0, A of taps 0 and 1
#pragma unroll
        for (int c = 0; c < 4; ++c) { W9_TR(blo[0][c], sb + offB[c], 0); W9_TR(bhi[0][c], sb + offB[c], 16 * 128); }
#pragma unroll
        for (int t = 0; t < 2; ++t) { W9_TR(alo[t], sb + offA[t], 0); W9_TR(ahi[t], sb + offA[t], 16 * 128); }
        bool bnd = false;                    // does the current 32-pixel block touch image column 0 or W - 1 ? (wave-uniform)
        int w0 = 0, w1 = 0;                  // this lane's image column for the two reads (valid when bnd)
#pragma unroll
        for (int n = 0; n < 18; ++n) {
            const int kk = n / 9, t = n % 9;
            if (t == 0) {
                int cbase = wc + ((kh * 64 + kk * 32) >> hs);
                while (cbase >= W) cbase -= W;
                bnd = cbase == 0 || cbase + ncol >= W;
                w0 = cbase + lc0; w1 = cbase + lc1;           // < 2 W (the plan requires W >= 32 / H)
                if (w0 >= W) w0 -= W;
                if (w1 >= W) w1 -= W;
            }
            // keep the stream two taps ahead; across the block boundary: next block's B at tap 7, its taps 0 / 1 at tap 8
            if (t <= 6) { W9_TR(alo[(n + 2) & 3], sb + offA[t + 2], kk * 32 * 128); W9_TR(ahi[(n + 2) & 3], sb + offA[t + 2], kk * 32 * 128 + 16 * 128); }
            if (kk == 0 && t == 7) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { W9_TR(blo[1][c], sb + offB[c], 32 * 128); W9_TR(bhi[1][c], sb + offB[c], 32 * 128 + 16 * 128); }
            }
            if (kk == 0 && t == 8) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { W9_TR(alo[(n + 1 + u) & 3], sb + offA[u], 32 * 128); W9_TR(ahi[(n + 1 + u) & 3], sb + offA[u], 32 * 128 + 16 * 128); }
            }
            // reads younger than tap t's that may stay in flight (LDS returns in order)
            if (t <= 6) W9_WAIT(4);
            else if (t == 7) { if (kk == 0) W9_WAIT(10); else W9_WAIT(2); }
            else { if (kk == 0) W9_WAIT(12); else W9_WAIT(0); }
            __builtin_amdgcn_sched_barrier(0);
            u32x2 lo = __builtin_bit_cast(u32x2, alo[n & 3]), hi = __builtin_bit_cast(u32x2, ahi[n & 3]);
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
        if (do_cs) {            // bias gradient: column sums of the dY tile (rows past kend are zero-filled)
            u32x4 v0, v1;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(sb + offC));
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(v1) : "v"(sb + offC));
            W9_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
            cs[0] += bf_lo(v0.x) + bf_lo(v1.x); cs[1] += bf_hi(v0.x) + bf_hi(v1.x); cs[2] += bf_lo(v0.y) + bf_lo(v1.y); cs[3] += bf_hi(v0.y) + bf_hi(v1.y);
            cs[4] += bf_lo(v0.z) + bf_lo(v1.z); cs[5] += bf_hi(v0.z) + bf_hi(v1.z); cs[6] += bf_lo(v0.w) + bf_lo(v1.w); cs[7] += bf_hi(v0.w) + bf_hi(v1.w);
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
    // sum the two pixel halves: waves kh = 1 hand their accumulators over through LDS ([wave][tap][c][r][lane], conflict free)
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

// dw[i] += sum_s part[s][i] (fixed order: deterministic);  dbias[co] += sum_s cs_part[s][co]
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(float* __restrict__ dw, const float* __restrict__ part, long n4, long slab4,
                                                            int S, float* __restrict__ dbias, const float* __restrict__ cs_part, int Cout) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 a = ((const f32x4*)dw)[i];
        for (int s = 0; s < S; ++s) a += ((const f32x4*)part)[s * slab4 + i];
        ((f32x4*)dw)[i] = a;
    }
    if (dbias != nullptr && blockIdx.x == 0)
        for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
            float a = dbias[c];
            for (int s = 0; s < S; ++s) a += cs_part[(long)s * Cout + c];
            dbias[c] = a;
        }
}

struct W9Plan { int S, k_per_split, map, T_ci, T_co; size_t bytes; };

static bool w9_plan(int M, int W, int H, int Cin, int Cout, W9Plan* p) {
    if ((Cin & 63) || (Cout & 63) || (H != 4 && H != 8 && H != 16) || M < 512 || (M % H) || W * H < 32) return false;
    const int T_ci = Cin / 64, T_co = Cout / 64, T = T_ci * T_co;
    int S = 1;
    while ((long)S * 2 * T <= 256 && M / (S * 2) >= 512) S *= 2;           // one workgroup per CU, at least four steps each
    p->S = S; p->T_ci = T_ci; p->T_co = T_co;
    p->k_per_split = ceil_div(ceil_div(M, S), 128) * 128;
    p->map = S >= 8 ? 1 : ((T % (8 / S)) == 0 ? 2 : 0);
    p->bytes = (size_t)S * ((size_t)9 * Cin * Cout + Cout) * sizeof(float);
    return true;
}

extern "C" int ocr_conv3x3_wgrad_workspace_size(int Nb, int W, int H, int Cin, int Cout, size_t* bytes) {
    if (!bytes || Nb <= 0 || W <= 0 || H <= 0 || Cin <= 0 || Cout <= 0) return OCR_ERR_INVALID;
    W9Plan p;
    *bytes = w9_plan(Nb * W * H, W, H, Cin, Cout, &p) ? p.bytes : 0;
    return OCR_OK;
}

// -1: shape not covered or workspace too small (caller falls back to the atomics kernels)
int wgrad9_try_dispatch(const void* x, const void* dy, float* dw, float* dbias, int Nb, int W, int H, int Cin, int Cout,
                        void* workspace, size_t ws_bytes, hipStream_t stream) {
    W9Plan p;
    const int M = Nb * W * H;
    if (!workspace || !w9_plan(M, W, H, Cin, Cout, &p) || ws_bytes < p.bytes) return -1;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad9_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W9_LDS) != hipSuccess) return OCR_ERR_EXEC;
        attr = true;
    }
    W9Args g = {};
    g.X = (const bf16_t*)x; g.dY = (const bf16_t*)dy; g.M = M; g.Cin = Cin; g.Cout = Cout; g.cW = W; g.cH = H;
    g.k_per_split = p.k_per_split; g.S = p.S; g.T_ci = p.T_ci; g.T_co = p.T_co; g.map = p.map;
    g.part = (float*)workspace;
    g.cs_part = dbias ? g.part + (size_t)p.S * 9 * Cin * Cout : nullptr;
    wgrad9_kernel<<<p.S * p.T_ci * p.T_co, 512, W9_LDS, stream>>>(g);
    OCR_CHECK_LAUNCH();
    const long n4 = (long)9 * Cin * Cout / 4;
    int blocks = (int)((n4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    wgrad9_reduce_kernel<<<blocks, 256, 0, stream>>>(dw, g.part, n4, n4, p.S, dbias, g.cs_part, Cout);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
