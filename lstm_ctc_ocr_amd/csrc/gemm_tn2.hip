// Second-generation weight-gradient kernel for gfx950 (same contract as gemm_tn.hip):
//     out[i][j] += scale * sum_m A[m][i] * B[m][j]        (contraction over the slow index of both operands)
// for the 3x3 SAME convolution weight gradient (A = activations shifted by the tap, B = dY — reference
// tf.gradients of network.py:166 at train.py:81) and the plain X^T dY products of conv5 / LSTM / FC.
//
// Changes against gemm_tn.hip (measured 250-495 TFLOP/s, register-staged 128x128x32 steps, 2 barriers per step):
//   * BK = 64 rows of m per step and LDS-DMA (`global_load_lds_dwordx4`) into two stages: half the barriers per flop,
//     no staging VGPRs / ds_write pass, the next step streams in under the MFMAs;
//   * the tile stays row-major [m][channel] (what DMA can write); the transposing read ds_read_b64_tr_b16 still
//     delivers 4 consecutive m of one channel per lane.  Bank conflicts are removed by XOR-ing the 32-byte slot index
//     of a row with (row & 7) — applied to the SOURCE chunk of the DMA and again in the read address — together with
//     the k-permutation "lane group g owns tile rows 4g..4g+3 and 16+4g..16+4g+3" of every 32-row MFMA step (identical
//     for A and B, so the contraction is unchanged): each 32-lane half of a read then touches 8 rows x 32 B = 64 banks;
//   * halo pixels / tails come from a zero page; bias-gradient column sums are taken from the LDS tile by the blocks
//     that see every B row once (first i-tile, centre tap).
// Requires I % 128 == 0 and J % 128 == 0 (256-byte tile rows); other shapes stay on gemm_tn.hip.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

struct Tn2Args {
    const bf16_t* A; const bf16_t* B;
    long lda, ldb;
    int Mk, I, J, k_per_split;     // k_per_split % 64 == 0
    int grp, skip; long a_row_off;
    int cW, cH, cC;
    float* out; long ldo; float scale; float* colsum;
    // workgroup -> (split, tap, i tile, j tile) map.  The dispatcher deals workgroups round-robin over the 8 XCDs (private
    // L2 each), so XCD x = id & 7 is given the splits s = xs (mod XS), i tiles = xi (mod XI), j tiles = xj (mod XJ) with
    // XS * XJ * XI = 8: an XCD then streams only 1/(XI*XS) of A and 1/(XJ*XS) of B through its L2 (every XCD used to
    // stream both operands completely: TCC hit rate 2-66 %, 180-245 MB fetched per launch against 33-50 MB of operands).
    int XS, XJ, XI, taps, itl, jtl;                // itl = i tiles per XCD, jtl = j tiles per XCD
    int nbatch; long sA, sB, sO, sC;               // batched plain mode: problem b uses A + b*sA, B + b*sB, out + b*sO, colsum + b*sC
    int owned;                                     // plain mode, one split: every output tile belongs to ONE workgroup -> `out +=` without atomics
    __device__ int cH_or1() const { return cH > 0 ? cH : 1; }
};

__device__ u32x4 tn2_zero_page[4];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// byte offset (inside a swizzled [64][128] tile, k-step 0) of the first transposing read of channel block cb:
// tile rows {4g..4g+3} (+16 for the second read, +32 per k-step), channel cb*16 + lane&15
__device__ __forceinline__ int tn2_frag_off(int cb, int lane) {
    const int g = lane >> 4, L = lane & 15;
    const int r0 = 4 * g + (L >> 2);                     // (r0 + 16) & 7 and (r0 + 32) & 7 equal r0 & 7
    const int q = cb * 2 + ((L & 3) >> 1);
    const int p = q ^ ((r0 & 7) << 1);
    return r0 * 256 + p * 16 + (L & 1) * 8;
}
__device__ __forceinline__ bf16x8 tn2_frag(const unsigned char* a0) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 256));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int MODE /*0 plain, 1 conv3x3, 2 conv3x3 with Cin == 64: taps (2p, 2p+1) share one 128-channel A tile*/,
          int BKR /* rows of m per pipeline stage: 64 or 32 */, int NST /* LDS stages: 2..4 */,
          int NWV = 4 /* waves: 4 (64 x 64 wave tiles) or 8 (64 x 32 wave tiles: two waves per SIMD from ONE workgroup per CU, i.e. without
                         doubling the split count and with it the atomic traffic) */>
__global__ __launch_bounds__(64 * NWV) void gemm_tn2_kernel(Tn2Args g) {
    constexpr int TILE = BKR * 256;                  // bytes of one operand tile: BKR rows x 128 channels bf16
    constexpr int NJ = BKR / (4 * NWV);              // DMA instructions per wave, operand and stage
    constexpr int FB = NWV == 4 ? 4 : 2;             // 16-column fragments of the wave tile along j
    constexpr int NKK = BKR / 32;                    // 32-row MFMA K steps per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NST stages x (A tile | B tile)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = NWV == 4 ? wave >> 1 : wave >> 2, wj = NWV == 4 ? (wave & 1) : (wave & 3);
    const int xcd = blockIdx.x & 7;
    int q = blockIdx.x >> 3;
    const int xs = xcd % g.XS, xj = (xcd / g.XS) % g.XJ, xi = xcd / (g.XS * g.XJ);
    const int it_l = q % g.itl; q /= g.itl;
    const int tapi = q % g.taps; q /= g.taps;
    const int jt_l = q % g.jtl; q /= g.jtl;
    const int bi = q % g.nbatch; q /= g.nbatch;        // q is now the XCD-local split index
    g.A += bi * g.sA; g.B += bi * g.sB; g.out += bi * g.sO;
    if (g.colsum) g.colsum += bi * g.sC;
    const int tap = (MODE == 1) ? tapi : (MODE == 2 ? 2 * tapi : 0);
    const int i0 = (MODE == 2) ? 0 : (it_l * g.XI + xi) * 128;
    const int j0 = (jt_l * g.XJ + xj) * 128;
    const int kbeg = (q * g.XS + xs) * g.k_per_split;
    const int kend = min(g.Mk, kbeg + g.k_per_split);
    if (kbeg >= kend) return;                          // rounding k_per_split up to 64 can leave trailing splits empty
    const int dw = tap / 3 - 1, dh = tap % 3 - 1;
    const bool do_cs = g.colsum != nullptr && i0 == 0 && (MODE == 0 || tap == 4);     // (MODE 2: the pair (4, 5) block)
    const bf16_t* zero = (const bf16_t*)tn2_zero_page;

    // DMA geometry: instruction j of wave w covers tile rows (w*4 + j)*4 .. +3, lane -> (row lane>>4, 16-B position lane&15).
    // Everything that does not change along the K loop is hoisted: per (lane, j) the row, the swizzled source chunk, the
    // operand base pointers; the pixel coordinates (w, h) of the row are advanced incrementally (+64 rows per step)
    // instead of being re-derived with two integer divisions per DMA instruction (that version spent 8.5 VALU
    // instructions per MFMA and was VALU-bound).
    const int rsub = lane >> 4, pos = lane & 15;
    int trow[NJ];
    const bf16_t* pa[NJ];
    const bf16_t* pb[NJ];
    int pw[NJ], ph[NJ];
    int ldw[NJ], ldh[NJ];
    bool ltap_ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        ldw[j] = 0; ldh[j] = 0; ltap_ok[j] = true;
        const int r = (wave * NJ + j) * 4 + rsub;
        const int q = pos ^ ((r & 7) << 1);                 // source chunk held at LDS position `pos`
        trow[j] = r;
        const long m = (long)kbeg + r;
        pb[j] = g.B + m * g.ldb + j0 + q * 8;
        if (MODE == 0) {
            pa[j] = g.A + g.a_row_off + i0 + q * 8;          // row term added per step (row-group skip is not affine)
            pw[j] = ph[j] = 0;
        } else if (MODE == 1) {
            pa[j] = g.A + (m + (long)dw * g.cH + dh) * g.cC + i0 + q * 8;
            ph[j] = (int)(m % g.cH);
            pw[j] = (int)((m / g.cH) % g.cW);
        } else {                                             // this lane's chunk belongs to tap `tap` (q < 8) or `tap + 1`
            const int tl = tap + (q >> 3);
            ldw[j] = tl / 3 - 1; ldh[j] = tl % 3 - 1; ltap_ok[j] = tl < 9;
            pa[j] = g.A + (m + (long)ldw[j] * g.cH + ldh[j]) * g.cC + (q & 7) * 8;
            ph[j] = (int)(m % g.cH);
            pw[j] = (int)((m / g.cH) % g.cW);
        }
    }
    const int step_h = BKR % g.cH_or1(), step_w = BKR / g.cH_or1();
    auto stage_load = [&](int k0, int buf) {
        unsigned char* sa = smem + buf * (2 * TILE);
        unsigned char* sb = sa + TILE;
        const long koff = (long)(k0 - kbeg);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = k0 + trow[j];
            const bf16_t* srca = zero;
            const bf16_t* srcb = zero;
            if (m < kend) {
                srcb = pb[j] + koff * g.ldb;
                if (MODE == 0) {
                    long phys = (long)m + (g.grp > 0 ? (long)(m / g.grp) * g.skip : 0);
                    srca = pa[j] + phys * g.lda;
                } else if (MODE == 1) {
                    if ((unsigned)(pw[j] + dw) < (unsigned)g.cW && (unsigned)(ph[j] + dh) < (unsigned)g.cH) srca = pa[j] + koff * g.cC;
                } else if (ltap_ok[j] && (unsigned)(pw[j] + ldw[j]) < (unsigned)g.cW && (unsigned)(ph[j] + ldh[j]) < (unsigned)g.cH) {
                    srca = pa[j] + koff * g.cC;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)srca, (lptr_t)(sa + (wave * NJ + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)srcb, (lptr_t)(sb + (wave * NJ + j) * 1024), 16, 0, 0);
            if (MODE != 0) {                                  // advance this row's pixel by BKR rows for the next step
                int h = ph[j] + step_h, w = pw[j] + step_w;
                if (h >= g.cH) { h -= g.cH; ++w; }
                while (w >= g.cW) w -= g.cW;
                ph[j] = h; pw[j] = w;
            }
        }
    };

    f32x4 acc[4][FB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    int offa[4], offb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { offa[a] = tn2_frag_off(wi * 4 + a, lane); offb[a] = tn2_frag_off(wj * FB + (a < FB ? a : 0), lane); }

    {
        // NST stages, NST-1 steps in flight.  The counted wait at the top of step t retires stage t only (the 2*NJ DMA
        // instructions of every younger stage stay outstanding); the raw s_barrier behind it makes that true for every wave's
        // pieces and also orders "everyone finished reading stage t-1" (each wave ends a step with lgkmcnt(0)) before stage
        // t+NST-1 is streamed into that buffer.  (__syncthreads() would drain the DMA queue: an LDS-DMA is a pending LDS
        // write on the VM counter.)
        // The transposing reads of a step are inline asm: hipcc puts an s_waitcnt vmcnt(0) between an LDS-DMA and a later
        // builtin ds_read_tr — it cannot prove that the DMA writes another stage — which serialised every step into "DMA round
        // trip, then compute" (MFMA 12-19 % busy).  Asm reads are neither counted nor fenced by the compiler; the waits are
        // placed by hand.  With two K-halves per stage 32 reads are outstanding and lgkmcnt holds at most 15, so lgkmcnt(15)
        // (>= 17 retired, LDS returns in order) covers the first half.
        const int nst = (kend - kbeg + BKR - 1) / BKR;
#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
            if (p < nst) stage_load(kbeg + p * BKR, p);
        int cur = 0;
        for (int t = 0; t < nst; ++t) {
            const int younger = min(NST - 2, nst - 1 - t);           // stages that may stay in flight
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NJ) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + NST - 1 < nst) { int nb = cur + NST - 1; if (nb >= NST) nb -= NST; stage_load(kbeg + (t + NST - 1) * BKR, nb); }
            const unsigned char* tb = smem + cur * (2 * TILE) + TILE;
            const unsigned sbase = lds0 + cur * (2 * TILE);
            s16x4 flo[NKK][8], fhi[NKK][8];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
                for (int f = 0; f < 4 + FB; ++f) {
                    const unsigned ad = sbase + (f < 4 ? offa[f] : TILE + offb[f - 4]);
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(flo[kk][f]) : "v"(ad), "n"(kk * 32 * 256));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fhi[kk][f]) : "v"(ad), "n"(kk * 32 * 256 + 16 * 256));
                }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                if (kk + 1 < NKK) { if (FB == 4) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); }
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 fr[8];
#pragma unroll
                for (int f = 0; f < 4 + FB; ++f) {
                    s16x8 v = {flo[kk][f][0], flo[kk][f][1], flo[kk][f][2], flo[kk][f][3], fhi[kk][f][0], fhi[kk][f][1], fhi[kk][f][2], fhi[kk][f][3]};
                    fr[f] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < FB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[a], fr[4 + b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (do_cs && tid < 256) {      // thread -> source chunk q = tid & 15 (8 columns), rows (tid >> 4) + 16 * i
                const int q = tid & 15;
#pragma unroll
                for (int i = 0; i < BKR / 16; ++i) {
                    const int r = (tid >> 4) + 16 * i;
                    u32x4 v = *(const u32x4*)(tb + r * 256 + ((q ^ ((r & 7) << 1)) << 4));
                    cs[0] += bf_lo(v.x); cs[1] += bf_hi(v.x); cs[2] += bf_lo(v.y); cs[3] += bf_hi(v.y);
                    cs[4] += bf_lo(v.z); cs[5] += bf_hi(v.z); cs[6] += bf_lo(v.w); cs[7] += bf_hi(v.w);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            cur = (cur + 1 == NST) ? 0 : cur + 1;
        }
        __syncthreads();                                         // tiles are dead: the colsum reduction reuses the LDS
    }

    if (do_cs) {
        float* red = (float*)smem;                     // all tiles are dead after the last barrier
#pragma unroll
        for (int e = 0; e < 8; ++e) if (tid < 256) red[tid * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
            for (int u = tid >> 3; u < 256; u += 16) t += red[u * 8 + (tid & 7)];
            atomicAdd(g.colsum + j0 + tid, t * g.scale);
        }
    }
    // lane owns i = ib + (lane>>4)*4 + r, j = jb + (lane&15)
    float* out = g.out;
    long ldo = g.ldo;
    if (MODE == 1) { out += (long)tap * g.cC * g.J; ldo = g.J; }
    if (MODE == 2) {                                  // rows 0..63 -> tap `tap`, rows 64..127 -> tap + 1 (wave row wi selects)
        if (tap + wi >= 9) return;
        out += (long)(tap + wi) * 64 * g.J - (long)wi * 64 * g.J; ldo = g.J;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int j = j0 + wj * (16 * FB) + b * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 64 + a * 16 + (lane >> 4) * 4 + r;
                if (g.owned) out[(long)i * ldo + j] += acc[a][b][r] * g.scale;       // one workgroup per output tile (no split, plain mode): no atomics
                else atomicAdd(out + (long)i * ldo + j, acc[a][b][r] * g.scale);
            }
        }
}

// returns -1 if the shape is not covered
int tn2_try_dispatch(const void* A, long lda, const void* B, long ldb, float* out, long ldo, int Mk, int I, int J, int mode,
                     int grp, int skip, long a_row_off, int cW, int cH, int cC, float scale, int splits, float* colsum,
                     hipStream_t stream, int nbatch = 1, long sA = 0, long sB = 0, long sO = 0, long sC = 0) {
    const bool pair = (mode == 1 && I == 64);          // Cin == 64: two taps per A tile
    if ((!pair && (I & 127)) || (J & 127) || Mk < 256) return -1;
    Tn2Args g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.lda = lda; g.ldb = ldb; g.Mk = Mk; g.I = I; g.J = J;
    g.grp = grp; g.skip = skip; g.a_row_off = a_row_off; g.cW = cW; g.cH = cH; g.cC = cC;
    g.out = out; g.ldo = ldo; g.scale = scale; g.colsum = colsum;
    const int taps = pair ? 5 : (mode == 1 ? 9 : 1);
    const int IT = pair ? 1 : I / 128, JT = J / 128;
    const long tiles = (long)taps * IT * JT * nbatch;
    g.nbatch = nbatch; g.sA = sA; g.sB = sB; g.sO = sO; g.sC = sC;
    int maxs = Mk / 768; if (maxs < 1) maxs = 1;      // at least 12 K steps per workgroup: shorter runs are all prologue + atomics
                                                      // (Mk = 4032: 4-6 splits measured best, tools/tn_plain_probe.py)
    // Workgroup count and XCD partition (sweep: tools/wgrad_part_sweep.py).  The K loop is latency-bound (two LDS stages, one
    // 32 KiB tile in flight per workgroup), so the fastest grids keep both workgroup slots of every CU filled: 400-512
    // workgroups.  Among the partitions (XS, XJ, XI) that reach that, take the least operand re-streaming plus atomic traffic
    // bytes(A) * XJ + bytes(B) * XI + 2 * S * bytes(out)   (fp32 atomics cost ~0.8 us per MB, tools/atomic_probe.py).
    // plain products run on 8-wave workgroups, ONE per CU (two waves per SIMD without doubling the split count and its atomic
    // traffic): conv5 30.3 -> 27.1 us, both BiLSTM directions 35.7 -> 32.9 us (tools/tn_split_sweep.py).  A/B knob OCR_TN2_NW=4.
    static int nw8 = -1;
    if (nw8 < 0) { const char* e = ocr_tune_env("OCR_TN2_NW"); nw8 = (e && atoi(e) == 4) ? 0 : 1; }
    const bool wide = nw8 && mode == 0;
    const long wg_lo = wide ? 200 : 400, wg_hi = wide ? 256 : 512;
    const double ideal = (wide ? 232.0 : 432.0) / (double)tiles;
    const double bytesA = 2.0 * Mk * (pair ? 64 : I) * nbatch, bytesB = 2.0 * Mk * J * nbatch, bytesO = 4.0 * taps * (pair ? 128 : I) * J * nbatch;
    double best = 1e300; int bXS = 1, bXJ = 1, bXI = 1, bS = 1;
    for (int XS = 8; XS >= 1; XS >>= 1)
        for (int XJ = 8 / XS; XJ >= 1; XJ >>= 1) {
            const int XI = 8 / XS / XJ;
            if (IT % XI || JT % XJ) continue;
            int S;
            if (splits > 0) { if (splits % XS) continue; S = splits; }      // an explicit split count is honoured exactly
            else {
                S = (int)(ideal / XS + 0.5) * XS;
                if (S < XS) S = XS;
                while (S > XS && ((long)S * tiles > wg_hi || S > maxs)) S -= XS;
                if (S > maxs && XS > 1) continue;
            }
            const long wg = (long)S * tiles;
            double cost = bytesA * XJ + bytesB * XI + 2.0 * S * bytesO;
            if (splits <= 0 && wg < wg_lo) cost += (wg_lo - wg) * 40.0 * Mk;   // worth of a filled slot grows with the K loop
            if (splits <= 0 && wg > wg_hi) cost += (wg - wg_hi) * 4.0e6;       // a third, ragged round of workgroups
            if (cost < best) { best = cost; bXS = XS; bXJ = XJ; bXI = XI; bS = S; }
        }
    static const char* part_env = ocr_tune_env("OCR_TN2_PART");    // experiment knob "XS,XJ,XI,S", read once per process
    if (const char* e = part_env) {
        int xs, xj, xi, sp;
        if (sscanf(e, "%d,%d,%d,%d", &xs, &xj, &xi, &sp) == 4 && xs * xj * xi == 8 && IT % xi == 0 && JT % xj == 0 && sp % xs == 0 && sp >= 1) {
            bXS = xs; bXJ = xj; bXI = xi; bS = sp; best = 0;
        }
    }
    if (best == 1e300) return -1;
    splits = bS;
    g.XS = bXS; g.XJ = bXJ; g.XI = bXI; g.taps = taps; g.itl = IT / bXI; g.jtl = JT / bXJ;
    g.k_per_split = ceil_div(ceil_div(Mk, splits), 64) * 64;
    g.owned = (mode == 0 && splits == 1) ? 1 : 0;
    static int cfg = -1;                           // A/B knob OCR_TN2_PIPE: 0 = 64-row stages x 2 (default), 1 = 32 x 4, 2 = 32 x 3, 3 = 64 x 3
    if (cfg < 0) { const char* e = ocr_tune_env("OCR_TN2_PIPE"); cfg = e ? atoi(e) : 0; if (cfg < 0 || cfg > 4) cfg = 0; }      // 4 = 64 x 4 (8-wave form only)
    const int km = pair ? 2 : mode;
    dim3 grid((unsigned)(tiles * splits));          // empty splits (kbeg >= Mk) return at once
#define TN2_LAUNCH(M_, BK_, NS_) do { \
        static bool attr = false; const int lds = NS_ * 2 * BK_ * 256; \
        if (!attr) { if (hipFuncSetAttribute((const void*)gemm_tn2_kernel<M_, BK_, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC; attr = true; } \
        gemm_tn2_kernel<M_, BK_, NS_><<<grid, 256, lds, stream>>>(g); } while (0)
#define TN2_MODE(BK_, NS_) do { if (km == 2) TN2_LAUNCH(2, BK_, NS_); else if (km == 1) TN2_LAUNCH(1, BK_, NS_); else TN2_LAUNCH(0, BK_, NS_); } while (0)
    if (wide) {                                    // 8 waves, 64-row stages x 2 (or x 3 with OCR_TN2_PIPE=3)
        static bool attr8[3] = {false, false, false};
        const int ns = cfg == 4 ? 4 : (cfg == 3 ? 3 : 2), lds = ns * 2 * 64 * 256;
        if (ns == 4) {
            if (!attr8[2]) { if (hipFuncSetAttribute((const void*)gemm_tn2_kernel<0, 64, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC; attr8[2] = true; }
            gemm_tn2_kernel<0, 64, 4, 8><<<grid, 512, lds, stream>>>(g);
        } else if (ns == 3) {
            if (!attr8[1]) { if (hipFuncSetAttribute((const void*)gemm_tn2_kernel<0, 64, 3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC; attr8[1] = true; }
            gemm_tn2_kernel<0, 64, 3, 8><<<grid, 512, lds, stream>>>(g);
        } else {
            if (!attr8[0]) { if (hipFuncSetAttribute((const void*)gemm_tn2_kernel<0, 64, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC; attr8[0] = true; }
            gemm_tn2_kernel<0, 64, 2, 8><<<grid, 512, lds, stream>>>(g);
        }
    }
    else if (cfg == 1 || cfg == 4) TN2_MODE(32, 4);
    else if (cfg == 2) TN2_MODE(32, 3);
    else if (cfg == 3) TN2_MODE(64, 3);
    else TN2_MODE(64, 2);
#undef TN2_MODE
#undef TN2_LAUNCH
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
