// Third-generation plain weight-gradient product for gfx950 (round 4):
//     out[i][j] += scale * sum_m A[m][i] * B[m][j]            (contraction over the slow index of both operands)
// for the X^T dY products of conv5 and of the BiLSTM cells (reference: tf.gradients of network.py:107,118-126 / 166 at train.py:81) —
// ONE launch for several products ("jobs"), one workgroup per 128 x 128 output tile over the WHOLE contraction: no split over m, hence no
// atomics and no partial-sum traffic, and bit-reproducible results.
//
// Why (profiles/r04b_tn_sweep.log): gemm_tn2's eight waves run in lock-step — DMA issue, 24 transposing reads, 16 MFMAs, barrier, one after
// the other: ~1800 clocks per 64-row step of which 512 are MFMA, whatever the pipeline depth (2, 3 or 4 stages measured equal) — so it
// needs 4 splits to fill the chip and then pays 0.8 us per MB of fp32 atomics (41 MB for the two products: most of their 58 us).
// Here the schedule is conv_k3.hip's: the two wave groups of a workgroup take the two K halves (rows 0-31 / 32-63) of every 64-row stage
// and run half an interval apart — one group multiplies on fragments it already holds while the other reads LDS and issues DMA — with ONE
// barrier per stage, four LDS stages (prefetch distance 2) and counted vmcnt waits; the K halves meet once through LDS at the end.
// Data layout, DMA swizzle and the transposing fragment reads are gemm_tn2.hip's (pinned by test_probe_tr16 / tools/wgrad9_model.py):
// a stage is [64 rows of m][128 channels] per operand, row-major as the DMA writes it, the 32-byte slot index XOR (row & 7) applied to the
// source chunk and again in the read address; ds_read_b64_tr_b16 delivers 4 consecutive m of one channel per lane.
// Requires I % 128 == 0, J % 128 == 0; other shapes stay on gemm_tn2 / gemm_tn.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct Tn3Job {                     // = ocr_tn_job (include/ocr_hip.h), element strides
    const bf16_t* A; long lda; long sA;
    const bf16_t* B; long ldb; long sB;
    float* out; long ldo; long sO;
    float* colsum; long sC;
    int Mk, I, J, nbatch;
    int grp, skip;                  // A's rows in groups of `grp` with `skip` unused rows behind each group (conv5 over overlapping rows); 0 = plain
    float scale; int reserved;
};
static_assert(sizeof(Tn3Job) == 120, "mirrored by lstm_ctc_ocr_amd/ops.py (ctypes.Structure)");

__device__ u32x4 tn3_zero_page[4];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// byte offset (inside a swizzled [64][128] tile, rows 0..31) of the first transposing read of channel block cb: tile rows {4g .. 4g+3}
// (+16 for the second read, +32 for the other K half), channel cb*16 + lane&15        (gemm_tn2.hip: tn2_frag_off)
__device__ __forceinline__ int tn3_frag_off(int cb, int lane) {
    const int g = lane >> 4, L = lane & 15;
    const int r0 = 4 * g + (L >> 2);
    const int q = cb * 2 + ((L & 3) >> 1);
    const int p = q ^ ((r0 & 7) << 1);
    return r0 * 256 + p * 16 + (L & 1) * 8;
}

template <int KH, int NST /* LDS stages: 4 (prefetch distance 2) or 5 (distance 3: all 160 KiB) */>
__device__ __forceinline__ void tn3_body(const Tn3Job& g, const int bid, unsigned char* smem) {
    constexpr int TILE = 64 * 256, STAGE = 2 * TILE, DEPTH = NST - 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = (wave & 3) >> 1, wj = wave & 1;

    const int IT = g.I >> 7, JT = g.J >> 7, T = IT * JT * g.nbatch;
    int L = bid;
    if ((T & 7) == 0) L = (bid & 7) * (T >> 3) + (bid >> 3);          // XCD-aware: each XCD (private L2) owns a contiguous run of tiles
    // an XCD's run of tiles re-reads one operand from every XCD and keeps the other disjoint: the LARGER operand stays disjoint (conv5: A = 16.5 MB
    // against B = 4 MB -> j fastest, an XCD owns two i-tiles; a BiLSTM cell: A = 6.2 MB, B = 8.3 MB -> i fastest, an XCD owns two j-tiles)
    int jt, it;
    if ((long)g.I >= (long)g.J) { jt = L % JT; it = (L / JT) % IT; }
    else { it = L % IT; jt = (L / IT) % JT; }
    const int bi = L / (JT * IT);
    const bf16_t* A = g.A + (long)bi * g.sA;
    const bf16_t* B = g.B + (long)bi * g.sB;
    float* out = g.out + (long)bi * g.sO;
    const int i0 = it * 128, j0 = jt * 128, Mk = g.Mk;
    const bool do_cs = g.colsum != nullptr && it == 0;
    const bf16_t* zero = (const bf16_t*)tn3_zero_page;

    // ---- DMA geometry: piece j of wave w covers tile rows (w*2 + j)*4 .. +3; lane -> (row lane>>4, 16-byte position lane&15), which holds
    //      source chunk pos ^ ((row & 7) << 1).  Row m of stage t is t*64 + r; A's physical row m + (m / grp) * skip is tracked incrementally.
    const int rsub = lane >> 4, pos = lane & 15;
    const bf16_t* pa[2]; const bf16_t* pb[2];
    int mrow[2], mq[2], mr[2];
    const int grp = g.grp, dq = grp > 0 ? 64 / grp : 0, dr = grp > 0 ? 64 % grp : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 4 + rsub;
        const int q = pos ^ ((r & 7) << 1);
        mrow[j] = r;
        mq[j] = grp > 0 ? r / grp : 0; mr[j] = grp > 0 ? r % grp : 0;
        pa[j] = A + i0 + q * 8;
        pb[j] = B + j0 + q * 8;
    }
    auto stage_load = [&](int buf) {                 // issues this wave's 4 pieces of the NEXT stage in sequence and advances the row state
        unsigned char* sa = smem + buf * STAGE;
        unsigned char* sb = sa + TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mrow[j];
            const bf16_t* srca = zero;
            const bf16_t* srcb = zero;
            if (m < Mk) {
                srca = pa[j] + ((long)m + (long)mq[j] * g.skip) * g.lda;
                srcb = pb[j] + (long)m * g.ldb;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)srca, (lptr_t)(sa + (wave * 2 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)srcb, (lptr_t)(sb + (wave * 2 + j) * 1024), 16, 0, 0);
            mrow[j] = m + 64;
            if (grp > 0) {
                int r2 = mr[j] + dr, q2 = mq[j] + dq;
                if (r2 >= grp) { r2 -= grp; ++q2; }
                mr[j] = r2; mq[j] = q2;
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned offf[8];                                // fragment addresses inside a stage: 4 A blocks, 4 B blocks, this group's K half
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        offf[a] = lds0 + tn3_frag_off(wi * 4 + a, lane) + KH * 32 * 256;
        offf[4 + a] = lds0 + TILE + tn3_frag_off(wj * 4 + a, lane) + KH * 32 * 256;
    }

    const int nst = (Mk + 63) >> 6;
    // ---- prologue: stages 0 .. DEPTH - 1, landed before barrier 0
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) stage_load(d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    s16x4 flo[8], fhi[8];                            // fragments: group 1 carries them across the barrier
    auto load = [&](int buf) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const unsigned ad = offf[f] + buf * STAGE;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(flo[f]) : "v"(ad));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(fhi[f]) : "v"(ad));
        }
    };
    auto comp = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(KH == 1 ? 2 : 1);  // the group that multiplies FIRST in an interval drains before its partner starts (conv_k3.hip)
        bf16x8 fr[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            s16x8 v = {flo[f][0], flo[f][1], flo[f][2], flo[f][3], fhi[f][0], fhi[f][1], fhi[f][2], fhi[f][3]};
            fr[f] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[a], fr[4 + b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // One interval = one 64-row stage, closed by ONE barrier.  Group 0: LOAD(s), DMA(s + DEPTH), COMP(s) [, column sums of B(s)];
    // group 1: COMP(s - 1), LOAD(s), DMA(s + DEPTH).  Stage s + DEPTH goes into the buffer of stage s - 2, whose last readers (group 1's
    // LOAD(s - 2), retired by the lgkmcnt(0) in front of its COMP in interval s - 1) passed barrier s - 1.  Every wave issues exactly 4 pieces
    // per interval (rows past the contraction come from a zero page), so "stage s + 1 has landed" is vmcnt(4 * (DEPTH - 1)).
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
        int nb = buf + DEPTH; if (nb >= NST) nb -= NST;
        if (KH == 0) {
            load(buf);
            stage_load(nb);
            comp();
            if (do_cs) {                              // threads 0..255: source chunk q = tid & 15 (8 columns), rows (tid >> 4) + 16 i of B(s)
                // (inline asm like the fragment reads: behind an LDS-DMA hipcc drains vmcnt(0) in front of every LDS read it can see)
                const unsigned tb = lds0 + buf * STAGE + TILE;
                const int q = tid & 15;
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = (tid >> 4) + 16 * i;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(tb + r * 256 + ((q ^ ((r & 7) << 1)) << 4)));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cs[0] += bf_lo(v[i].x); cs[1] += bf_hi(v[i].x); cs[2] += bf_lo(v[i].y); cs[3] += bf_hi(v[i].y);
                    cs[4] += bf_lo(v[i].z); cs[5] += bf_hi(v[i].z); cs[6] += bf_lo(v[i].w); cs[7] += bf_hi(v[i].w);
                }
            }
        } else {
            if (s > 0) comp();
            load(buf);
            stage_load(nb);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (KH == 1) comp();                              // the last stage's second K half
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pieces streamed past the last stage: LDS is reused below
    __syncthreads();

    // ---- the two K halves meet: a wave keeps the i-blocks [KH*2, KH*2 + 2) of its 64 x 64 tile and hands the others to its partner (wave ^ 4)
    {
        f32x4* mine = (f32x4*)smem + (size_t)wave * (8 * 64);
        const f32x4* theirs = (const f32x4*)smem + (size_t)(wave ^ 4) * (8 * 64);
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 4; ++b) mine[(a2 * 4 + b) * 64 + lane] = acc[(1 - KH) * 2 + a2][b];
        __syncthreads();
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[KH * 2 + a2][b] += theirs[(a2 * 4 + b) * 64 + lane];
    }
    if (do_cs) {                                      // block-uniform
        __syncthreads();                              // the exchange region has been read
        float* red = (float*)smem;
        if (KH == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
        }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
            for (int u = tid >> 3; u < 256; u += 16) t += red[u * 8 + (tid & 7)];
            g.colsum[(long)bi * g.sC + j0 + tid] += t * g.scale;       // this tile row (it == 0) is the only writer of these 128 columns
        }
    }
    // lane owns i = ib + (lane >> 4) * 4 + r, j = jb + (lane & 15); this workgroup is the only writer of the tile: plain read-modify-write,
    // all 32 loads issued before the first store (written as one loop the compiler waits for every load in front of the store behind it)
    const float scale = g.scale;
    float old[2][4][4];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                old[a2][b][r] = __builtin_nontemporal_load(out + (long)(i0 + wi * 64 + (KH * 2 + a2) * 16 + (lane >> 4) * 4 + r) * g.ldo + j0 + wj * 64 + b * 16 + (lane & 15));
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(long)(i0 + wi * 64 + (KH * 2 + a2) * 16 + (lane >> 4) * 4 + r) * g.ldo + j0 + wj * 64 + b * 16 + (lane & 15)] =
                    old[a2][b][r] + acc[KH * 2 + a2][b][r] * scale;
}

template <int NST>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn3_kernel(Tn3Job j0, Tn3Job j1, int nblk0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int kh = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int bid = (int)blockIdx.x;
    if (bid < nblk0) {
        if (kh == 0) tn3_body<0, NST>(j0, bid, smem); else tn3_body<1, NST>(j0, bid, smem);
    } else {
        if (kh == 0) tn3_body<0, NST>(j1, bid - nblk0, smem); else tn3_body<1, NST>(j1, bid - nblk0, smem);
    }
}

static bool tn3_covers(const Tn3Job& j) {
    if (!j.A || !j.B || !j.out || j.Mk < 256 || j.I <= 0 || j.J <= 0 || (j.I & 127) || (j.J & 127) || j.nbatch < 1) return false;
    if ((j.lda & 7) || (j.ldb & 7) || (j.sA & 7) || (j.sB & 7) || ((size_t)j.A & 15) || ((size_t)j.B & 15)) return false;     // 16-byte DMA chunks
    if (j.grp < 0 || j.skip < 0 || (j.grp == 0 && j.skip != 0)) return false;
    return (long)(j.I >> 7) * (j.J >> 7) * j.nbatch <= 0x7fffffL;
}
// 1 = ocr_gemm_tn_jobs_bf16 takes these jobs (host-only query)
extern "C" int ocr_gemm_tn_jobs_supported(const void* jobs, int njobs) {
    if (!jobs || njobs < 1 || njobs > 2) return 0;
    const Tn3Job* j = (const Tn3Job*)jobs;
    for (int k = 0; k < njobs; ++k) if (!tn3_covers(j[k])) return 0;
    return 1;
}
// jobs: HOST array of 1 or 2 ocr_tn_job; every product out_b[I][J] += scale * A_b^T B_b (b < nbatch) and colsum_b[J] += scale * column sums of
// B_b of both jobs in ONE launch.  OCR_STATUS_INVALID where a job is not covered (the caller then uses ocr_gemm_tn_bf16 / _batched).
extern "C" int ocr_gemm_tn_jobs_bf16(const void* jobs, int njobs, void* stream) {
    if (!ocr_gemm_tn_jobs_supported(jobs, njobs)) return OCR_ERR_INVALID;
    const Tn3Job* j = (const Tn3Job*)jobs;
    const int t0 = (j[0].I >> 7) * (j[0].J >> 7) * j[0].nbatch;
    const int t1 = njobs > 1 ? (j[1].I >> 7) * (j[1].J >> 7) * j[1].nbatch : 0;
    static int nst = -1;                             // kernel-selection knob OCR_TN3_NST: 4 stages (prefetch distance 2) / 5 (distance 3, all 160 KiB of LDS)
    if (nst < 0) { const char* e = getenv("OCR_TN3_NST"); nst = (e && atoi(e) == 5) ? 5 : 4; }      // measured equal (profiles/r04d: 1.2682 / 1.2681 ms per step): 4
    static bool attr[2] = {false, false};
    const int lds = nst * 2 * 64 * 256;              // stages x (A tile | B tile); the K-half exchange (64 KiB) reuses them
    if (nst == 4) {
        if (!attr[0]) { if (hipFuncSetAttribute((const void*)gemm_tn3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC; attr[0] = true; }
        gemm_tn3_kernel<4><<<t0 + t1, 512, lds, (hipStream_t)stream>>>(j[0], njobs > 1 ? j[1] : j[0], t0);
    } else {
        if (!attr[1]) { if (hipFuncSetAttribute((const void*)gemm_tn3_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC; attr[1] = true; }
        gemm_tn3_kernel<5><<<t0 + t1, 512, lds, (hipStream_t)stream>>>(j[0], njobs > 1 ? j[1] : j[0], t0);
    }
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
