// Shared device/host helpers for the gfx950 (MI355X, CDNA4) CRNN-OCR kernels.
// Everything in this directory is written for wave64 / MFMA / 160 KiB LDS only;
// there is deliberately no other backend.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OCR_WAVE 64

// Status codes mirror warp-ctc's ctcStatus_t numbering (0 ok, 1 memops, 2 invalid
// value, 3 execution failed) so a caller written against that ABI keeps working.
enum {
    OCR_OK = 0,
    OCR_ERR_MEMOPS = 1,
    OCR_ERR_INVALID = 2,
    OCR_ERR_EXEC = 3,
};

#define OCR_CHECK_LAUNCH()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return OCR_ERR_EXEC;               \
    } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// round-to-nearest-even fp32 -> bf16 (same rule as torch's .to(bfloat16)); NaN kept quiet.
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (gfx950 native, round-to-nearest-even like f2bf)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// hardware-rate gate non-linearities: v_exp_f32 + v_rcp_f32 (relative error ~1e-6, far below the bf16 storage of h)
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    float e = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow
    float t = (1.0f - e) * __frcp_rn(1.0f + e);
    return copysignf(t, x);
}

// LDS-DMA of 16 bytes per lane in SADDR form: address = uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset (one VGPR), LDS
// destination = uniform base in M0 + lane * 16.  With a loop-invariant lane offset and a scalar base that advances per step a
// piece costs a handful of scalar instructions — the builtin with a per-lane 64-bit pointer (+ zero-page select) measured
// 200-260 cycles per piece in the convolution kernels (tools/pp_stamps.py), mostly address arithmetic.
// Not counted by the compiler: the caller waits with s_waitcnt vmcnt(N).  s_nop 4: an SGPR written by SALU just before may not
// be read by VMEM earlier; s_nop 0: M0 write -> LDS-DMA.  M0 is saved / restored (the compiler owns it).
__device__ __forceinline__ void dma16_saddr(unsigned lds_dst, unsigned voff, const void* sbase) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// Tuning knobs (tile policies, stage counts, ...): read from the environment only by `make EXPERIMENTS=1` builds — the A/B tools load that
// library through OCR_NATIVE_LIB; the product library uses the measured defaults.  Kernel-selection knobs that the parity tests force
// (OCR_CONV_K2 / OCR_CONV_K3 / OCR_K2_CFG / OCR_W9_PLANES / OCR_LSTM_PROTO / OCR_LSTM_ROWS) are read by both.
static inline const char* ocr_tune_env(const char* name) {
#ifdef OCR_EXPERIMENTS
    return getenv(name);
#else
    (void)name; return nullptr;
#endif
}
// Clock diagnostic of the convolution kernels (ocr_conv_halo_clock_debug, device int64[8]): workgroup 0's first thread stamps
// {shader-clock counter (s_memtime), 100 MHz wall clock (s_memrealtime)} into [0], [1] at entry and [2], [3] at exit, and ADDS its lifetime to
// [4] (shader clocks), [5] (wall ticks), [6] (launches): launches of one stream are serial, so plain adds do.  MHz = [4] / [5] * 100 over
// every stamped launch of a run — the counter passes divide the matrix pipes' busy cycles by the kernels' OWN cycles with it.
__device__ __forceinline__ long long ocr_wall_clock() { return (long long)__builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void ocr_clk_enter(long long* clk) {
    clk[0] = (long long)__builtin_amdgcn_s_memtime(); clk[1] = ocr_wall_clock();
}
__device__ __forceinline__ void ocr_clk_exit(long long* clk) {
    const long long c = (long long)__builtin_amdgcn_s_memtime(), t = ocr_wall_clock();
    clk[2] = c; clk[3] = t;
    clk[4] += c - clk[0]; clk[5] += t - clk[1]; clk[6] += 1;
}
__host__ __device__ static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
