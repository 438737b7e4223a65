// Library identity, status strings and device probes.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe_tr16_kernel(const int* __restrict__ addr, int* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    int a = addr[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + a));
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (int)v[j];
}

extern "C" int ocr_probe_tr16(const int* addr, int* out, void* stream) {
    if (!addr || !out) return OCR_ERR_INVALID;
    probe_tr16_kernel<<<1, 64, 0, (hipStream_t)stream>>>(addr, out);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

// Which XCD does workgroup `id` of a 1-D grid land on?  out[id] = XCC_ID hardware register (id 20, bits 3:0), out[n + id] = CU id
// (HW_ID register: id 4).  Evidence for the "workgroup id & 7 == XCD" placement the XCD-aware tile maps assume.
__global__ void probe_xcc_kernel(int* __restrict__ out, int n) {
    if (threadIdx.x == 0) {
        out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        out[n + blockIdx.x] = (int)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    }
}
extern "C" int ocr_probe_xcc(int* out, int nblocks, int threads, void* stream) {
    if (!out || nblocks <= 0 || threads <= 0 || threads > 1024) return OCR_ERR_INVALID;
    probe_xcc_kernel<<<nblocks, threads, 0, (hipStream_t)stream>>>(out, nblocks);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

// Calibration of the SQ_VALU_MFMA_BUSY_CYCLES counter (the busy fraction and the FLOP fraction of conv_k3 did not reconcile).
// Every wave issues `iters` x 8 v_mfma_f32_16x16x32_bf16 on eight independent accumulators (no operand loads, no LDS, no waits inside the
// loop), so the matrix pipe of its SIMD is saturated by construction and the number of MFMA instructions is known exactly:
// grid x (threads / 64) x iters x 8.  At 1024 flop / clock / SIMD (2.5 PFLOP/s = 256 CUs x 4 SIMDs x 1024 x 2.4 GHz) one such MFMA occupies
// the pipe for 16 clocks.  clk[0..3] = shader clock / 100 MHz wall clock of workgroup 0 at loop entry and exit (the clock the loop ran at).
// tools/mfma_busy_probe.py runs it under rocprofv3 --pmc and derives the counter's unit.
__global__ void __launch_bounds__(512) mfma_busy_probe_kernel(float* __restrict__ out, int iters, long long* __restrict__ clk) {
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 3); b[j] = (__bf16)(float)((threadIdx.x * 3 + j) & 1); }
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = (long long)__builtin_amdgcn_s_memtime(); clk[1] = (long long)wall_clock64(); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = (long long)__builtin_amdgcn_s_memtime(); clk[3] = (long long)wall_clock64(); }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (r == 123456.789f) out[0] = r;             // keeps the accumulators alive; never true for these operands
}
extern "C" int ocr_mfma_busy_probe(float* out, int nblocks, int threads, int iters, long long* clk, void* stream) {
    if (!out || nblocks <= 0 || threads <= 0 || threads > 512 || (threads & 63) || iters <= 0) return OCR_ERR_INVALID;
    mfma_busy_probe_kernel<<<nblocks, threads, 0, (hipStream_t)stream>>>(out, iters, clk);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

// One-GPU stand-in for the CUs a concurrent RCCL collective occupies (OCR_FAKE_WORLD's doubling kernel is a ~10 us
// elementwise pass and cannot show what a 0.2-0.3 ms all-reduce kernel does to the convolution launches running beside it).  `nblocks`
// workgroups of `threads` threads, each holding `lds_bytes` of LDS, stay resident for `us` microseconds (100 MHz wall clock, s_sleep
// between looks: no issue slots, no memory traffic — the CU's LDS and wave slots are what is taken, as by RCCL's long-running channel
// kernels, one workgroup per channel).  A workgroup with more than half of a CU's LDS keeps every 160 KB convolution tile off its CU.
__global__ void occupy_kernel(long long ticks, int* __restrict__ sink) {
    extern __shared__ int occ_lds[];
    const long long t0 = ocr_wall_clock();
    if (sink == (int*)1) occ_lds[threadIdx.x] = 1;          // (never: keeps the dynamic LDS allocation referenced)
    while (ocr_wall_clock() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink == (int*)1) sink[0] = occ_lds[0];
}
extern "C" int ocr_occupy_cus(int nblocks, int threads, int lds_bytes, float us, void* stream) {
    if (nblocks <= 0 || nblocks > 1024 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || !(us >= 0.f) || us > 1e5f) return OCR_ERR_INVALID;
    static int attr = 0;
    if (lds_bytes > attr) {
        if (hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OCR_ERR_EXEC;
        attr = 160 * 1024;
    }
    occupy_kernel<<<nblocks, threads, lds_bytes, (hipStream_t)stream>>>((long long)(us * 100.0f), nullptr);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}

extern "C" const char* ocr_status_string(int status) {
    switch (status) {
        case OCR_OK: return "no error";
        case OCR_ERR_MEMOPS: return "cuda memcpy or memset failed";      // wording kept from warp-ctc's ctcGetStatusString
        case OCR_ERR_INVALID: return "invalid value";
        case OCR_ERR_EXEC: return "execution failed";
        default: return "unknown error";
    }
}
extern "C" int ocr_abi_version(void) { return 1; }
