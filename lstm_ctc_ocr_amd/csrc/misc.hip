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
