// Stand-alone microbenchmark (not part of libocrhip.so): how many bytes per clock can ONE CU pull from the L2 into its LDS with
// LDS-DMA, as a function of the number of issuing waves, the pieces kept in flight and the source pattern?  Background: round 3's
// conv_k2 kernel halves conv_halo's DMA bytes per flop and fragment reads per MFMA and runs at the SAME speed; both move ~20.5 KB of
// unique L2 -> CU bytes per 4.2 MFLOP K step and both take ~1900 clocks for it = ~10.5 B/clk/CU (the guide's "~10-13 B/cyc/CU").
//   build:  make -C lstm_ctc_ocr_amd/csrc experiments      run (GPU box):  tools/bin/dma_probe
// Every workgroup (one per CU, 256 of them) streams the SAME `footprint` bytes over and over (L2 hits after the first touch, as the
// weight tiles of a convolution are for the 8 CUs that share them), 1 KiB per wave-instruction, into a per-wave LDS ring.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Args { const unsigned char* src; long footprint; long row_stride; int iters; int inflight; long long* out; int share; };

// MODE 0: buffer_load ... lds (descriptor + lane offset + scalar offset); 1: global_load_lds (flat per-lane pointer);
// 2: global_load_dwordx4 into registers (no LDS)
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void probe(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a piece = 8 rows x 128 B; row_stride = 128 makes it 1 KiB contiguous
    const long lane_off = (long)(lane >> 3) * a.row_stride + (lane & 7) * 16;
    // the footprint is `groups` groups of 8 rows of row_stride bytes; successive pieces of a group advance 128 B along the rows (the K
    // steps of a weight tile), then the next group follows
    const long ksteps = a.row_stride / 128, groups = a.footprint / (8 * a.row_stride);
    const unsigned char* base = a.src + (a.share ? 0 : (long)blockIdx.x * a.footprint);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)a.footprint, 0x00020000);
    u32x4 sink = {0, 0, 0, 0};
    long long t0 = 0;
    long p = wave;                                                   // waves take pieces round-robin
    for (int it = 0; it < a.iters; ++it) {
        if (it == 4) { __syncthreads(); t0 = (long long)__builtin_amdgcn_s_memtime(); }
        const long off = ((p / ksteps) % groups) * 8 * a.row_stride + (p % ksteps) * 128;
        unsigned char* dst = smem + wave * 16384 + (it & 15) * 1024;
        if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)dst, 16, (int)lane_off, (int)off, 0, 0);
        else if (MODE == 1) __builtin_amdgcn_global_load_lds((gptr_t)(base + off + lane_off), (lptr_t)dst, 16, 0, 0);
        else { u32x4 v = *(const u32x4*)(base + off + lane_off); sink ^= v; }
        p += NW;
        if (MODE != 2) {
            switch (a.inflight) {
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { a.out[blockIdx.x * 2] = t1 - t0; a.out[blockIdx.x * 2 + 1] = (long long)(sink.x ^ sink.y ^ sink.z ^ sink.w); }
    if (MODE == 2 && sink.x == 0x12345u) a.out[0] = 0;
}

template <int MODE, int NW>
static double run(Args a, int blocks, double* ms_out) {
    hipFuncSetAttribute((const void*)probe<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NW><<<blocks, 64 * NW, NW * 16384, 0>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, NW><<<blocks, 64 * NW, NW * 16384, 0>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); *ms_out = ms;
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), a.out, blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0; for (int b = 0; b < blocks; ++b) cyc += (double)h[b * 2];
    cyc /= blocks;
    return (double)(a.iters - 4) * NW * 1024.0 / cyc;                // bytes per (s_memtime) clock per CU
}

int main() {
    const long SRC = 64L << 20;
    unsigned char* src; hipMalloc(&src, SRC); hipMemset(src, 1, SRC);
    long long* out; hipMalloc(&out, 4096 * sizeof(long long));
    const int blocks = 256, iters = 2052;
    printf("%-28s %4s %8s %9s %9s %7s | %8s %10s\n", "mode", "NW", "inflight", "footprint", "rowstride", "shared", "B/clk/CU", "chip TB/s");
    struct Cfg { long fp; long rs; int share; } cfgs[] = {{65536, 128, 1}, {1 << 20, 128, 1}, {589824, 9216, 1}, {4718592, 9216, 1}, {65536, 128, 0}, {1 << 20, 128, 0}, {4 << 20, 1024, 1}};
    for (auto c : cfgs)
        for (int mode = 0; mode < 3; ++mode)
            for (int nw : {4, 8})
                for (int inf : {2, 8, 32}) {
                    if (mode == 2 && inf != 32) continue;
                    Args a = {src, c.fp, c.rs, iters, inf, out, c.share};
                    double ms = 0, bpc = 0;
                    if (mode == 0) bpc = nw == 4 ? run<0, 4>(a, blocks, &ms) : run<0, 8>(a, blocks, &ms);
                    else if (mode == 1) bpc = nw == 4 ? run<1, 4>(a, blocks, &ms) : run<1, 8>(a, blocks, &ms);
                    else bpc = nw == 4 ? run<2, 4>(a, blocks, &ms) : run<2, 8>(a, blocks, &ms);
                    const double tbs = (double)iters * nw * 1024.0 * blocks / (ms * 1e-3) / 1e12;
                    printf("%-28s %4d %8d %9ld %9ld %7d | %8.2f %10.2f\n", mode == 0 ? "buffer_load..lds" : mode == 1 ? "global_load_lds" : "global_load_dwordx4->vgpr",
                           nw, inf, c.fp, c.rs, c.share, bpc, tbs);
                    fflush(stdout);
                }
    return 0;
}
