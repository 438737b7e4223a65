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

struct Args { const unsigned char* src; unsigned fp_mask; unsigned rs; unsigned rs_shift /* log2(rs / 128) */; unsigned grp_mask; int iters; int inflight; long long* out; int share; unsigned fp; };

// MODE 0: buffer_load ... lds (descriptor + lane offset + scalar offset); 1: global_load_lds (flat per-lane pointer)
// All index arithmetic is 32-bit shifts and masks (the first version of this probe spent its time in two 64-bit divisions per piece).
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void probe(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a piece = 8 rows x 128 B; rs = 128 makes it 1 KiB contiguous.  The footprint is groups of 8 rows of rs bytes; successive pieces of
    // a group advance 128 B along the rows (the K steps of a weight tile), then the next group follows.
    const unsigned lane_off = (unsigned)(lane >> 3) * a.rs + (lane & 7) * 16;
    const unsigned char* base = a.src + (a.share ? 0 : (size_t)blockIdx.x * a.fp);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)a.fp, 0x00020000);
    long long t0 = 0;
    unsigned p = wave;                                                // waves take pieces round-robin
    auto issue = [&](unsigned slot) {
        const unsigned off = (((p >> a.rs_shift) & a.grp_mask) * 8 * a.rs + (p & ((1u << a.rs_shift) - 1)) * 128) & a.fp_mask;
        unsigned char* dst = smem + wave * 8192 + slot * 1024;
        if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)dst, 16, (int)lane_off, (int)off, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(base + off + lane_off), (lptr_t)dst, 16, 0, 0);
        p += NW;
    };
    for (int it = 0; it < a.iters; it += 4) {
        if (it == 8) { __syncthreads(); t0 = (long long)__builtin_amdgcn_s_memtime(); }
        issue(0); issue(1); issue(2); issue(3);
        switch (a.inflight) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) a.out[blockIdx.x * 2] = t1 - t0;
}

template <int MODE, int NW>
static double run(Args a, int blocks, double* ms_out) {
    hipFuncSetAttribute((const void*)probe<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NW><<<blocks, 64 * NW, NW * 8192, 0>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, NW><<<blocks, 64 * NW, NW * 8192, 0>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); *ms_out = ms;
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), a.out, blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0; for (int b = 0; b < blocks; ++b) cyc += (double)h[b * 2];
    cyc /= blocks;
    return (double)(a.iters - 8) * NW * 1024.0 / cyc;                // bytes per (s_memtime) clock per CU
}

static unsigned lg2(unsigned v) { unsigned r = 0; while ((1u << r) < v) ++r; return r; }

int main() {
    const size_t SRC = 512u << 20;
    unsigned char* src; if (hipMalloc(&src, SRC) != hipSuccess) return 1; hipMemset(src, 1, SRC);
    long long* out; hipMalloc(&out, 4096 * sizeof(long long));
    const int blocks = 256, iters = 4104;
    printf("%-18s %3s %8s %9s %9s %6s | %9s %10s\n", "mode", "NW", "inflight", "footprint", "rowstride", "shared", "B/clk/CU", "chip TB/s");
    struct Cfg { unsigned fp; unsigned rs; int share; } cfgs[] = {{16384, 128, 1}, {65536, 128, 1}, {1u << 20, 128, 1}, {1u << 20, 8192, 1}, {4u << 20, 8192, 1},
                                                                  {65536, 128, 0}, {1u << 20, 128, 0}};
    for (auto c : cfgs)
        for (int mode = 0; mode < 2; ++mode)
            for (int nw : {1, 2, 4, 8, 16})
                for (int inf : {0, 8, 32}) {
                    if (mode == 1 && (inf != 32 || (nw != 4 && nw != 8))) continue;
                    if (!c.share && (size_t)c.fp * blocks > SRC) continue;
                    const unsigned groups = c.fp / (8 * c.rs);
                    Args a = {src, c.fp - 1, c.rs, lg2(c.rs / 128), groups - 1, iters, inf, out, c.share, c.fp};
                    double ms = 0, bpc = 0;
#define RUN(M_, W_) bpc = run<M_, W_>(a, blocks, &ms)
                    if (mode == 0) { if (nw == 1) RUN(0, 1); else if (nw == 2) RUN(0, 2); else if (nw == 4) RUN(0, 4); else if (nw == 8) RUN(0, 8); else RUN(0, 16); }
                    else { if (nw == 4) RUN(1, 4); else RUN(1, 8); }
                    const double tbs = (double)iters * nw * 1024.0 * blocks / (ms * 1e-3) / 1e12;
                    printf("%-18s %3d %8d %9u %9u %6d | %9.2f %10.2f\n", mode == 0 ? "buffer_load..lds" : "global_load_lds", nw, inf, c.fp, c.rs, c.share, bpc, tbs);
                    fflush(stdout);
                }
    return 0;
}
