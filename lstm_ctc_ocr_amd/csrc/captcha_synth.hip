// GPU-side captcha synthesis: the training images of the reference's generator
// (/root/reference/lib/lstm/utils/gen.py:31-37 generateImg -> captcha.ImageCaptcha, :41-67 groupBatch: resize to height 32, [W, 32] rows,
// right-padded with 0) composed on the device from a resident atlas of glyph masks.  One workgroup per image, the whole image in LDS:
//
//   A  canvas[60][canvas_w] = background; per glyph, in order: the rotated coverage (affine transform of the atlas mask with bilinear sampling,
//      double arithmetic, (UINT8) truncation — Pillow's Geometry.c affine_transform + bilinear_filter8) blended in with the paste's
//      DIV255(out * (255 - m) + ink * m)
//   B  canvas wider than the captcha: horizontal bicubic resampling to [60][width] (Resample.c: double coefficients normalised per output
//      column, quantised to 22 fractional bits, integer accumulation)
//   C  30 noise dots (the footprint of ImageDraw's 3-wide diagonal line, passed in) and the noise arc (thin ellipse outline between the two
//      end normals — the one primitive that is NOT Pillow's algorithm, see tools/synth_model.py arc_pixels)
//   D  3 x 3 SMOOTH (Filter.c ImagingFilter3x3: float32, border pixels copied)
//   E  bilinear resampling to [32][nw_out], horizontal pass then vertical pass, each rounded to 8 bits (Resample.c again)
//   F  [W][32] uint8 rows to HBM, columns >= nw_out zero
//
// The parameters (every random draw and the integer geometry that follows) come from the host: lstm_ctc_ocr_amd/utils/synth.draw_params.
// Bit-exact against the numpy model tools/synth_model.py, which is bit-exact against Pillow 12 itself without the arc (tests/test_synth.py,
// tests/test_gpu_synth.py) — hence no floating-point contraction in this file: Pillow's C code is compiled without fused multiply-adds.
// HBM traffic per image: ~400 bytes of parameters + <= 30 KB of atlas masks (L2-resident: the atlas is 285 KB) read, W * 32 bytes written.
// Bound: neither HBM nor the matrix pipes, and not the vector ALU's throughput either — the LATENCY of one workgroup's dependent work: counters of
// the launch alone (tools/prof_synth_pmc.sh, profiles/r06*_synth_pmc.txt, average of the three workloads): 1.0 MB read + 0.44 MB written per launch
// (0.015 TB/s), 4.6 M vector + 2.9 M scalar + 0.44 M LDS wave instructions, vector ALU active 30 % of the 64 busy CUs' cycles, waves waiting on
// a dependency half of theirs (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.50), LDS bank conflicts 16 % of the LDS-active cycles.  Per image the critical
// path is: glyph passes one after the other (a barrier each), double-precision coordinate / weight chains per pixel, one IEEE division per
// resampling tap on the few waves that own a column — all prescribed by Pillow's arithmetic and order.  Measured alone (tools/synth_bench.py,
// batch of 64): 57 us at W = 88 (4-6 glyphs), 104 us at W = 256 (10 glyphs) = 1.1 M / 0.6 M images/s, 13x / 11x what the training step consumes.
// By stage, cut off one after the other in the version before this one (63 / 126 us; profiles/r06l_synth_stages.log, us at W = 88 / 256): empty
// launch 8 / 8, A 15 / 27, B 20 / 32 (the slowest image of the batch decides: one that needs the bicubic pass), C 1.4 / 2.2, D 9 / 26, E rows
// 6 / 15, E columns + stores 3 / 15; this version took the runtime divisions out of the resampling loops (a wave per row), four pixels per thread
// in D and 16-byte stores through LDS in F: -6 / -22 us, bit-identical.  Beside a training step the launch holds 64 CUs' LDS and a third of their
// issue slots for 57-104 us of a 750-1140 us step: the live loop runs at 0.97-0.98x of the device-resident rate.
#include "common.h"
#pragma clang fp contract(off)

#define SY_H 60
#define SY_HDR 32
#define SY_NDOTS 30
#define SY_GW 20
#define SY_TAPS 32
#define SY_PREC 22
#define SY_NT 1024           // threads per image: 16 waves, 4 per SIMD — with 4 the LDS / L2 latencies of every stage were exposed (173 us per batch)
#define SY_STAGE_G 12        // glyph masks staged in LDS before stage A (<= 3072 bytes each; others are sampled from L2)

struct SynthArgs {
    const int* params; int S; int G;
    const uint8_t* atlas; const int* stamp; int nstamp;
    uint8_t* out; int W; int ccap; int wcap; int out_h;
};

__device__ __forceinline__ double sy_dbl(const int* p) {
    return __hiloint2double(p[1], p[0]);
}
__device__ __forceinline__ double sy_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
__device__ __forceinline__ double sy_bilinear(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output position xx: taps -> coef[k * 256 + slot], returns (first input index, count)
template <bool BICUBIC>
__device__ __forceinline__ void sy_coeffs(int in_size, int out_size, int xx, int slot, int* coef, int* cmin, int* cn) {
    const double scale = (double)in_size / (double)out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = (BICUBIC ? 2.0 : 1.0) * fs;
    const double ss = 1.0 / fs;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    int n = xmax - xmin;
    if (n > SY_TAPS) n = SY_TAPS;                       // (the host refuses such ratios: utils/synth.draw_params)
    double ww = 0.0;
    for (int x = 0; x < n; ++x) {
        const double t = (x + xmin - center + 0.5) * ss;
        ww += BICUBIC ? sy_bicubic(t) : sy_bilinear(t);
    }
    for (int x = 0; x < n; ++x) {
        const double t = (x + xmin - center + 0.5) * ss;
        double w = BICUBIC ? sy_bicubic(t) : sy_bilinear(t);
        if (ww != 0.0) w = w / ww;
        coef[x * 256 + slot] = w < 0 ? (int)(-0.5 + w * (double)(1 << SY_PREC)) : (int)(0.5 + w * (double)(1 << SY_PREC));
    }
    cmin[slot] = xmin; cn[slot] = n;
}
// One resampling pass over `lines` independent lines: element i of line r is src[r * s_line + i * s_pos]; output likewise in dst.  Both in LDS
// (forced inline: the compiler then sees the address space and issues ds_ instead of flat_ accesses).  ROWS: a wave takes a line, its lanes the
// output positions (horizontal passes: positions are contiguous bytes); otherwise a thread takes one (line, position) pair with the positions
// fastest (vertical pass: 32 positions per line) — no runtime division in either loop (two per output were a third of the instructions).
template <bool BICUBIC, bool ROWS>
__device__ __forceinline__ void sy_resample(const uint8_t* src, int s_line, int s_pos, int in_size, int lines,
                                            uint8_t* dst, int d_line, int d_pos, int out_size, int* coef, int* cmin, int* cn) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c0 = 0; c0 < out_size; c0 += 256) {
        const int nc = min(256, out_size - c0);
        if ((int)threadIdx.x < nc) sy_coeffs<BICUBIC>(in_size, out_size, c0 + threadIdx.x, threadIdx.x, coef, cmin, cn);
        __syncthreads();
        if (ROWS) {
            for (int c = lane; c < nc; c += 64) {
                const int x0 = cmin[c], n = cn[c];
                for (int r = wave; r < lines; r += SY_NT / 64) {
                    const uint8_t* s = src + r * s_line + x0 * s_pos;
                    int acc = 1 << (SY_PREC - 1);
                    for (int k = 0; k < n; ++k) acc += (int)s[k * s_pos] * coef[k * 256 + c];
                    acc >>= SY_PREC;
                    dst[r * d_line + (c0 + c) * d_pos] = (uint8_t)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
                }
            }
        } else {
            const int total = nc * lines;
            const bool pow32 = nc == 32;
            for (int idx = threadIdx.x; idx < total; idx += SY_NT) {
                const int r = pow32 ? idx >> 5 : idx / nc, c = pow32 ? idx & 31 : idx - (idx / nc) * nc;
                const int x0 = cmin[c], n = cn[c];
                const uint8_t* s = src + r * s_line + x0 * s_pos;
                int acc = 1 << (SY_PREC - 1);
                for (int k = 0; k < n; ++k) acc += (int)s[k * s_pos] * coef[k * 256 + c];
                acc >>= SY_PREC;
                dst[r * d_line + (c0 + c) * d_pos] = (uint8_t)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SY_NT) void captcha_synth_kernel(SynthArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* canvas = smem;                                   // [60][canvas_w]; later the smoothed image [60][width]
    uint8_t* img = smem + SY_H * a.ccap;                      // [60][width]; later the horizontally reduced image [60][nw_out]
    int* coef = (int*)(smem + SY_H * (a.ccap + a.wcap));      // [SY_TAPS][256]
    int* cmin = coef + SY_TAPS * 256;
    int* cn = cmin + 256;
    const int tid = threadIdx.x;
    const int* p = a.params + (long)blockIdx.x * a.S;
    const int width = p[0], cw = p[1], nw_out = p[2];
    const int L = min(p[3], a.G);
    const int bg = p[4], fg = p[5];
    const int ws = (width + 3) & ~3;                          // row stride of img / the smoothed image: dword-aligned rows for stage D

    // A: background, glyphs in order.  The masks come out of the atlas ONCE, all of them in flight together, into the LDS the later stages use
    // (img + coefficient table, idle until stage B): sampled straight from L2, four dependent byte loads per pixel made this stage most of
    // the kernel's time.
    for (int i = tid; i < SY_H * cw; i += SY_NT) canvas[i] = (uint8_t)bg;
    const int stage_cap = SY_H * a.wcap + (SY_TAPS * 256 + 512) * 4;
    {
        uint8_t v[SY_STAGE_G][3];
#pragma unroll
        for (int g = 0; g < SY_STAGE_G; ++g) {
            const int* q = p + SY_HDR + 2 * SY_NDOTS + (g < L ? g : 0) * SY_GW;
            const int sz = g < L ? q[1] * q[2] : 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) v[g][j] = (tid + j * SY_NT < sz) ? a.atlas[q[0] + tid + j * SY_NT] : (uint8_t)0;
        }
        int off = 0;
#pragma unroll
        for (int g = 0; g < SY_STAGE_G; ++g) {
            const int* q = p + SY_HDR + 2 * SY_NDOTS + (g < L ? g : 0) * SY_GW;
            const int sz = g < L ? q[1] * q[2] : 0;
            if (sz <= 3 * SY_NT && off + sz <= stage_cap) {
#pragma unroll
                for (int j = 0; j < 3; ++j) if (tid + j * SY_NT < sz) img[off + tid + j * SY_NT] = v[g][j];
            }
            off += (sz + 15) & ~15;
        }
    }
    __syncthreads();
    int moff = 0;
    for (int g = 0; g < L; ++g) {
        const int* q = p + SY_HDR + 2 * SY_NDOTS + g * SY_GW;
        const int mw = q[1], mh = q[2], px = q[3], py = q[4], nw = q[5], nh = q[6];
        const bool staged = g < SY_STAGE_G && mw * mh <= 3 * SY_NT && moff + mw * mh <= stage_cap;
        const uint8_t* mask = staged ? (const uint8_t*)(img + moff) : a.atlas + q[0];
        moff += (mw * mh + 15) & ~15;
        const double m0 = sy_dbl(q + 8), m1 = sy_dbl(q + 10), m2 = sy_dbl(q + 12), m3 = sy_dbl(q + 14), m4 = sy_dbl(q + 16), m5 = sy_dbl(q + 18);
        const unsigned rnw = nw > 1 ? (unsigned)((0x100000000ull + nw - 1) / nw) : 0u;       // i / nw = umulhi(i, ceil(2^32 / nw)) for i * nw < 2^32
        for (int i = tid; i < nw * nh; i += SY_NT) {
            const int yo = nw > 1 ? (int)__umulhi((unsigned)i, rnw) : i, xo = i - yo * nw;
            const int X = px + xo, Y = py + yo;
            if (X < 0 || X >= cw || Y < 0 || Y >= SY_H) continue;
            double xin = m0 * (xo + 0.5) + m1 * (yo + 0.5) + m2;
            double yin = m3 * (xo + 0.5) + m4 * (yo + 0.5) + m5;
            if (xin < 0.0 || xin >= (double)mw || yin < 0.0 || yin >= (double)mh) continue;       // fill 0: the blend leaves the pixel
            xin -= 0.5; yin -= 0.5;
            const int x = (int)floor(xin), y = (int)floor(yin);
            const int x0 = x < 0 ? 0 : x < mw ? x : mw - 1;
            const int x1 = x + 1 < 0 ? 0 : x + 1 < mw ? x + 1 : mw - 1;
            const int yc = y < 0 ? 0 : y < mh ? y : mh - 1;
            const bool row2 = y + 1 >= 0 && y + 1 < mh;
            const uint8_t* in = mask + yc * mw;
            const uint8_t* in2 = mask + (row2 ? y + 1 : yc) * mw;
            const int t00 = in[x0], t01 = in[x1], t10 = in2[x0], t11 = in2[x1];
            if ((t00 | t01 | t10 | t11) == 0) continue;                                           // most of a rotated box is empty: coverage 0
            const double dx = xin - x, dy = yin - y;
            double v1 = (double)t00 + ((double)t01 - (double)t00) * dx, v2 = v1;
            if (row2) v2 = (double)t10 + ((double)t11 - (double)t10) * dx;
            v1 = v1 + (v2 - v1) * dy;
            const int m = (int)v1 & 255;
            if (m == 0) continue;
            const int t = (int)canvas[Y * cw + X] * (255 - m) + fg * m + 128;
            canvas[Y * cw + X] = (uint8_t)(((t >> 8) + t) >> 8);
        }
        __syncthreads();
    }

    // B: to the captcha's width
    if (cw > width) {
        sy_resample<true, true>(canvas, cw, 1, cw, SY_H, img, ws, 1, width, coef, cmin, cn);
    } else {
        for (int r = tid >> 6; r < SY_H; r += SY_NT / 64)
            for (int x = tid & 63; x < width; x += 64) img[r * ws + x] = canvas[r * cw + x];
        __syncthreads();
    }

    // C: noise dots and noise arc (same ink everywhere: the order of the stores does not matter)
    for (int i = tid; i < SY_NDOTS * a.nstamp; i += SY_NT) {
        const int d = i / a.nstamp, s = i - d * a.nstamp;
        const int x = p[SY_HDR + 2 * d] + a.stamp[2 * s], y = p[SY_HDR + 2 * d + 1] + a.stamp[2 * s + 1];
        if (x >= 0 && x < width && y >= 0 && y < SY_H) img[y * ws + x] = (uint8_t)fg;
    }
    {
        const int bx0 = p[6], by0 = p[7], bx1 = p[8], by1 = p[9];
        const double start = (double)p[10], end = (double)p[11];
        const double ea = (bx1 - bx0) / 2.0, eb = (by1 - by0) / 2.0, cx = (bx0 + bx1) / 2.0, cy = (by0 + by1) / 2.0;
        const double l0 = sy_dbl(p + 16), l1 = sy_dbl(p + 18), l2 = sy_dbl(p + 20), r0 = sy_dbl(p + 22), r1 = sy_dbl(p + 24), r2 = sy_dbl(p + 26);
        auto on_arc = [&](int ix, int iy) -> bool {
            const double u = ix - cx, v = iy - cy;
            double t = atan2(v / eb, u / ea) * (180.0 / 3.14159265358979323846);
            if (t < 0.0) t += 360.0;
            if (t <= start + 30.0 || t >= 330.0) return l0 * u + l1 * v + l2 >= 0.0;
            if (end - 30.0 <= t && t <= end + 30.0) return r0 * u + r1 * v + r2 >= 0.0;
            return start < t && t < end;
        };
        if (ea > 0.0 && eb > 0.0) {
            const int xa = max(0, bx0), xb = min(width - 1, bx1), ya = max(0, by0), yb = min(SY_H - 1, by1);
            const int ncol = max(0, xb - xa + 1), nrow = max(0, yb - ya + 1);
            for (int i = tid; i < 2 * (ncol + nrow); i += SY_NT) {
                const double sg = (i & 1) ? -1.0 : 1.0;
                const int j = i >> 1;
                if (j < ncol) {                                        // flat stretches: one pixel per column
                    const int x = xa + j;
                    const double u = (x - cx) / ea;
                    if (fabs(u) > 1.0) continue;
                    const double s = sqrt(1.0 - u * u);
                    if (fabs(u) * eb > s * ea) continue;
                    const int yi = (int)floor(cy + sg * eb * s + 0.5);
                    if (yi >= 0 && yi < SY_H && on_arc(x, yi)) img[yi * ws + x] = (uint8_t)fg;
                } else {                                               // steep stretches: one pixel per row
                    const int y = ya + (j - ncol);
                    const double v = (y - cy) / eb;
                    if (fabs(v) > 1.0) continue;
                    const double s = sqrt(1.0 - v * v);
                    if (fabs(v) * ea >= s * eb) continue;
                    const int xi = (int)floor(cx + sg * ea * s + 0.5);
                    if (xi >= 0 && xi < width && on_arc(xi, y)) img[y * ws + xi] = (uint8_t)fg;
                }
            }
        }
    }
    __syncthreads();

    // D: SMOOTH, img -> canvas region [60][ws].  Four pixels per thread: three aligned dwords of each of the three rows instead of 36 byte reads, every
    // product v * k formed once (the same product wherever it is used, so the sums are Filter.c's, term by term in its order).
    {
        const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
        uint8_t* sm = canvas;
        const int wq = ws >> 2;
        for (int y = tid >> 6; y < SY_H; y += SY_NT / 64) {
            const bool edge_row = y == 0 || y == SY_H - 1;
            const uint32_t* rm = (const uint32_t*)(img + (y > 0 ? y - 1 : y) * ws);
            const uint32_t* r0 = (const uint32_t*)(img + y * ws);
            const uint32_t* rp = (const uint32_t*)(img + (y < SY_H - 1 ? y + 1 : y) * ws);
            for (int q = tid & 63; q < wq; q += 64) {
                const uint32_t c0 = r0[q];
                uint32_t outw = c0;
                if (!edge_row) {
                    const int ql = q > 0 ? q - 1 : q, qr = q < wq - 1 ? q + 1 : q;
                    // bytes x0 - 1 .. x0 + 4 of the three rows as floats times k1 (and the centre row's own bytes times k5)
                    float pm[6], pc[6], pp[6], pc5[4];
                    const uint32_t m0 = rm[q], p0 = rp[q];
                    const uint32_t ml = rm[ql], cl = r0[ql], pl = rp[ql], mr = rm[qr], cr = r0[qr], pr = rp[qr];
                    pm[0] = (float)(ml >> 24) * k1; pc[0] = (float)(cl >> 24) * k1; pp[0] = (float)(pl >> 24) * k1;
                    pm[5] = (float)(mr & 255) * k1; pc[5] = (float)(cr & 255) * k1; pp[5] = (float)(pr & 255) * k1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float vm = (float)((m0 >> (8 * j)) & 255), vc = (float)((c0 >> (8 * j)) & 255), vp = (float)((p0 >> (8 * j)) & 255);
                        pm[j + 1] = vm * k1; pc[j + 1] = vc * k1; pp[j + 1] = vp * k1; pc5[j] = vc * k5;
                    }
                    outw = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int x = 4 * q + j;
                        uint32_t o = (c0 >> (8 * j)) & 255;
                        if (x > 0 && x < width - 1) {
                            float ss = 0.5f;
                            ss += pp[j] + pp[j + 1] + pp[j + 2];
                            ss += pc[j] + pc5[j] + pc[j + 2];
                            ss += pm[j] + pm[j + 1] + pm[j + 2];
                            o = ss <= 0.0f ? 0u : ss >= 255.0f ? 255u : (uint32_t)(int)ss;
                        }
                        outw |= o << (8 * j);
                    }
                }
                ((uint32_t*)(sm + y * ws))[q] = outw;
            }
        }
    }
    __syncthreads();

    // E: bilinear to [out_h][nw_out]: rows of the smoothed image first (-> img region, [60][nw_out]), then columns (-> canvas region as the
    // [nw_out][out_h] rows the engine binds: the smoothed image is dead by then), F: 16-byte stores of those rows to HBM, right padding 0
    sy_resample<false, true>(canvas, ws, 1, width, SY_H, img, nw_out, 1, nw_out, coef, cmin, cn);
    sy_resample<false, false>(img, 1, nw_out, SY_H, nw_out, canvas, a.out_h, 1, a.out_h, coef, cmin, cn);
    uint8_t* o = a.out + (long)blockIdx.x * a.W * a.out_h;
    const int nb = nw_out * a.out_h, tb = a.W * a.out_h;
    if (((nb | tb) & 15) == 0 && (((size_t)o) & 15) == 0) {
        for (int i = tid; i < (tb >> 4); i += SY_NT)
            ((u32x4*)o)[i] = i < (nb >> 4) ? ((const u32x4*)canvas)[i] : (u32x4){0u, 0u, 0u, 0u};
    } else {
        for (int i = tid; i < tb; i += SY_NT) o[i] = i < nb ? canvas[i] : (uint8_t)0;
    }
}

// params: [n_images][words_per_image] int32 records of utils/synth.draw_params (max_glyphs glyph slots each); atlas: the concatenated glyph masks;
// stamp: n_stamp (dx, dy) pairs; out: [n_images][W][out_h] uint8.  canvas_cap / width_cap: upper bounds of the records' canvas_w / width
// (they size the LDS image: 60 * (canvas_cap + width_cap) + 34816 bytes <= 160 KB).
extern "C" int ocr_captcha_synth(const int* params, int n_images, int words_per_image, int max_glyphs, const void* atlas, const int* stamp,
                                 int n_stamp, void* out, int W, int out_h, int canvas_cap, int width_cap, void* stream) {
    if (!params || !atlas || !out || (n_stamp && !stamp) || n_images < 0 || max_glyphs < 0 || W <= 0 || out_h <= 0 || out_h > SY_H ||
        words_per_image < SY_HDR + 2 * SY_NDOTS + SY_GW * max_glyphs || canvas_cap < width_cap || width_cap < 8)
        return OCR_ERR_INVALID;
    if (n_images == 0) return OCR_OK;
    const int ccap = (canvas_cap + 15) & ~15, wcap = (width_cap + 15) & ~15;
    const int lds = SY_H * (ccap + wcap) + (SY_TAPS * 256 + 512) * 4;
    if (lds > 160 * 1024) return OCR_ERR_INVALID;
    static int lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute((const void*)captcha_synth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return OCR_ERR_EXEC;
        lds_set = lds;
    }
    SynthArgs a{params, words_per_image, max_glyphs, (const uint8_t*)atlas, stamp, n_stamp, (uint8_t*)out, W, ccap, wcap, out_h};
    captcha_synth_kernel<<<n_images, SY_NT, lds, (hipStream_t)stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
