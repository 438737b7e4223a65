// CTC prefix beam search on the device — the decoder the reference actually runs:
//   tf.nn.ctc_beam_search_decoder(logits, seq_len, merge_repeated=True)   (lib/networks/network.py:656, lib/lstm/test.py:30)
// with TF's defaults beam_width = 100, top_paths = 1 and TF's conventions: BLANK = num_classes - 1, per-beam
// (p_blank, p_label) in log space, a child that is already in the beam receives its parent's mass through the
// "parent active" update, children that are not in the beam compete with the current leaves, the `beam_width` best
// totals survive each frame, and merge_repeated collapses adjacent equal labels of the emitted sequence.
// (TF's CPU op streams candidates through a bounded heap; on per-frame log-softmax scores that is exactly "top
// beam_width of {updated leaves} U {new children}", which is what is computed here — ties aside.)
//
// One workgroup (256 threads) per sample; per frame: log-softmax row -> leaf update -> child scores in LDS ->
// exact K-th-largest threshold by 32-step bisection on the order-preserving integer image of the floats -> ordered
// compaction into the next beam -> trie bookkeeping (node pool in a caller-owned workspace).  This path runs at
// validation / test time only; it is latency-, not throughput-critical.
#include "common.h"
#include <math.h>

#define BEAM_MAX 128
#define NEGINF (-INFINITY)

__device__ __forceinline__ float blse(float a, float b) {
    float m = fmaxf(a, b);
    if (m == NEGINF) return NEGINF;
    return m + logf(expf(a - m) + expf(b - m));
}
// order-preserving float -> uint key (larger float <=> larger key; -inf is the smallest finite-ordered key)
__device__ __forceinline__ unsigned fkey(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct BeamArgs {
    const float* act; const int* input_len; int T, N, C, K, merge_repeated, pad_value;
    int* out; int* out_len; float* neg_log_prob;
    int* node_parent; short* node_label; int* node_slot;     // per sample: T*K + 2 entries
};

__global__ __launch_bounds__(256) void ctc_beam_kernel(BeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, K = a.K, blank = C - 1;
    const int Tn = min(a.input_len[n], a.T);
    const int pool = a.T * K + 2;
    int* npar = a.node_parent + (size_t)n * pool;
    short* nlab = a.node_label + (size_t)n * pool;
    int* nslot = a.node_slot + (size_t)n * pool;

    // ---- LDS carve-up
    float* lp = (float*)smem_raw;                         // [C]
    float* cand = lp + ((C + 3) & ~3);                    // [K + K*C]
    short* table = (short*)(cand + K + K * C);            // [K][C] child slot of (parent slot, label) or -1
    int* b_node = (int*)(table + ((K * C + 1) & ~1));     // beam arrays, two generations
    int* b_last = b_node + 2 * BEAM_MAX;
    int* b_ps = b_last + 2 * BEAM_MAX;                    // slot of the parent entry if it is in the beam, else -1
    float* b_pb = (float*)(b_ps + 2 * BEAM_MAX);
    float* b_pnb = b_pb + 2 * BEAM_MAX;
    float* b_tot = b_pnb + 2 * BEAM_MAX;
    int* sel_src = (int*)(b_tot + 2 * BEAM_MAX);          // [BEAM_MAX] candidate index of each new slot
    __shared__ int s_cnt[4];
    __shared__ unsigned s_key;
    __shared__ int s_misc[4];

    // root: empty prefix, p_blank = 1, p_label = 0
    if (tid == 0) {
        b_node[0] = 0; b_last[0] = -1; b_ps[0] = -1; b_pb[0] = 0.f; b_pnb[0] = NEGINF; b_tot[0] = 0.f;
        npar[0] = -1; nlab[0] = -1; nslot[0] = 0;
        s_misc[0] = 1;        // beam size
        s_misc[1] = 1;        // nodes allocated
    }
    for (int i = tid; i < K * C; i += 256) table[i] = -1;
    __syncthreads();
    int gen = 0;
    for (int t = 0; t < Tn; ++t) {
        const int nb = s_misc[0];
        int* node = b_node + gen * BEAM_MAX; int* last = b_last + gen * BEAM_MAX; int* ps = b_ps + gen * BEAM_MAX;
        float* pb = b_pb + gen * BEAM_MAX; float* pnb = b_pnb + gen * BEAM_MAX; float* tot = b_tot + gen * BEAM_MAX;
        // (a) log-softmax of the frame
        const float* row = a.act + ((size_t)t * a.N + n) * C;
        {
            float m = NEGINF;
            for (int k = tid; k < C; k += 256) m = fmaxf(m, row[k]);
            m = wave_max(m);
            if (lane == 0) ((float*)s_cnt)[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(((float*)s_cnt)[0], ((float*)s_cnt)[1]), fmaxf(((float*)s_cnt)[2], ((float*)s_cnt)[3]));
            __syncthreads();
            float sum = 0.f;
            for (int k = tid; k < C; k += 256) sum += expf(row[k] - m);
            sum = wave_sum(sum);
            if (lane == 0) ((float*)s_cnt)[wave] = sum;
            __syncthreads();
            float lse = m + logf(((float*)s_cnt)[0] + ((float*)s_cnt)[1] + ((float*)s_cnt)[2] + ((float*)s_cnt)[3]);
            __syncthreads();
            for (int k = tid; k < C; k += 256) lp[k] = row[k] - lse;
        }
        __syncthreads();
        // (b) leaves stay in the beam with updated probabilities; cand[s] = new total
        float my_npb = NEGINF, my_npnb = NEGINF;
        if (tid < nb) {
            const int s = tid;
            float nl = NEGINF;
            if (last[s] >= 0) {
                nl = pnb[s];
                const int p = ps[s];
                if (p >= 0) nl = blse(nl, (last[s] == last[p]) ? pb[p] : tot[p]);
                if (nl != NEGINF) nl += lp[last[s]];
            }
            my_npnb = nl;
            my_npb = tot[s] + lp[blank];
            cand[s] = blse(my_npb, my_npnb);
        }
        // (c) children that are not in the beam: cand[nb + b*C + c]
        for (int i = tid; i < nb * C; i += 256) {
            const int b = i / C, c = i - b * C;
            float v = NEGINF;
            if (c != blank && table[b * C + c] < 0) {
                const float prev = (c == last[b]) ? pb[b] : tot[b];
                if (prev != NEGINF) v = prev + lp[c];
            }
            cand[nb + i] = v;
        }
        __syncthreads();
        // (d) K-th largest candidate: bisection over the integer image
        const int ncand = nb + nb * C;
        unsigned lo = 0u, hi = 0xffffffffu;                // invariant: count(key >= lo) >= K or lo == 0
        {
            int finite = 0;
            for (int i = tid; i < ncand; i += 256) finite += (cand[i] != NEGINF);
            finite = (int)wave_sum((float)finite);
            if (lane == 0) s_cnt[wave] = finite;
            __syncthreads();
            finite = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            __syncthreads();
            if (finite <= K) { lo = fkey(NEGINF) + 1u; hi = lo; }      // keep every finite candidate
        }
        while (lo < hi) {                                   // find the largest key `lo` with count(key >= lo) >= K
            const unsigned mid = lo + (unsigned)(((unsigned long long)hi - lo + 1ull) >> 1);
            int cnt = 0;
            for (int i = tid; i < ncand; i += 256) cnt += (fkey(cand[i]) >= mid);
            cnt = (int)wave_sum((float)cnt);
            if (lane == 0) s_cnt[wave] = cnt;
            __syncthreads();
            cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            __syncthreads();
            if (cnt >= K) lo = mid; else hi = mid - 1u;
        }
        const unsigned thr = lo;
        // (e) ordered compaction: everything above the threshold, then ties in index order until the beam is full
        if (tid == 0) { s_misc[2] = 0; }
        __syncthreads();
        if (wave == 0) {
            int filled = 0;
            for (int pass = 0; pass < 2; ++pass) {
                for (int base = 0; base < ncand; base += 64) {
                    const int i = base + lane;
                    bool take = false;
                    if (i < ncand) {
                        const unsigned k = fkey(cand[i]);
                        take = (pass == 0) ? (k > thr) : (k == thr && cand[i] != NEGINF);
                    }
                    unsigned long long mask = __ballot(take);
                    int posn = filled + __popcll(mask & ((1ull << lane) - 1ull));
                    if (take && posn < K) sel_src[posn] = i;
                    filled = min(K, filled + __popcll(mask));
                }
            }
            if (lane == 0) s_misc[2] = filled;
        }
        __syncthreads();
        const int nnew = s_misc[2];
        // (f) next generation
        const int g2 = gen ^ 1;
        int* node2 = b_node + g2 * BEAM_MAX; int* last2 = b_last + g2 * BEAM_MAX; int* ps2 = b_ps + g2 * BEAM_MAX;
        float* pb2 = b_pb + g2 * BEAM_MAX; float* pnb2 = b_pnb + g2 * BEAM_MAX; float* tot2 = b_tot + g2 * BEAM_MAX;
        // stash the updated leaf probabilities where the compaction can find them (old arrays are re-read below first)
        float* upd_pb = cand;          // reuse cand[0..nb) AFTER the totals were consumed? totals are still needed -> use table space? no:
        __syncthreads();
        // leaves' new (p_blank, p_label) live in registers of thread s; publish them through the dead half of the beam arrays
        if (tid < nb) { pb2[tid] = my_npb; pnb2[tid] = my_npnb; }     // temporarily indexed by OLD slot
        if (tid < nb) nslot[node[tid]] = -1;                           // everything is out of the beam until re-registered
        __syncthreads();
        float r_pb = NEGINF, r_pnb = NEGINF, r_tot = NEGINF; int r_node = -1, r_last = -1;
        if (tid < nnew) {
            const int i = sel_src[tid];
            r_tot = cand[i];
            if (i < nb) {                                   // surviving leaf
                r_node = node[i]; r_last = last[i]; r_pb = pb2[i]; r_pnb = pnb2[i];
            } else {                                        // new child of leaf b with label c
                const int b = (i - nb) / C, c = (i - nb) - b * C;
                const int id = atomicAdd(&s_misc[1], 1);
                npar[id] = node[b]; nlab[id] = (short)c;
                r_node = id; r_last = c; r_pb = NEGINF; r_pnb = r_tot;
            }
        }
        __syncthreads();
        if (tid < nnew) {
            node2[tid] = r_node; last2[tid] = r_last; pb2[tid] = r_pb; pnb2[tid] = r_pnb; tot2[tid] = r_tot;
            nslot[r_node] = tid;
        }
        for (int i = tid; i < K * C; i += 256) table[i] = -1;
        __syncthreads();
        if (tid < nnew) {
            const int par = npar[node2[tid]];
            const int p = (par >= 0) ? nslot[par] : -1;
            ps2[tid] = p;
            if (p >= 0) table[p * C + last2[tid]] = (short)tid;
        }
        if (tid == 0) s_misc[0] = nnew;
        __syncthreads();
        gen = g2;
        (void)upd_pb;
    }
    // ---- best leaf, label sequence (root-ward walk), merge_repeated, output
    if (tid == 0) {
        const int nb = s_misc[0];
        float* tot = b_tot + gen * BEAM_MAX; int* node = b_node + gen * BEAM_MAX;
        int best = 0;
        for (int s = 1; s < nb; ++s) if (tot[s] > tot[best]) best = s;
        int* o = a.out + (size_t)n * a.T;
        int len = 0;
        for (int id = node[best]; id > 0; id = npar[id]) o[len++] = nlab[id];      // reversed
        for (int i = 0; i < len / 2; ++i) { int tmp = o[i]; o[i] = o[len - 1 - i]; o[len - 1 - i] = tmp; }
        if (a.merge_repeated) {
            int w = 0;
            for (int i = 0; i < len; ++i) if (i == 0 || o[i] != o[i - 1]) o[w++] = o[i];
            len = w;
        }
        for (int i = len; i < a.T; ++i) o[i] = a.pad_value;
        a.out_len[n] = len;
        if (a.neg_log_prob) a.neg_log_prob[n] = -tot[best];
    }
}

static size_t beam_lds_bytes(int C, int K) {
    size_t b = (size_t)((C + 3) & ~3) * 4 + (size_t)(K + K * C) * 4 + (size_t)((K * C + 1) & ~1) * 2 + (size_t)BEAM_MAX * 2 * 6 * 4 +
               (size_t)BEAM_MAX * 4;
    return (b + 15) & ~(size_t)15;
}

extern "C" int ocr_ctc_beam_workspace_size(int alphabet_size, int minibatch, int max_time, int beam_width, size_t* bytes) {
    if (!bytes || alphabet_size < 2 || minibatch <= 0 || max_time <= 0 || beam_width <= 0 || beam_width > BEAM_MAX)
        return OCR_ERR_INVALID;
    if (beam_lds_bytes(alphabet_size, beam_width) > 160 * 1024) return OCR_ERR_INVALID;
    size_t pool = (size_t)max_time * beam_width + 2;
    *bytes = (size_t)minibatch * pool * (sizeof(int) * 2 + sizeof(short));
    *bytes = (*bytes + 255) & ~(size_t)255;
    return OCR_OK;
}

extern "C" int ocr_ctc_beam_decode(const float* activations, const int* input_lengths, int alphabet_size, int minibatch,
                                   int max_time, int beam_width, int merge_repeated, int pad_value, int* decoded,
                                   int* decoded_lengths, float* neg_log_prob, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    size_t need = 0;
    if (!activations || !input_lengths || !decoded || !decoded_lengths || !workspace) return OCR_ERR_INVALID;
    if (ocr_ctc_beam_workspace_size(alphabet_size, minibatch, max_time, beam_width, &need) != OCR_OK || workspace_bytes < need)
        return OCR_ERR_INVALID;
    const size_t pool = (size_t)max_time * beam_width + 2;
    BeamArgs a = {activations, input_lengths, max_time, minibatch, alphabet_size, beam_width, merge_repeated, pad_value,
                  decoded, decoded_lengths, neg_log_prob, nullptr, nullptr, nullptr};
    a.node_parent = (int*)workspace;
    a.node_slot = a.node_parent + (size_t)minibatch * pool;
    a.node_label = (short*)(a.node_slot + (size_t)minibatch * pool);
    const size_t lds = beam_lds_bytes(alphabet_size, beam_width);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute((const void*)ctc_beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return OCR_ERR_EXEC;
        lds_set = lds;
    }
    ctc_beam_kernel<<<minibatch, 256, lds, stream>>>(a);
    OCR_CHECK_LAUNCH();
    return OCR_OK;
}
