/* ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline may use it;
 * the product path never links or calls this file).
 *
 * CPU restatement of the CTC loss/gradient the reference obtains from baidu warp-ctc through
 * `warpctc_tensorflow.ctc(activations, flat_labels, label_lengths, input_lengths)` with the default
 * blank_label = 0 (reference lib/networks/network.py:653-654).  warp-ctc is NOT vendored in the reference
 * (README.md:20 points at upstream master, unpinned), so this follows its published algorithm
 * (Graves et al. 2006, eq. 6-16; warp-ctc include/detail/cpu_ctc.h: softmax with max subtraction, log-space
 * alpha/beta over the blank-extended label, cost 0 / zero gradient when L + repeats > T, gradient w.r.t. the
 * UNNORMALISED activations = softmax - posterior).  PARITY PIN: warp-ctc's own known-answer vector
 * (tests/test_cpu.cpp small_test: cost 2.46286, grads 0.177031 / -0.708125), brute-force path enumeration and
 * torch.nn.functional.ctc_loss — see tests/test_oracle_ctc.py.  All arithmetic in double.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double lse2(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

/* activations: [T][N][C] float; grads (may be NULL): [T][N][C] float; returns 0 on success */
int ctc_ref_loss(const float* act, float* grad, const int* flat_labels, const int* label_lengths,
                 const int* input_lengths, int C, int N, int T, int blank, float* costs) {
    int off = 0;
    for (int n = 0; n < N; ++n) {
        const int L = label_lengths[n];
        const int Tn = input_lengths[n] < T ? input_lengths[n] : T;
        const int* lab = flat_labels + off;
        off += L;
        const int S = 2 * L + 1;
        if (grad)
            for (int t = 0; t < T; ++t) memset(grad + ((size_t)t * N + n) * C, 0, sizeof(float) * C);
        int repeats = 0;
        for (int i = 1; i < L; ++i) repeats += (lab[i] == lab[i - 1]);
        if (L + repeats > Tn || Tn <= 0) { costs[n] = 0.f; continue; }

        int* ext = (int*)malloc(sizeof(int) * S);
        for (int s = 0; s < S; ++s) ext[s] = (s & 1) ? lab[s >> 1] : blank;
        double* logy = (double*)malloc(sizeof(double) * Tn * C);     /* log softmax */
        for (int t = 0; t < Tn; ++t) {
            const float* row = act + ((size_t)t * N + n) * C;
            double m = row[0];
            for (int k = 1; k < C; ++k) if (row[k] > m) m = row[k];
            double sum = 0;
            for (int k = 0; k < C; ++k) sum += exp((double)row[k] - m);
            double l = m + log(sum);
            for (int k = 0; k < C; ++k) logy[t * C + k] = (double)row[k] - l;
        }
        double* alpha = (double*)malloc(sizeof(double) * Tn * S);
        double* beta = (double*)malloc(sizeof(double) * Tn * S);
        for (int i = 0; i < Tn * S; ++i) { alpha[i] = -INFINITY; beta[i] = -INFINITY; }
        alpha[0] = logy[ext[0]];
        if (S > 1) alpha[1] = logy[ext[1]];
        for (int t = 1; t < Tn; ++t)
            for (int s = 0; s < S; ++s) {
                double a = alpha[(t - 1) * S + s];
                if (s >= 1) a = lse2(a, alpha[(t - 1) * S + s - 1]);
                if (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) a = lse2(a, alpha[(t - 1) * S + s - 2]);
                if (a != -INFINITY) a += logy[t * C + ext[s]];
                alpha[t * S + s] = a;
            }
        double logp = alpha[(Tn - 1) * S + S - 1];
        if (S > 1) logp = lse2(logp, alpha[(Tn - 1) * S + S - 2]);
        costs[n] = (float)(-logp);
        if (grad) {
            beta[(Tn - 1) * S + S - 1] = logy[(Tn - 1) * C + ext[S - 1]];
            if (S > 1) beta[(Tn - 1) * S + S - 2] = logy[(Tn - 1) * C + ext[S - 2]];
            for (int t = Tn - 2; t >= 0; --t)
                for (int s = 0; s < S; ++s) {
                    double b = beta[(t + 1) * S + s];
                    if (s + 1 < S) b = lse2(b, beta[(t + 1) * S + s + 1]);
                    if (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s]) b = lse2(b, beta[(t + 1) * S + s + 2]);
                    if (b != -INFINITY) b += logy[t * C + ext[s]];
                    beta[t * S + s] = b;
                }
            double* post = (double*)malloc(sizeof(double) * C);
            for (int t = 0; t < Tn; ++t) {
                for (int k = 0; k < C; ++k) post[k] = 0;
                for (int s = 0; s < S; ++s) {
                    double ab = alpha[t * S + s] + beta[t * S + s];
                    if (ab == -INFINITY) continue;
                    post[ext[s]] += exp(ab - logy[t * C + ext[s]] - logp);
                }
                float* g = grad + ((size_t)t * N + n) * C;
                for (int k = 0; k < C; ++k) g[k] = (float)(exp(logy[t * C + k]) - post[k]);
            }
            free(post);
        }
        free(ext); free(logy); free(alpha); free(beta);
    }
    return 0;
}
