"""ORACLE (test infrastructure): CTC decoders with the semantics the reference observes.

* ``greedy_decode``: per-frame argmax, merge repeats, drop blank (0) — the best-path labelling.
* ``beam_search_tf``: tf.nn.ctc_beam_search_decoder(inputs, seq_len, beam_width=100, top_paths=1,
  merge_repeated=True) as called at reference lib/networks/network.py:656 and lib/lstm/test.py:30 — prefix beam
  search with BLANK = C-1 (TF convention), followed by sparse_to_dense(default 0) at network.py:657.
  TF is not installable here; the implementation follows the published algorithm (Graves 2012 / Hannun 2014,
  TF core/util/ctc/ctc_beam_search.h: per-beam (p_blank, p_nonblank), merge_repeated collapsing adjacent equal
  labels of the emitted sequence) and is pinned on brute-force enumeration for tiny problems.
"""
import numpy as np


def greedy_decode(logits_tnc, seq_len, blank=0):
    T, N, C = logits_tnc.shape
    out = []
    for n in range(N):
        am = np.argmax(logits_tnc[:seq_len[n], n, :], axis=-1)
        seq, prev = [], -1
        for a in am:
            a = int(a)
            if a != blank and a != prev:
                seq.append(a)
            prev = a
        out.append(seq)
    return out


def dense(seqs, pad=0):
    m = max([len(s) for s in seqs] + [0])
    arr = np.full((len(seqs), m), pad, np.int32)
    for i, s in enumerate(seqs):
        arr[i, :len(s)] = s
    return arr


def _logsumexp(*xs):
    m = max(xs)
    if m == -np.inf:
        return -np.inf
    return m + np.log(sum(np.exp(x - m) for x in xs))


def beam_search_tf(logits_tnc, seq_len, beam_width=100, merge_repeated=True, blank=None, top_paths=1):
    """Returns (list of label lists, list of log-probabilities) for the top path per sample; with top_paths > 1 every
    entry is a list of the `top_paths` best (labels / log-probabilities) in rank order, as TF's decoded[k] outputs."""
    T, N, C = logits_tnc.shape
    if blank is None:
        blank = C - 1
    results, scores = [], []
    for n in range(N):
        x = np.asarray(logits_tnc[:seq_len[n], n, :], np.float64)
        m = x.max(axis=-1, keepdims=True)
        logp = x - (m + np.log(np.exp(x - m).sum(-1, keepdims=True)))
        beams = {(): (0.0, -np.inf)}            # prefix -> (log p_blank, log p_nonblank)
        for t in range(x.shape[0]):
            nxt = {}

            def add(prefix, pb, pnb):
                ob, onb = nxt.get(prefix, (-np.inf, -np.inf))
                nxt[prefix] = (_logsumexp(ob, pb), _logsumexp(onb, pnb))

            for prefix, (pb, pnb) in beams.items():
                tot = _logsumexp(pb, pnb)
                add(prefix, tot + logp[t, blank], -np.inf)            # emit blank
                for c in range(C):
                    if c == blank:
                        continue
                    lp = logp[t, c]
                    if prefix and prefix[-1] == c:
                        add(prefix, -np.inf, pnb + lp)                 # repeat, no blank in between: same prefix
                        add(prefix + (c,), -np.inf, pb + lp)           # after a blank: extends
                    else:
                        add(prefix + (c,), -np.inf, tot + lp)
            ranked = sorted(nxt.items(), key=lambda kv: -_logsumexp(*kv[1]))[:beam_width]
            beams = dict(ranked)
        ranked = sorted(beams.items(), key=lambda kv: -_logsumexp(*kv[1]))[:max(1, top_paths)]
        seqs, scs = [], []
        for best, (pb, pnb) in ranked:
            seq = list(best)
            if merge_repeated:
                merged, prev = [], None
                for s in seq:
                    if s != prev:
                        merged.append(s)
                    prev = s
                seq = merged
            seqs.append(seq)
            scs.append(_logsumexp(pb, pnb))
        results.append(seqs[0] if top_paths == 1 else seqs)
        scores.append(scs[0] if top_paths == 1 else scs)
    return results, scores


def reference_decode(logits_tnc, seq_len, beam_width=100):
    """What the reference's dense_decoded holds after its own post-processing: beam search with blank = C-1,
    densified with 0, then zeros stripped by accuracy_calculation / decodeRes (training.py:32, test.py:80)."""
    seqs, _ = beam_search_tf(logits_tnc, seq_len, beam_width=beam_width, merge_repeated=True)
    return [[v for v in s if v != 0] for s in seqs]
