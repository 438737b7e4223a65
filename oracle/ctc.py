"""ORACLE (test infrastructure): CTC loss/gradient, three ways.

* ``ctc_loss_c``      — the C restatement in oracle/ctc_ref.c (double precision), via ctypes.
* ``ctc_loss_numpy``  — the same recursion in numpy (log space), independent code.
* ``ctc_brute_force`` — exact enumeration of all C**T paths for tiny problems (probability space).

All follow the warp-ctc conventions the reference relies on at lib/networks/network.py:653-654:
activations [T, N, C] unnormalised, flat int labels, blank_label = 0 by default, cost = -log p(l|x),
gradient w.r.t. the unnormalised activations, cost 0 / grad 0 when L + repeats > T.
"""
import ctypes
import itertools
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/ctc_ref.c (gcc) into oracle/_build/libctcref.so."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libctcref.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.ctc_ref_loss.restype = ctypes.c_int
    return _LIB


def ctc_loss_c(act, flat_labels, label_lengths, input_lengths, blank=0, want_grad=True):
    act = np.ascontiguousarray(act, dtype=np.float32)
    T, N, C = act.shape
    labels = np.ascontiguousarray(flat_labels, dtype=np.int32)
    ll = np.ascontiguousarray(label_lengths, dtype=np.int32)
    il = np.ascontiguousarray(input_lengths, dtype=np.int32)
    costs = np.zeros(N, np.float32)
    grad = np.zeros_like(act) if want_grad else None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    rc = _lib().ctc_ref_loss(p(act), p(grad), p(labels), p(ll), p(il), C, N, T, blank, p(costs))
    assert rc == 0
    return costs, grad


def _log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    return x - (m + np.log(np.exp(x - m).sum(axis=-1, keepdims=True)))


def ctc_loss_numpy(act, flat_labels, label_lengths, input_lengths, blank=0):
    act = np.asarray(act, np.float64)
    T, N, C = act.shape
    costs = np.zeros(N)
    grad = np.zeros_like(act)
    off = 0
    for n in range(N):
        L = int(label_lengths[n]); Tn = min(int(input_lengths[n]), T)
        lab = [int(v) for v in flat_labels[off:off + L]]; off += L
        rep = sum(1 for i in range(1, L) if lab[i] == lab[i - 1])
        if L + rep > Tn or Tn <= 0:
            continue
        ext = [blank] * (2 * L + 1); ext[1::2] = lab
        S = len(ext)
        logy = _log_softmax(act[:Tn, n, :])
        a = np.full((Tn, S), -np.inf); b = np.full((Tn, S), -np.inf)
        a[0, 0] = logy[0, ext[0]]
        if S > 1: a[0, 1] = logy[0, ext[1]]
        for t in range(1, Tn):
            for s in range(S):
                c = [a[t - 1, s]]
                if s >= 1: c.append(a[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]: c.append(a[t - 1, s - 2])
                a[t, s] = np.logaddexp.reduce(c) + logy[t, ext[s]]
        tail = [a[Tn - 1, S - 1]] + ([a[Tn - 1, S - 2]] if S > 1 else [])
        logp = np.logaddexp.reduce(tail)
        costs[n] = -logp
        b[Tn - 1, S - 1] = logy[Tn - 1, ext[S - 1]]
        if S > 1: b[Tn - 1, S - 2] = logy[Tn - 1, ext[S - 2]]
        for t in range(Tn - 2, -1, -1):
            for s in range(S):
                c = [b[t + 1, s]]
                if s + 1 < S: c.append(b[t + 1, s + 1])
                if s + 2 < S and ext[s + 2] != blank and ext[s + 2] != ext[s]: c.append(b[t + 1, s + 2])
                b[t, s] = np.logaddexp.reduce(c) + logy[t, ext[s]]
        for t in range(Tn):
            post = np.zeros(C)
            for s in range(S):
                ab = a[t, s] + b[t, s]
                if np.isfinite(ab):
                    post[ext[s]] += np.exp(ab - logy[t, ext[s]] - logp)
            grad[t, n, :] = np.exp(logy[t]) - post
    return costs, grad


def collapse(path, blank=0):
    out, prev = [], None
    for p in path:
        if p != prev and p != blank:
            out.append(p)
        prev = p
    return tuple(out)


def ctc_brute_force(act_tc, label, blank=0):
    """Exact -log p(label | x) for ONE sample, act_tc [T, C]; enumerates C**T paths (tiny T, C only).
    Also returns the dict {labelling: probability} (what an exhaustive beam search must rank)."""
    act_tc = np.asarray(act_tc, np.float64)
    T, C = act_tc.shape
    y = np.exp(_log_softmax(act_tc))
    table = {}
    for path in itertools.product(range(C), repeat=T):
        p = 1.0
        for t, k in enumerate(path):
            p *= y[t, k]
        key = collapse(path, blank)
        table[key] = table.get(key, 0.0) + p
    p = table.get(tuple(int(v) for v in label), 0.0)
    return (-np.log(p) if p > 0 else np.inf), table
