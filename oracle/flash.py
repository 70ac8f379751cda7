"""Oracle for the prefill attention provider (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

flash-attn v2.5.8 is an un-vendored dependency of the reference (README.md:77,104; not under /root/reference), so this
restates its published semantics: per sequence and head, softmax(scale * Q K^T + causal mask) V with the mask aligned to
the bottom-right corner (key j visible to query i iff j <= i + len_k - len_q), everything in float64 on the fp16 inputs.
parity unpinned (the reference holds no test or vector for this call); the GPU bar is 2e-3 absolute on fp16 outputs."""
import numpy as np


def attention_varlen(q, k, v, cu_q, cu_k, scale=None, causal=True):
    """q [Tq, H, D], k/v [Tk, Hkv, D] float16; cu_* int32 [B+1]  ->  float32 [Tq, H, D]."""
    Tq, H, D = q.shape
    Hkv = k.shape[1]
    G = H // Hkv
    scale = 1.0 / np.sqrt(D) if scale is None else scale
    out = np.zeros((Tq, H, D), np.float32)
    for b in range(len(cu_q) - 1):
        qs, qe, ks, ke = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        lq, lk = qe - qs, ke - ks
        if lq == 0:
            continue
        for h in range(H):
            Q = q[qs:qe, h].astype(np.float64)
            K = k[ks:ke, h // G].astype(np.float64)
            V = v[ks:ke, h // G].astype(np.float64)
            S = (Q @ K.T) * scale
            if causal:
                i = np.arange(lq)[:, None]
                j = np.arange(lk)[None, :]
                S = np.where(j <= i + (lk - lq), S, -np.inf)
            m = S.max(axis=1, keepdims=True) if lk > 0 else np.zeros((lq, 1))
            m = np.where(np.isfinite(m), m, 0.0)
            P = np.exp(S - m)
            l = P.sum(axis=1, keepdims=True)
            O = np.where(l > 0, (P @ V) / np.where(l > 0, l, 1.0), 0.0)
            out[qs:qe, h] = O.astype(np.float32)
    return out


def attention_rows(q, k, v, cu_q, cu_k, rows, heads, scale=None, causal=True):
    """The same definition for SAMPLED (global query row, head) pairs only: float32 [len(rows), len(heads), D].  For sequences
    whose full score matrix would not fit (BASELINE config 5: 8 k tokens)."""
    D = q.shape[2]
    G = q.shape[1] // k.shape[1]
    scale = 1.0 / np.sqrt(D) if scale is None else scale
    out = np.zeros((len(rows), len(heads), D), np.float32)
    cu_q = np.asarray(cu_q); cu_k = np.asarray(cu_k)
    for ri, t in enumerate(rows):
        b = int(np.searchsorted(cu_q, t, side="right") - 1)
        i = t - int(cu_q[b])
        lq, ks, ke = int(cu_q[b + 1] - cu_q[b]), int(cu_k[b]), int(cu_k[b + 1])
        lk = ke - ks
        nvis = min(lk, i + (lk - lq) + 1) if causal else lk
        for hi, h in enumerate(heads):
            if nvis <= 0:
                continue
            K = k[ks:ks + nvis, h // G].astype(np.float64)
            V = v[ks:ks + nvis, h // G].astype(np.float64)
            sc = (K @ q[t, h].astype(np.float64)) * scale
            p = np.exp(sc - sc.max())
            out[ri, hi] = ((p @ V) / p.sum()).astype(np.float32)
    return out
