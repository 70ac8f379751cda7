"""Shared helpers for the GPU parity tests."""
import numpy as np
import torch


def dev(x, device="cuda:0"):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def f16_bits(t):
    return t.detach().cpu().numpy().view(np.uint16)


def ulp_diff_f16(a, b):
    """Distance in fp16 ulps between two float16 numpy arrays (monotone integer mapping)."""
    def key(x):
        u = x.view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7FFF), 0x8000 + u)   # -0 and +0 map to the same key
    return np.abs(key(np.ascontiguousarray(a)) - key(np.ascontiguousarray(b)))


def unpack_qweight_torch(qweight):
    """Reference packed int8 [N,K/2] -> uint8 [N,K] on the GPU (inverse of w4a8_linear.py:196-226)."""
    N, K2 = qweight.shape
    K = K2 * 2
    p = qweight.view(torch.uint8).reshape(N // 32, K // 32, 8, 4, 2, 2, 4)   # n32 k32 c e d b f
    lo = (p & 0xF).permute(0, 5, 2, 1, 4, 3, 6)                             # n32 b c k32 d e f
    hi = (p >> 4).permute(0, 5, 2, 1, 4, 3, 6)
    out = torch.stack([lo, hi], dim=1)                                       # n32 a b c k32 d e f
    return out.reshape(N, K)


def int_matmul_torch(A, W):
    """Exact int64 A[M,K] @ W[N,K]^T on the GPU via fp32 slices (|partial sums| < 2**24)."""
    M, K = A.shape
    acc = torch.zeros((M, W.shape[0]), dtype=torch.int64, device=A.device)
    step = 1000
    for k0 in range(0, K, step):
        acc += (A[:, k0:k0 + step].float() @ W[:, k0:k0 + step].float().T).round().to(torch.int64)
    return acc
