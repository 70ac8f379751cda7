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


def pack_qweight_torch(q):
    """uint4 values q [N,K] (torch, any device) -> reference packed int8 [N,K/2]: the torch twin of
    oracle.w4a8.pack_qweight (w4a8_linear.py:196-226), for problem sizes where numpy packing would take minutes.
    tests/test_gemm_gpu.py cross-checks it against the oracle packer."""
    N, K = q.shape
    w = q.to(torch.uint8).reshape(N // 32, 2, 2, 8, K // 32, 2, 4, 4)        # n32 a b c k32 d e f
    w = w.permute(0, 4, 3, 6, 5, 2, 7, 1)                                     # n32 k32 c e d b f a
    return ((w[..., 1] << 4) | w[..., 0]).contiguous().reshape(N, K // 2).view(torch.int8)


def permute_group_meta_torch(x):
    """[K/G, N] natural channel order -> the reference's per-32 storage order (oracle.w4a8.permute_group_meta)."""
    ng, N = x.shape
    return x.reshape(ng, N // 32, 4, 8).permute(0, 1, 3, 2).contiguous().reshape(ng, N)


def per_group_problem_torch(M, N, K, device, seed=0):
    """QoQ-style per-group problem inside the protective range, generated on the device (same recipe as
    oracle.synth.per_group_problem(valid=True)).  Returns tensors + the dequantised int8 weights w8 [N,K]."""
    g = torch.Generator(device=device).manual_seed(seed)
    ng = K // 128
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=device, generator=g)
    w8 = torch.randint(-119, 120, (N, ng, 128), device=device, generator=g, dtype=torch.int32)
    mx, mn = w8.amax(dim=2), w8.amin(dim=2)
    s2 = torch.clamp(torch.ceil((mx - mn).float() / 15.0), min=1).to(torch.int32)
    z = torch.clamp(torch.round(-mn.float() / s2.float()), 0, 15).to(torch.int32)
    q = torch.clamp(torch.round(w8.float() / s2[..., None].float()) + z[..., None], 0, 15).to(torch.int32)
    lo = torch.clamp(torch.ceil(-128.0 / s2.float() + z.float()), 0, 15).to(torch.int32)
    hi = torch.minimum(torch.clamp(torch.floor(127.0 / s2.float() + z.float()), 0, 15).to(torch.int32), 255 // s2)
    q = torch.minimum(torch.maximum(q, lo[..., None]), hi[..., None])
    deq = (q - z[..., None]) * s2[..., None]
    assert int((q * s2[..., None]).max()) <= 255 and int(deq.min()) >= -128 and int(deq.max()) <= 127
    qweight = pack_qweight_torch(q.reshape(N, K))
    s2_scales = permute_group_meta_torch(s2.t().contiguous()).to(torch.int8)
    s2_zeros = permute_group_meta_torch((-z * s2).t().contiguous()).to(torch.int8)     # wraps mod 256 like the packer
    wscales = (torch.rand((N,), device=device, generator=g) * 0.018 + 0.002).half()
    ascales = (torch.rand((M,), device=device, generator=g) * 0.045 + 0.005).half()
    return dict(A=A, qweight=qweight, s2_scales=s2_scales, s2_zeros=s2_zeros, wscales=wscales, ascales=ascales,
                w8=deq.reshape(N, K).to(torch.int8))
