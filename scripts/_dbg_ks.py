import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from qserve_amd import _lib
from oracle import w4a8, synth

import qserve_backend.qgemm_w4a8_per_chn as op
gpu = torch.device("cuda:0")
for variant, M, N, K in ((4211, 16, 64, 2048), (4221, 32, 64, 2048), (4222, 32, 128, 2048), (4241, 64, 64, 2048)):
    pr = synth.per_channel_problem(M, N, K, seed=M + N + K)
    acc_ref, out_ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
    A = torch.from_numpy(pr["A"]).to(gpu); W = torch.from_numpy(pr["qweight"]).to(gpu)
    # per-slice references
    half = K // 2
    for rep in range(3):
        _lib.lib.qs_set_gemm_variant(variant)
        acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
        op.gemm_forward_acc(A, W, acc)
        _lib.lib.qs_set_gemm_variant(-1)
        got = acc.cpu().numpy()
        bad = got != acc_ref
        print(variant, "rep", rep, "mismatches", bad.sum(), "of", bad.size, "bad cols", np.unique(np.nonzero(bad)[1])[:40], "bad rows", np.unique(np.nonzero(bad)[0])[:20])
