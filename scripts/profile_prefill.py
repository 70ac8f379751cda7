#!/usr/bin/env python3
"""One prompt phase (64 x 1024 tokens, Llama-3-8B shapes) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qserve_amd import decode as D
eng = D.DecodeEngine(D.LLAMA3_8B, 64, 1024, 8)
for _ in range(2):
    eng.prefill(1024)
torch.cuda.synchronize()
print("prefill done")
