"""Does this torch build's gloo backend all-reduce CUDA tensors (two ranks sharing cuda:0)?  Used to decide how the
N > 1 flow of bench.py can be exercised on a single-GPU box."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def work(rank, world):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29533"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = torch.full((4, 4096), float(rank + 1), dtype=torch.float16, device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    ok = bool((t == 3).all())
    print(f"rank {rank}: gloo all_reduce on a CUDA fp16 tensor -> {'OK' if ok else 'WRONG'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(work, args=(2,), nprocs=2)
