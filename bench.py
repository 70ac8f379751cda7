#!/usr/bin/env python3
"""bench.py -- decode throughput of the W4A8KV4 hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one decode step of a Llama-3-8B-shaped W4A8KV4 model (BASELINE.json configs[1]: per-channel W4A8, KV4,
bs=64, context 1024 -> +512) over synthetic random-quantised weights and a cache written by the prefill writer:
32 layers x (4 W4A8 GEMMs + paged KV4 attention + the 5 activation-side kernels) + final norm + fp16 lm_head +
greedy sampling, replayed from a hipGraph.  N > 1 = tensor parallel over N GPUs (column/row shards, 2 RCCL
all-reduces per layer), strong scaling (the batch and the model are fixed).

Prints ONE JSON line on rank 0 with the contract's keys plus `roofline` (dominant kernel, timed live with HIP events
on the launch stream) and `cpu_baseline` (the numpy oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--max-new", type=int, default=512)
    ap.add_argument("--group-size", type=int, default=-1, choices=[-1, 128])
    ap.add_argument("--kv8", action="store_true")
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama2-7b", "llama2-70b", "qwen1.5-72b", "tiny"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-bench", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the prompt-phase measurement")
    ap.add_argument("--op-by-op", action="store_true",
                    help="issue the reference's ops one by one (no fused pairs) in the timed step")
    ap.add_argument("--gemm-variant", type=int, default=-1, help="A/B: qs_set_gemm_variant code (include/qserve_amd.h)")
    ap.add_argument("--tp-full-graph", action="store_true",
                    help="N>1: capture the all-reduces into the step's hipGraph as well (default: one graph per segment "
                         "between the collectives, collectives issued eagerly - independent of capture support in RCCL)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1 (tensor parallel): weak = --batch sequences PER GPU (global batch = batch x N, so every "
                         "rank keeps N=1's GEMM MACs and KV bytes); strong = --batch is the global batch")
    return ap.parse_args()


def gemm_bytes(M, N, K, group):
    meta = 4 * N + 4 * M if group == -1 else 2 * (K // 128) * N + 2 * N + 2 * M
    return M * K + N * K // 2 + 2 * M * N + meta


def attn_bytes(B, H, Hkv, L, int4):
    per_tok = Hkv * (2 * 128 * (4 if int4 else 8) // 8 + 8)
    return B * (L * per_tok + 2 * H * 128 * 2)             # KV bytes counted once + q/out


def time_kernel(fn, reps, torch):
    """Average GPU duration (us) of one launch: `reps` launches issued by fn(i) are captured into a hipGraph (so no
    host time sits between them, exactly as in the timed decode step) and the replays are bracketed by HIP events
    recorded on the stream the graph - and therefore every kernel - runs on."""
    for i in range(min(2, reps)):
        fn(i)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(reps):
                fn(i)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(4):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * reps)


def kernel_bench(eng, torch):
    """Per-kernel timing of the hot-path kernels at the step's own shapes and data (rotating over the layers so the
    256 MiB Infinity Cache cannot serve the 109 MB of weights per layer)."""
    import qserve_backend.fused_attention as fa
    B, nl = eng.B, len(eng.layers)
    res = []
    qa, qo = eng.q_act, eng.q_attn
    specs = [("qkv", qa, eng.qkv_buf), ("o", qo, eng.proj_out), ("gate_up", qa, eng.gate_up_buf),
             ("down", eng.q_mlp, eng.proj_out)]
    for name, x, out in specs:
        lin0 = eng.layers[0][name]
        us = time_kernel(lambda i: eng.layers[i % nl][name](x, eng.q_scale, eng.q_sum, out), 4 * nl, torch)
        by = gemm_bytes(B, lin0.n, lin0.k, eng.group_size)
        res.append(dict(kernel=f"w4a8_gemm[{name} M={B} N={lin0.n} K={lin0.k}]", us=us, bytes=by,
                        gbs=by / us / 1e3, tops=2.0 * B * lin0.n * lin0.k / us / 1e6, per_step=nl))
    L = int(eng.lengths.max().item())
    q, k, v = eng.qkv_buf.split([eng.H * 128, eng.Hkv * 128, eng.Hkv * 128], dim=-1)
    q, k, v = q.reshape(B, eng.H, 128), k.reshape(B, eng.Hkv, 128), v.reshape(B, eng.Hkv, 128)
    eng.qkv_buf.normal_()

    def attn(i):
        fa.single_query_attention(q, k, v, eng.tables[i % nl], eng.lengths, None, 8192, 64, eng.size_per_token,
                                  eng.max_len, 128, eng.cfg["rope_theta"], True, eng.int4, True)
    us = time_kernel(attn, 4 * nl, torch)
    by = attn_bytes(B, eng.H, eng.Hkv, L - 1, eng.int4)
    res.append(dict(kernel=f"decode_attention[B={B} H={eng.H} Hkv={eng.Hkv} L={L}]", us=us, bytes=by,
                    gbs=by / us / 1e3, per_step=nl))
    return res


def cpu_baseline(args, cfg):
    """The numpy oracle ("port" of the reference algorithm) on the host cores, on a bounded sample of the step: the
    four per-channel GEMMs of a layer at M = batch, repeated over fresh weights for about 10 s, plus decode attention
    for 16 sequences of the batch; extrapolated to tokens/s of a whole step (layers x (GEMMs + batch sequences))."""
    import numpy as np
    from oracle import kvattn, synth, w4a8
    B = args.batch
    H, Hkv, hid, inter = cfg["heads"], cfg["kv_heads"], cfg["hidden"], cfg["inter"]
    shapes = [((H + 2 * Hkv) * 128, hid), (hid, hid), (2 * inter, hid), (hid, inter)]
    rng = np.random.default_rng(0)
    t_gemm, reps = 0.0, 0
    while reps < 2 or (t_gemm < 10.0 and reps < 64):
        for N, K in shapes:
            A = rng.integers(-127, 128, (B, K), dtype=np.int8)
            qw = rng.integers(-128, 128, (N, K // 2), dtype=np.int8)
            ws = rng.uniform(0.002, 0.02, N).astype(np.float16)
            sa = rng.uniform(0.005, 0.05, B).astype(np.float16)
            t0 = time.perf_counter()
            w4a8.gemm_per_chn(A, qw, ws, sa, ws, sa)
            t_gemm += time.perf_counter() - t0
        reps += 1
    t_layer = t_gemm / reps
    nseq, L = 16, args.prompt_len + 1
    pr = synth.attention_problem(nseq, H, Hkv, [L] * nseq, seed=1)
    pool = kvattn.PagePool(pr["nblocks"], Hkv, 128, not args.kv8)
    pool.k[:] = rng.integers(0, 256, pool.k.shape, dtype=np.uint8)
    pool.v[:] = rng.integers(0, 256, pool.v.shape, dtype=np.uint8)
    for p in (pool.k, pool.v):   # sane fp16 scales / zeros
        meta = p[:, pool.scale_off:].view(np.float16)
        meta[:] = np.float16(0.25)
    t0 = time.perf_counter()
    kvattn.decode_attention(pr["q"], pr["k"], pr["v"], pr["tables"], pr["lengths"], pool, cfg["rope_theta"], "fp32")
    t_attn_all = time.perf_counter() - t0
    t_attn = t_attn_all / nseq
    step_s = cfg["layers"] * (t_layer + B * t_attn)
    threads = os.cpu_count()
    try:                                   # threads numpy's BLAS actually uses for the GEMM slices (the rest is 1 thread)
        from threadpoolctl import threadpool_info
        blas = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            threads = max(blas)
    except Exception:
        pass
    return dict(value=B / step_s, unit="tokens/s", cores=threads, kind="port",
                sample=f"numpy oracle (GEMM slices on numpy's BLAS threads = `cores`, everything else single-threaded), "
                       f"{t_gemm + t_attn_all:.1f} s of CPU work: one layer's per-channel GEMMs (M={B}) "
                       f"x {reps} weight sets = {t_layer:.2f} s per layer; decode attention {t_attn:.3f} s/sequence at "
                       f"L={L} ({nseq} sequences timed); step = {cfg['layers']} layers x (GEMMs + {B} sequences)")


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    per_gpu_batch = args.batch
    if world > 1 and args.scaling == "weak":
        args.batch *= world          # global batch; the TP shards (heads / N / K splits) divide the work back
    # functional runs of the N > 1 flow on a single-GPU box (tests/test_bench_tp_gpu.py): QS_DIST_BACKEND=gloo (gloo
    # all-reduces CUDA tensors through the host) and QS_DIST_DEVICE=0 (every rank on cuda:0).  Never set by the driver.
    backend = os.environ.get("QS_DIST_BACKEND", "nccl")
    dev_index = int(os.environ["QS_DIST_DEVICE"]) if "QS_DIST_DEVICE" in os.environ else (local_rank if world > 1 else 0)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = f"cuda:{dev_index}"

    from qserve_amd import build
    if rank == 0:
        build.build(verbose=False)
    if world > 1:
        dist.barrier()
    from qserve_amd import decode as D
    if args.gemm_variant != -1:
        from qserve_amd import _lib
        _lib.lib.qs_set_gemm_variant(args.gemm_variant)
    cfg = {"llama3-8b": D.LLAMA3_8B, "llama2-7b": D.LLAMA2_7B, "llama2-70b": D.LLAMA2_70B, "qwen1.5-72b": D.QWEN15_72B,
           "tiny": D.TINY}[args.model]
    eng = D.DecodeEngine(cfg, args.batch, args.prompt_len, args.max_new, group_size=args.group_size,
                         int4_kv=not args.kv8, device=dev, tp_rank=rank, tp_world=world,
                         fuse_pairs=not args.op_by_op)
    eng.prefill_cache(args.prompt_len)
    graphed = False
    if not args.no_graph:
        try:
            eng.capture(piecewise=False if args.tp_full_graph else None)
            graphed = True
        except Exception as e:   # e.g. a collective refusing capture: fall back to eager launches
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {str(e).splitlines()[0]}); running eagerly",
                      file=sys.stderr)
            eng.graph = None
            eng.pieces = None
            # the invalidated capture leaves a stale HIP error behind that the next checked call would re-raise:
            # drain it (measured with the gloo functional run, tests/test_bench_tp_gpu.py) before touching the engine
            for _ in range(4):
                try:
                    torch.cuda.synchronize()
                    eng.lengths.fill_(args.prompt_len + 1)
                    torch.cuda.synchronize()
                    break
                except Exception:
                    continue
    steps2 = 0 if args.op_by_op else min(args.steps, 32)      # secondary timing of the op-by-op sequence
    assert args.warmup + args.steps + steps2 + 8 <= args.max_new, "steps exceed the page budget (prompt_len + max_new)"

    for _ in range(args.warmup):
        eng.run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3

    # the same model with every reference op issued on its own (fused pairs off): reported next to the headline value
    ms_op = None
    if steps2 and graphed and world == 1:
        eng.fuse_pairs = False
        eng.capture()
        for _ in range(2):
            eng.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps2):
            eng.run()
        torch.cuda.synchronize()
        ms_op = (time.perf_counter() - t0) / steps2 * 1e3
        eng.fuse_pairs = True

    # the prompt phase through the same library (W4A8 GEMMs at M = batch*prompt_len, prefill KV writer, causal flash
    # attention): prompt tokens/s, and the reference's end-to-end protocol (qserve_benchmark.py:48-67,108: generated
    # tokens / (prefill + decode wall time)) composed from the two measured phases
    prefill_ms = None
    if not args.no_prefill and world == 1:
        saved_len = eng.lengths.clone()
        try:
            eng.prefill(args.prompt_len)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.prefill(args.prompt_len)
            torch.cuda.synchronize()
            prefill_ms = (time.perf_counter() - t0) * 1e3
            eng.lengths.copy_(saved_len)               # the per-kernel timing below uses the step's context length
        except RuntimeError as e:                      # e.g. out of memory for the [tokens, 2*inter] buffer
            print(f"[bench] prefill phase skipped: {e}", file=sys.stderr)

    roof, kernels = None, None
    if not args.no_kernel_bench:
        try:
            kernels = kernel_bench(eng, torch)      # every rank times its own shard's kernels; rank 0 reports
        except Exception as e:
            if world == 1:
                raise
            print(f"[bench] rank {rank}: per-kernel timing skipped ({type(e).__name__}: {e})", file=sys.stderr)
    if kernels is not None:
        tot = {}
        for r in kernels:
            key = "w4a8_gemm" if r["kernel"].startswith("w4a8") else "decode_attention"
            tot[key] = tot.get(key, 0.0) + r["us"] * r["per_step"]
        dom_kind = max(tot, key=tot.get)
        dom = max((r for r in kernels if r["kernel"].startswith(dom_kind)), key=lambda r: r["us"] * r["per_step"])
        # HBM bytes per launch of that kernel from the L2 memory-side PMC counters: they need their own rocprofv3
        # passes (scripts/gpu_pmc.sh), so the committed measurement of the same kernel + shape is quoted here
        traffic, traffic_src = None, None
        try:
            import glob
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            newest = sorted(glob.glob(os.path.join(pdir, "r*_pmc_traffic.json")))[-1]   # named per round
            ent = json.load(open(newest))["kernels"].get(dom["kernel"])
            if ent:
                traffic = ent["hbm_bytes"]
                traffic_src = f"profiles/{os.path.basename(newest)} ({ent['kernel_symbol']}, rocprofv3 --pmc, offline pass)"
        except (OSError, ValueError, KeyError, IndexError):
            pass
        roof = dict(bound="hbm", kernel=dom["kernel"], achieved=round(dom["gbs"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(dom["gbs"] / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                    us_per_launch=round(dom["us"], 2), algorithmic_bytes=dom["bytes"])
        for r in kernels:
            r["us"], r["gbs"] = round(r["us"], 2), round(r["gbs"], 1)
            if "tops" in r:
                r["tops"] = round(r["tops"], 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, cfg)
        cpu["value"] = round(cpu["value"], 3)

    if rank == 0:
        out = {
            "metric": "decode tokens/sec/GPU Llama-3-8B W4A8KV4 bs=64" if args.model == "llama3-8b" and per_gpu_batch == 64
            else f"decode tokens/sec {cfg['name']} bs={args.batch}",
            "value": round(args.batch / (ms / 1e3), 1),
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "int8xint4->int32 (MFMA) + fp16",
            "data": "synthetic random-quantised weights/activations, KV cache written by the prefill writer",
            "config": {"workload": f"{cfg['name']} W4A8{'g128' if args.group_size == 128 else ' per-channel'} "
                                   f"KV{'8' if args.kv8 else '4'} decode step, bs={args.batch}"
                                   f"{f' ({per_gpu_batch} per GPU x tp{world})' if world > 1 else ''}, context "
                                   f"{args.prompt_len}->+{args.max_new} (BASELINE.json configs[1])",
                       "global_batch": args.batch, "context_start": args.prompt_len + 1 + args.warmup,
                       "parallelism": f"tp{world}",
                       "hipgraph": ("piecewise (collectives issued eagerly between the pieces)"
                                    if graphed and world > 1 and not args.tp_full_graph else graphed), "layers": cfg["layers"],
                       "op_sequence": "reference ops one by one" if args.op_by_op else
                       "reference ops; (residual add, layer norm) and (silu_and_mul, quant) issued as bit-identical "
                       "fused pairs (qserve_amd/fused.py)",
                       "op_by_op_tokens_per_s": round(args.batch / (ms_op / 1e3), 1) if ms_op else None,
                       "prefill_tokens_per_s": round(args.batch * args.prompt_len / (prefill_ms / 1e3), 1)
                       if prefill_ms else None,
                       "prefill_ms": round(prefill_ms, 2) if prefill_ms else None,
                       "e2e_tokens_per_s": round(args.batch * args.max_new / ((prefill_ms + args.max_new * ms) / 1e3), 1)
                       if prefill_ms else None,
                       "e2e_note": "reference protocol (qserve_benchmark.py: generated tokens / (prefill + decode) wall "
                                   f"time) for {args.max_new} generated tokens per sequence, composed from the measured "
                                   "prefill time and the measured decode step time"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
