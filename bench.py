#!/usr/bin/env python3
"""bench.py -- decode throughput of the W4A8KV4 hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one decode step of a Llama-3-8B-shaped W4A8KV4 model (BASELINE.json configs[1]: per-channel W4A8, KV4,
bs=64, context 1024 -> +512) over synthetic random-quantised weights and a cache written by the prefill writer:
32 layers x (4 W4A8 GEMMs + paged KV4 attention + the activation-side kernels) + final norm + fp16 lm_head +
greedy sampling, replayed from a hipGraph.  N > 1 = tensor parallel over N GPUs (column / row shards of the quantised
weights, 2 RCCL all-reduces per layer); default WEAK scaling (64 sequences per GPU, `--scaling strong` keeps the
global batch).  `python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run.

`--config` selects the other BASELINE.json configurations (3: g128 bs=128; 4: Qwen1.5-72B TP=8 global bs=64, strong;
5: KV8, 8k context, bs=8); they are parity-test cases first and bench lines only on request.

Prints ONE JSON line on rank 0 with the contract's keys plus
  roofline         the single kernel with the largest per-step time, timed live with HIP events on the launch stream
  roofline_family  the W4A8 GEMM family (all four decode GEMMs of a layer): family bytes / family time
  cpu_baseline     the PyTorch-CPU dequant + matmul / attention path (oracle/torch_cpu.py) on this box's host cores
  kernels[]        every hot-path kernel at the step's shapes; attention at start / mid / end of generation; the
                   compute-bound 4096^3 GEMMs (BASELINE configs[0]) with their fraction of the INT8 MFMA peak
  config.*         step time at mid / end context, eager (no hipGraph) and op-by-op rates, the reference's end-to-end
                   protocol (prefill + 511 decode steps, qserve_benchmark.py:48-67,108) MEASURED, not composed.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
INT8_PEAK_TOPS = 5000.0   # 256 CU x 4 SIMD x 2048 int8 op/clk x 2.4 GHz dense (guide's ubench ceiling: >= 3944)

CONFIGS = {   # BASELINE.json `configs` index -> flags
    2: dict(model="llama3-8b", batch=64, prompt_len=1024, max_new=512, group_size=-1, kv8=False),
    3: dict(model="llama3-8b", batch=128, prompt_len=1024, max_new=512, group_size=128, kv8=False),
    4: dict(model="qwen1.5-72b", batch=64, prompt_len=1024, max_new=512, group_size=-1, kv8=False, scaling="strong", gpus=8),
    5: dict(model="llama3-8b", batch=8, prompt_len=7680, max_new=512, group_size=-1, kv8=True),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[] index (1-based as in the task text: 2 = the headline configuration)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--prompt-len", type=int, default=None)
    ap.add_argument("--max-new", type=int, default=None)
    ap.add_argument("--group-size", type=int, default=None, choices=[-1, 128])
    ap.add_argument("--kv8", action="store_true", default=None)
    ap.add_argument("--model", default=None, choices=["llama3-8b", "llama2-7b", "llama2-70b", "qwen1.5-72b", "tiny"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-bench", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the prompt phase / end-to-end protocol measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the context sweep and the eager / op-by-op timings")
    ap.add_argument("--op-by-op", action="store_true",
                    help="issue the reference's ops one by one (no fused pairs) in the timed step")
    ap.add_argument("--gemm-variant", type=int, default=-1, help="A/B: qs_set_gemm_variant code (include/qserve_amd.h)")
    ap.add_argument("--attn-variant", type=int, default=0, help="A/B: qs_set_attention_variant code")
    ap.add_argument("--tp-full-graph", action="store_true",
                    help="N>1: capture the all-reduces into the step's hipGraph as well (default: one graph per segment "
                         "between the collectives, collectives issued eagerly - independent of capture support in RCCL)")
    ap.add_argument("--collective", default="auto", choices=["auto", "rccl", "direct"],
                    help="N>1: how the row-parallel partials are summed.  rccl = torch.distributed all-reduce, issued eagerly "
                         "between hipGraph pieces; direct = the library's direct-access all-reduce (qs_comm_*: one kernel per "
                         "collective, whole step in one hipGraph); auto (default) = direct if its start-up self-check on THIS "
                         "machine passes (bit-exact against the gathered inputs, no time-out, not slower than the process "
                         "group's all-reduce), else rccl - the line says which and why")
    ap.add_argument("--direct-allreduce", action="store_true", help="same as --collective direct")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N>1 (tensor parallel): weak = --batch sequences PER GPU (global batch = batch x N, so every "
                         "rank keeps N=1's GEMM MACs and KV bytes); strong = --batch is the global batch")
    a = ap.parse_args()
    preset = CONFIGS[a.config] if a.config else CONFIGS[2]
    for k, v in preset.items():
        if getattr(a, k, None) is None:
            setattr(a, k, v)
    if a.scaling is None:
        a.scaling = "weak"
    if a.gpus is None:
        a.gpus = 1
    a.kv8 = bool(a.kv8)
    return a


SELF_CHECK_ITERS = 2000


def direct_allreduce_or_none(numel, rank, world, dev, mode):
    """The library's direct-access all-reduce, checked on THIS machine before it is trusted (its protocol assumes that an
    acknowledged store to uncached peer memory is visible at the peer - unverified on xGMI until a run like this one says
    so).  All of the following must hold on every rank, or the step runs on torch.distributed:
      * SELF_CHECK_ITERS eager all-reduces with fresh random payloads of varying size, each verified ON THE DEVICE against
        the fp32 rank-order sum of the inputs gathered through the process group (bit for bit; one host sync at the end);
      * a hipGraph of 8 back-to-back all-reduces (each input derived from the previous output), replayed 64 times with
        changing data, against the same chain through the process group - bit for bit;
      * no bounded wait timed out;
      * mode "auto": not slower than the process group's all-reduce.
    All ranks take the same decision.  -> (communicator | None, note)."""
    import torch
    import torch.distributed as dist
    from qserve_amd import tp as TP

    def agree(ok):
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        return all(flags)

    try:
        comm = TP.DirectAllReduce(numel, device=dev)
    except Exception as e:                                       # raised consistently on every rank
        return None, f"torch.distributed ({str(e)[:200]})"
    why = None
    try:
        unit = 8 * world
        g = torch.Generator(device=dev).manual_seed(977 + rank)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        gathered = [torch.empty((numel,), dtype=torch.float16, device=dev) for _ in range(world)]

        def rank_order_sum(parts, n):
            acc = torch.zeros((n,), dtype=torch.float32, device=dev)
            for p_ in parts:
                acc = acc + p_[:n].float()
            return acc.half()
        import random
        sizes = random.Random(4242)                              # the same size sequence on every rank
        for it in range(SELF_CHECK_ITERS):
            n = numel if it % 3 == 0 else sizes.randrange(1, numel // unit + 1) * unit
            x = (torch.randn((numel,), device=dev, generator=g) * 2).half()
            comm.input((numel,)).copy_(x)
            dist.all_gather(gathered, x)
            comm.all_reduce(n)
            bad += (comm.output((numel,))[:n].view(torch.int16) != rank_order_sum(gathered, n).view(torch.int16)).sum()
            if it % 400 == 399 and not agree(int(bad.item()) == 0 and not comm.error()):
                why = f"self-check failed within {it + 1} eager all-reduces (bit mismatch or a timed-out wait)"
                break
        if why is None:                                          # graph replays: 8 dependent all-reduces per replay
            chain = 8
            seed = torch.empty((numel,), dtype=torch.float16, device=dev)

            def run_chain(reduce_fn, src_buf, dst_of):
                cur = seed
                for _ in range(chain):
                    src_buf.copy_(cur * 0.5 + 0.25)
                    cur = reduce_fn()
                return dst_of(cur)
            ref = torch.empty((numel,), dtype=torch.float16, device=dev)
            tmp = torch.empty((numel,), dtype=torch.float16, device=dev)

            def ref_reduce():
                dist.all_gather(gathered, tmp)
                ref.copy_(rank_order_sum(gathered, numel))
                return ref

            def direct_reduce():
                comm.all_reduce(numel)
                return comm.output((numel,))
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                run_chain(direct_reduce, comm.input((numel,)), lambda c: c)
            for rep in range(64):
                seed.copy_((torch.randn((numel,), device=dev, generator=g)).half())
                want = run_chain(ref_reduce, tmp, lambda c: c.clone())
                graph.replay()
                bad += (comm.output((numel,)).view(torch.int16) != want.view(torch.int16)).sum()
            if not agree(int(bad.item()) == 0 and not comm.error()):
                why = "self-check failed in the hipGraph replays (bit mismatch or a timed-out wait)"
        if why is None and mode == "auto":
            buf = torch.zeros((numel,), dtype=torch.float16, device=dev)
            times = []
            for fn in (lambda: dist.all_reduce(buf), lambda: comm.all_reduce(numel)):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / 30)
            if not agree(times[1] <= times[0] and not comm.error()):
                why = f"slower than the process group's all-reduce here ({times[1] * 1e3:.1f} vs {times[0] * 1e3:.1f} us on rank {rank})"
    except Exception as e:
        why = f"self-check raised {type(e).__name__}: {str(e)[:200]}"
        try:
            agree(False)
        except Exception:
            pass
    if why is not None:
        comm.close()
        return None, f"torch.distributed ({why})"
    return comm, (f"library direct-access all-reduce (qs_comm_all_reduce_f16); start-up self-check passed: {SELF_CHECK_ITERS} eager "
                  f"all-reduces of varying size + 64 replays of an 8-deep hipGraph chain, bit-exact against the rank-order fp32 sum")


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-execute under torch.distributed.run with
    one rank per GPU (the form the driver uses explicitly) and relay its output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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
        meas = []
        for _ in range(3):                 # median of three measurements of four replays each
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(4):
                g.replay()
            e1.record(st)
            torch.cuda.synchronize()
            meas.append(e0.elapsed_time(e1) * 1e3 / (4 * reps))
    return sorted(meas)[1]


def time_steps(eng, n, torch):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def kernel_bench(eng, args, torch, contexts):
    """Per-kernel timing of the hot-path kernels at the step's own shapes and data (rotating over the layers so the
    256 MiB Infinity Cache cannot serve the 109 MB of weights per layer)."""
    import qserve_backend.fused_attention as fa
    B, nl = eng.B, len(eng.layers)
    res = []
    qa, qo = eng.q_act, eng.q_attn
    specs = [("qkv", qa, eng.qkv_buf), ("o", qo, eng.proj_out), ("gate_up", qa, eng.gate_up_buf),
             ("down", eng.q_mlp, eng.proj_out)]
    sums = eng.q_sum if eng.group_size == -1 else None
    for name, x, out in specs:
        lin0 = eng.layers[0][name]
        by = gemm_bytes(B, lin0.n, lin0.k, eng.group_size)
        label = name
        if name == "gate_up" and eng.fuse_pairs and lin0.bias is None:
            # what the step launches: the GEMM with the silu * mul epilogue (writes [M, N/2] instead of [M, N])
            us = time_kernel(lambda i: eng.layers[i % nl][name].silu_mul(x, eng.q_scale, eng.q_sum, eng.mlp_act, out),
                             4 * nl, torch)
            by -= B * lin0.n
            label = "gate_up+silu*mul"
        else:
            us = time_kernel(lambda i: eng.layers[i % nl][name](x, eng.q_scale, eng.q_sum, out), 4 * nl, torch)
        res.append(dict(kernel=f"w4a8_gemm[{label} M={B} N={lin0.n} K={lin0.k}]", family="w4a8_gemm", us=us, bytes=by,
                        gbs=by / us / 1e3, tops=2.0 * B * lin0.n * lin0.k / us / 1e6, per_step=nl))
    q, k, v = eng.qkv_buf.split([eng.H * 128, eng.Hkv * 128, eng.Hkv * 128], dim=-1)
    q, k, v = q.reshape(B, eng.H, 128), k.reshape(B, eng.Hkv, 128), v.reshape(B, eng.Hkv, 128)
    eng.qkv_buf.normal_()
    saved = eng.lengths.clone()

    def attn(i):
        fa.single_query_attention(q, k, v, eng.tables[i % nl], eng.lengths, None, 8192, 64, eng.size_per_token,
                                  eng.max_len, 128, eng.cfg["rope_theta"], True, eng.int4, True)
    # what the step launches when the pair fusions are on: attention + invoke_quant(_fuse_sum) of its output in ONE launch (the
    # last workgroup of a sequence quantises the row, qserve_amd/fused.py).  The roofline object is computed on this launch -
    # its time includes the quantiser's seam (~0.7 us), its bytes the int8 row and statistics; the plain kernel is listed too.
    fused_attn = eng.fuse_pairs and eng.tp_world == 1
    sums = eng.q_sum if eng.group_size == -1 else None

    def attn_quant(i):
        from qserve_amd import fused as fz
        fz.single_query_attention_quant(q, k, v, eng.tables[i % nl], eng.lengths, eng.q_attn, eng.q_scale, 8192, 64,
                                        eng.size_per_token, eng.max_len, 128, eng.cfg["rope_theta"], True, eng.int4, True,
                                        quant_sum=sums)
    for j, L in enumerate(contexts):           # context INCLUDING the new token; first entry = the timed step's context
        eng.lengths.fill_(L)
        us = time_kernel(attn, 4 * nl, torch)
        by = attn_bytes(B, eng.H, eng.Hkv, L - 1, eng.int4)
        res.append(dict(kernel=f"decode_attention[B={B} H={eng.H} Hkv={eng.Hkv} L={L}]", family="decode_attention",
                        us=us, bytes=by, gbs=by / us / 1e3, frac_of_hbm_peak=by / us / 1e3 / HBM_PEAK_GBS,
                        per_step=0 if fused_attn else (nl if j == 0 else 0)))
        if j == 0 and fused_attn:
            us2 = time_kernel(attn_quant, 4 * nl, torch)
            by2 = by + B * eng.H * 128 + 4 * B            # + the int8 row and the per-token statistics
            res.append(dict(kernel=f"decode_attention+quant[B={B} H={eng.H} Hkv={eng.Hkv} L={L}]", family="decode_attention_step",
                            us=us2, bytes=by2, gbs=by2 / us2 / 1e3, frac_of_hbm_peak=by2 / us2 / 1e3 / HBM_PEAK_GBS,
                            per_step=nl))
    eng.lengths.copy_(saved)
    return res


def other_configs(torch, D, dev):
    """BASELINE.json configs[2] (g128 per-group, bs = 128) and configs[4] (KV8, bs = 8, 8k context) on the driver-observed
    line: a short graph-replayed run each (8 warm-up + 16 timed steps at the start of the generation, no CPU leg), with the
    decode attention's fraction of the HBM peak and the GEMM family's at that configuration's shapes."""
    res = {}
    for idx, name in ((3, "config3"), (5, "config5")):
        c = CONFIGS[idx]
        cfg = D.LLAMA3_8B
        try:
            e = D.DecodeEngine(cfg, c["batch"], c["prompt_len"], c["max_new"], group_size=c["group_size"],
                               int4_kv=not c["kv8"], device=dev)
            start = c["prompt_len"] + 1
            e.prefill_cache(c["prompt_len"] + 40)
            e.lengths.fill_(start)
            e.capture()
            e.lengths.fill_(start)
            for _ in range(8):
                e.run()
            ms = time_steps(e, 16, torch)
            ns = argparse.Namespace(no_extras=True)
            ks = kernel_bench(e, ns, torch, [start + 8])
            att = next(r for r in ks if r["family"] == "decode_attention")
            fam = [r for r in ks if r["family"] == "w4a8_gemm"]
            fb, fu = sum(r["bytes"] for r in fam), sum(r["us"] for r in fam)
            res[name] = dict(workload=f"Llama-3-8B W4A8{'g128' if c['group_size'] == 128 else ' per-channel'} "
                                      f"KV{'8' if c['kv8'] else '4'}, bs={c['batch']}, context {start + 8}..{start + 23}",
                             tokens_per_s=round(c["batch"] / (ms / 1e3), 1), ms_per_step=round(ms, 4), steps=16, warmup=8,
                             attention_us=round(att["us"], 2), attention_frac=round(att["gbs"] / HBM_PEAK_GBS, 4),
                             gemm_family_us=round(fu, 2), gemm_family_frac=round(fb / fu / 1e3 / HBM_PEAK_GBS, 4))
            del e
            torch.cuda.empty_cache()
        except RuntimeError as ex:
            res[name] = dict(skipped=str(ex)[:200])
    return res


NOMINAL_CLOCK_GHZ = 2.4      # the clock INT8_PEAK_TOPS is quoted at (MI355X_MICROARCH.md)


def measured_gemm_clock_ghz(fn, reps, torch, dev):
    """The engine clock a compute-bound GEMM launch holds under its own load: `reps` launches back to back (the chip settles
    into the power state of the timed loop), the LAST one stamped from inside the kernel - every workgroup reports its life in
    shader cycles (s_memtime) and in ticks of the constant 100 MHz counter (include/qserve_amd.h qs_debug_gemm_clock_probe);
    median over the workgroups of cycles / (ticks x 10 ns).  None if the launch is not one of the probed kernels."""
    from qserve_amd._lib import lib
    cap = 4096
    buf = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
    for i in range(reps - 1):
        fn(i)
    if lib.qs_debug_gemm_clock_probe(buf.data_ptr(), cap) != 0:
        return None
    try:
        fn(reps - 1)
        torch.cuda.synchronize()
    finally:
        lib.qs_debug_gemm_clock_probe(None, 0)
    b = buf.cpu()
    live = b[:, 1] > 0
    if int(live.sum()) == 0:
        return None
    ghz = (b[live, 0].double() / (b[live, 1].double() * 10.0)).sort().values     # cycles per ns
    return float(ghz[len(ghz) // 2])


def gemm_config1_bench(torch, dev):
    """BASELINE.json configs[0] on the GPU: 4096 x 4096 x 4096 W4A8 GEMM, per-channel and per-group, compute-bound;
    TOPS against the dense INT8 MFMA peak (north_star: >= 70 %)."""
    import qserve_backend.qgemm_w4a8_per_chn as gc
    import qserve_backend.qgemm_w4a8_per_group as gg
    M = N = K = 4096
    g = torch.Generator(device=dev).manual_seed(4096)
    nset = 4
    A = [torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g) for _ in range(nset)]
    W = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(nset)]
    ws = (torch.rand((N,), device=dev, generator=g) * 0.01 + 0.002).half()
    sa = (torch.rand((M,), device=dev, generator=g) * 0.04 + 0.005).half()
    s2 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev, generator=g)
    z2 = (-(torch.randint(0, 16, (K // 128, N), device=dev, generator=g).to(torch.int16) * s2.to(torch.int16))).to(torch.int8)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    res = []
    ops = 2.0 * M * N * K
    for name, fn, grp in (("per_channel", lambda i: gc.gemm_forward_cuda(A[i % nset], W[i % nset], ws, sa, ws, sa, out), -1),
                          ("per_group", lambda i: gg.gemm_forward_cuda(A[i % nset], W[i % nset], z2, s2, ws, sa, out), 128)):
        us = time_kernel(fn, 8, torch)
        clk = measured_gemm_clock_ghz(fn, 8, torch, dev)
        ent = dict(kernel=f"w4a8_gemm[config1 {name} M={M} N={N} K={K}]", family="w4a8_gemm_compute_bound", us=us,
                   bytes=gemm_bytes(M, N, K, grp), tops=ops / us / 1e6, frac_of_int8_mfma_peak=ops / us / 1e6 / INT8_PEAK_TOPS,
                   peak_tops=INT8_PEAK_TOPS, per_step=0)
        if clk:
            # "compute peak from the measured engine clock during the run and state it" - the nominal 5.0 POPS are
            # 256 CUs x 4 SIMDs x 2048 int8 ops per clock at 2.4 GHz; under this launch's own load the chip holds `clk` GHz
            ent.update(sustained_clock_ghz=clk, peak_tops_at_measured_clock=INT8_PEAK_TOPS * clk / NOMINAL_CLOCK_GHZ,
                       frac_of_peak_at_measured_clock=ops / us / 1e6 / (INT8_PEAK_TOPS * clk / NOMINAL_CLOCK_GHZ))
        res.append(ent)
    return res


FP16_PEAK_TFLOPS = 2500.0       # dense fp16 MFMA peak (MI355X_MICROARCH.md: 2.5 PF dense, 256 CUs x 4 SIMDs x 1024 flop / clock at 2.4 GHz)


def prefill_attention_bench(torch, dev, cfg, batch, prompt_len):
    """The prompt phase's attention provider (flash_attn_varlen_func -> csrc/flash_prefill.hip) at the headline shape: `batch`
    prompts of `prompt_len` tokens, causal; flops counted on the lower triangle (4 B H L^2 Dh / 2), against the dense fp16 MFMA
    peak.  Not part of the decode step (`per_step` 0); VERDICT r05 item 7."""
    from flash_attn.flash_attn_interface import flash_attn_varlen_func
    H, Hkv, L = cfg["heads"], cfg["kv_heads"], prompt_len
    T = batch * L
    g = torch.Generator(device=dev).manual_seed(7)
    qkv = torch.randn((T, (H + 2 * Hkv) * 128), dtype=torch.float16, device=dev, generator=g)
    q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
    q, k, v = q.reshape(T, H, 128), k.reshape(T, Hkv, 128), v.reshape(T, Hkv, 128)
    cu = torch.arange(0, batch + 1, dtype=torch.int32, device=dev) * L
    for _ in range(2):
        flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
    torch.cuda.synchronize()
    meas = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
        e1.record()
        torch.cuda.synchronize()
        meas.append(e0.elapsed_time(e1) * 1e3 / 5)
    us = sorted(meas)[1]
    flops = 4.0 * batch * H * L * L * 128 / 2
    return [dict(kernel=f"flash_attn_varlen[causal B={batch} L={L} H={H} Hkv={Hkv}]", family="prefill_attention", us=us,
                 bytes=int(2 * T * (2 * H + 2 * Hkv) * 128), tflops=flops / us / 1e6, frac_of_fp16_mfma_peak=flops / us / 1e6 / FP16_PEAK_TFLOPS,
                 peak_tflops=FP16_PEAK_TFLOPS, per_step=0)]


def cpu_baseline(args, cfg, torch):
    """The reference's PyTorch-CPU linear / attention path (oracle/torch_cpu.py: unpack + de-quantise + fp32 matmul; page
    gather + de-quantise + fp32 softmax attention) on the host cores, on a bounded sample of the step: the four GEMMs of a
    layer at M = batch over fresh weights for ~10 s, decode attention for 8 sequences; extrapolated to tokens/s of a whole
    step (layers x (GEMMs + batch sequences))."""
    from oracle import torch_cpu as T
    B = args.batch
    H, Hkv, hid, inter = cfg["heads"], cfg["kv_heads"], cfg["hidden"], cfg["inter"]
    shapes = [((H + 2 * Hkv) * 128, hid), (hid, hid), (2 * inter, hid), (hid, inter)]
    g = torch.Generator().manual_seed(0)
    t_gemm, reps = 0.0, 0
    while reps < 2 or (t_gemm < 10.0 and reps < 32):
        for N, K in shapes:
            A = torch.randint(-127, 128, (B, K), dtype=torch.int8, generator=g)
            qw = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, generator=g)
            ws = (torch.rand((N,), generator=g) * 0.018 + 0.002).half()
            sa = (torch.rand((B,), generator=g) * 0.045 + 0.005).half()
            t0 = time.perf_counter()
            if args.group_size == -1:
                T.linear_per_channel(A, qw, ws, sa, ws)
            else:
                s2 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, generator=g)
                T.linear_per_group(A, qw, s2, s2, ws, sa)
            if reps:                     # first pass = warm-up (thread pool, allocator)
                t_gemm += time.perf_counter() - t0
        reps += 1
    t_layer = t_gemm / (reps - 1)
    nseq, L = 8, args.prompt_len
    int4 = not args.kv8
    mb = (L + 63) // 64
    pb = Hkv * 64 * (64 if int4 else 128) + Hkv * 256
    kp = torch.randint(0, 256, (nseq * mb, pb), dtype=torch.uint8, generator=g)
    vp = torch.randint(0, 256, (nseq * mb, pb), dtype=torch.uint8, generator=g)
    nd = Hkv * 64 * (64 if int4 else 128)
    for p in (kp, vp):                   # sane fp16 scales / zeros
        p[:, nd:].view(torch.float16).fill_(0.25)
    tables = torch.stack([torch.randperm(nseq * mb, generator=g).reshape(nseq, mb),
                          torch.randperm(nseq * mb, generator=g).reshape(nseq, mb)], dim=1)
    qr = torch.randn((nseq, H, 128), generator=g)
    T.decode_attention(qr, kp, vp, tables, L, Hkv, int4)
    t0 = time.perf_counter()
    n_att = 0
    while n_att < 2 or time.perf_counter() - t0 < 4.0:
        T.decode_attention(qr, kp, vp, tables, L, Hkv, int4)
        n_att += 1
    t_attn = (time.perf_counter() - t0) / n_att / nseq
    step_s = cfg["layers"] * (t_layer + B * t_attn)
    return dict(value=B / step_s, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"PyTorch-CPU path of BASELINE.md section 3 (oracle/torch_cpu.py: unpack + de-quantise + fp32 "
                       f"torch.matmul; page gather + de-quantise + fp32 softmax attention), torch threads = `cores` of "
                       f"{os.cpu_count()} logical CPUs, {t_gemm + n_att * t_attn * nseq:.1f} s of CPU work: one layer's "
                       f"{'per-channel' if args.group_size == -1 else 'g128'} GEMMs (M={B}) x {reps - 1} weight sets = "
                       f"{t_layer:.3f} s per layer; decode attention {t_attn * 1e3:.2f} ms/sequence at L={L} ({nseq} "
                       f"sequences x {n_att} runs); step = {cfg['layers']} layers x (GEMMs + {B} sequences)")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): running tp{world}", file=sys.stderr)
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
    from qserve_amd import _lib
    from qserve_amd import decode as D
    if args.gemm_variant != -1:
        _lib.lib.qs_set_gemm_variant(args.gemm_variant)
    if args.attn_variant != 0:
        _lib.lib.qs_set_attention_variant(args.attn_variant)
    cfg = {"llama3-8b": D.LLAMA3_8B, "llama2-7b": D.LLAMA2_7B, "llama2-70b": D.LLAMA2_70B, "qwen1.5-72b": D.QWEN15_72B,
           "tiny": D.TINY}[args.model]
    direct, collective_note = None, None
    mode = "direct" if args.direct_allreduce else args.collective
    if world > 1 and mode != "rccl":
        direct, collective_note = direct_allreduce_or_none(args.batch * cfg["hidden"], rank, world, dev, mode)
    eng = D.DecodeEngine(cfg, args.batch, args.prompt_len, args.max_new, group_size=args.group_size,
                         int4_kv=not args.kv8, device=dev, tp_rank=rank, tp_world=world,
                         fuse_pairs=not args.op_by_op, direct_allreduce=direct)
    # the whole cache of the generation (prompt + max_new - 1 positions) is written by the prefill writer up front, so
    # that any context of the run (start / mid / end) reads real quantised pages; `lengths` selects the context
    full_ctx = args.prompt_len + args.max_new - 1
    eng.prefill_cache(full_ctx)
    start_len = args.prompt_len + 1
    eng.lengths.fill_(start_len)
    graphed = False
    full_graph = args.tp_full_graph
    if full_graph and world > 1 and direct is None and backend != "nccl":
        # a host-staged backend (gloo) synchronises inside the collective: capturing it can only fail, and a failed capture
        # leaves this HIP context unusable (measured: every later call returns hipErrorStreamCaptureInvalidated) - refuse
        if rank == 0:
            print(f"[bench] --tp-full-graph ignored: the '{backend}' backend cannot be captured; piecewise graphs instead",
                  file=sys.stderr)
        full_graph = False
    if not args.no_graph:
        try:
            eng.capture(piecewise=False if full_graph else None)
            graphed = True
        except Exception as e:
            # A collective that refuses capture (an RCCL build without graph support) invalidates the capture, and HIP keeps
            # returning hipErrorStreamCaptureInvalidated to this process afterwards: there is nothing to fall back to IN this
            # process.  Say so and stop - every rank takes this path together (the capture is collective).
            print(f"[bench] rank {rank}: hipGraph capture of the step failed ({type(e).__name__}: {str(e).splitlines()[0]}).  "
                  f"{'Re-run without --tp-full-graph (piecewise graphs need no capturable collective).' if full_graph else 'Re-run with --no-graph.'}",
                  file=sys.stderr)
            os._exit(3)
    eng.lengths.fill_(start_len)
    assert args.warmup + args.steps + 1 <= args.max_new, "steps exceed the page budget (prompt_len + max_new)"

    # ---- the contract's timed region: W warm-up steps, then exactly K steps between barriers + synchronize ----------
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
    extra = {}
    eng.check()                         # bounded in-launch waits (direct all-reduce): raise if one gave up

    # ---- per-kernel timing (feeds the roofline objects below): taken HERE, straight after the timed steps and in the same
    # thermal / clock state - not behind the prompt phase, the end-to-end protocol and the reference-engine child process,
    # after which the same kernels measured up to 8 % slower on some boxes (round 4: 18.0 vs 19.5 us for the attention)
    kernels = None
    if not args.no_kernel_bench:
        try:
            ctxs = [start_len + args.warmup]
            if world == 1 and not args.no_extras:
                ctxs += [args.prompt_len + args.max_new // 2, args.prompt_len + args.max_new - 1]
            kernels = kernel_bench(eng, args, torch, ctxs)      # every rank times its own shard's kernels; rank 0 reports
            if world == 1 and not args.no_extras and args.model == "llama3-8b":
                kernels += gemm_config1_bench(torch, dev)
                kernels += prefill_attention_bench(torch, dev, cfg, per_gpu_batch, args.prompt_len)
        except Exception as e:
            if world == 1:
                raise
            print(f"[bench] rank {rank}: per-kernel timing skipped ({type(e).__name__}: {e})", file=sys.stderr)

    # ---- secondary timings (single GPU): other contexts of the generation, eager launches, op-by-op sequence ---------
    sweep = {}
    n2 = min(args.steps, 24)
    mid_len, end_len = args.prompt_len + args.max_new // 2, args.prompt_len + args.max_new - n2 - 3
    if world == 1 and not args.no_extras:
        for L in (mid_len, end_len):
            eng.lengths.fill_(L)
            eng.run()
            sweep[L + 1] = round(args.batch / (time_steps(eng, n2, torch) / 1e3), 1)
        extra["decode_tokens_per_s_by_context"] = {str(start_len + args.warmup): round(args.batch / (ms / 1e3), 1),
                                                  **{str(k): v for k, v in sweep.items()}}
        extra["decode_tokens_per_s_mid"] = sweep[mid_len + 1]
        if graphed:
            g_saved, p_saved = eng.graph, eng.pieces
            eng.graph, eng.pieces = None, None
            eng.lengths.fill_(start_len)
            for _ in range(2):
                eng.run()
            extra["eager_tokens_per_s"] = round(args.batch / (time_steps(eng, n2, torch) / 1e3), 1)
            eng.graph, eng.pieces = g_saved, p_saved
            if not args.op_by_op:
                eng.fuse_pairs = False
                eng.lengths.fill_(start_len)
                eng.capture()
                eng.lengths.fill_(start_len)
                for _ in range(2):
                    eng.run()
                extra["op_by_op_tokens_per_s"] = round(args.batch / (time_steps(eng, n2, torch) / 1e3), 1)
                eng.fuse_pairs = True
                eng.graph, eng.pieces = g_saved, p_saved

    # ---- one rank's COMPUTE-ONLY share of the tensor-parallel step (N = 1 line only): what a real N-GPU run should take per
    # step before communication - the prediction the driver's scaling run can be checked against --------------------------
    if world == 1 and not args.no_extras and args.model == "llama3-8b" and not args.op_by_op:
        share = {}
        for tpw in (2, 4, 8):
            try:
                e2 = D.DecodeEngine(cfg, per_gpu_batch * tpw, args.prompt_len, args.max_new, group_size=args.group_size,
                                    int4_kv=not args.kv8, device=dev, tp_rank=0, tp_world=tpw)
                e2.prefill_cache(args.prompt_len + 16)
                e2.lengths.fill_(start_len)
                e2.capture(piecewise=False)          # no process group: the all-reduces are no-ops, one graph
                e2.lengths.fill_(start_len)
                for _ in range(3):
                    e2.run()
                share[f"tp{tpw}"] = round(time_steps(e2, 16, torch), 3)
                del e2
                torch.cuda.empty_cache()
            except RuntimeError as e:
                print(f"[bench] tp{tpw} rank share skipped: {str(e)[:200]}", file=sys.stderr)
        extra["tp_rank_share_ms"] = share
        extra["tp_rank_share_note"] = ("ms per step of ONE rank's shard at tp2 / tp4 / tp8 (weak scaling: 64 sequences per GPU) "
                                       "with the collectives as no-ops, measured on this GPU: an N-GPU run adds 2 all-reduces "
                                       "per layer + the greedy head's exchange on top; N > 1 itself is UNMEASURED here")

    # ---- the reference's end-to-end protocol, measured: prompt phase + (max_new - 1) decode steps ---------------------
    # (qserve_benchmark.py:48-67,108: the prompt step yields the first of `max_new` tokens; tokens / total wall time)
    if not args.no_prefill and world == 1:
        try:
            eng.prefill(args.prompt_len)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.prefill(args.prompt_len)
            torch.cuda.synchronize()
            prefill_ms = (time.perf_counter() - t0) * 1e3
            t1 = time.perf_counter()
            for _ in range(args.max_new - 1):
                eng.run()
            torch.cuda.synchronize()
            decode_ms = (time.perf_counter() - t1) * 1e3
            extra.update(prefill_ms=round(prefill_ms, 2),
                         prefill_tokens_per_s=round(args.batch * args.prompt_len / (prefill_ms / 1e3), 1),
                         e2e_decode_ms=round(decode_ms, 2),
                         e2e_tokens_per_s=round(args.batch * args.max_new / ((prefill_ms + decode_ms) / 1e3), 1),
                         e2e_decode_only_tokens_per_s=round(args.batch * (args.max_new - 1) / (decode_ms / 1e3), 1),
                         e2e_note="reference protocol (qserve_benchmark.py:48-67,108) MEASURED in this run: one prompt "
                                  f"phase of {args.batch} x {args.prompt_len} tokens (W4A8 GEMMs, prefill KV writer, causal "
                                  f"flash attention) + {args.max_new - 1} decode steps; tokens = batch x {args.max_new}")
        except RuntimeError as e:                      # e.g. out of memory for the [tokens, 2*inter] buffer
            print(f"[bench] end-to-end protocol skipped: {e}", file=sys.stderr)
        eng.lengths.fill_(start_len + args.warmup)

    # ---- the reference's OWN engine running its own benchmark protocol over this backend (when its tree is staged under the
    # git-ignored oracle/_ref/, scripts/stage_reference.sh): qserve_benchmark.py:process_requests, LLMEngine, scheduler, block
    # manager, model code - all unchanged - in a child process (scripts/run_reference_engine.py), Llama-3-8B shape, synthetic
    # random-quantised weights, batch x 1024 -> 512.  Beside e2e_tokens_per_s: same protocol through this repository's engine.
    if (world == 1 and not args.no_extras and not args.no_prefill and not args.op_by_op and args.model == "llama3-8b"
            and args.group_size == -1 and not args.kv8 and os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "qserve"))):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_reference_engine.py"), "--mode", "protocol",
                                "--backend", "ext", "--batch", str(args.batch), "--prompt-len", str(args.prompt_len),
                                "--gen-len", str(args.max_new), "--rounds", "2"], capture_output=True, text=True, timeout=600)
            rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            extra.update(reference_engine_tokens_per_s=rec["reference_engine_tokens_per_s"],
                         reference_engine_note="the reference's unchanged LLMEngine + qserve_benchmark.py:process_requests over "
                                               "the compiled qserve_backend extension of this repository (staged copy under "
                                               "oracle/_ref/, second of two rounds as the reference's script reports; eager "
                                               "launches, the reference's op-by-op sequence and Python per step); logits "
                                               f"finite: {rec['rounds'][-1]['all_logits_finite']}")
        except Exception as e:                                    # absent extension, no JSON line, time-out ...
            print(f"[bench] reference engine leg skipped: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr)

    # ---- collective cost (N > 1): one fp16 all-reduce of the row-parallel partial [batch, hidden] ---------------------
    if world > 1:
        buf = torch.zeros_like(eng.proj_out)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        extra.update(all_reduce_us=round(e0.elapsed_time(e1) * 1e3 / 50, 2), all_reduce_bytes=buf.numel() * 2,
                     all_reduces_per_step=2 * cfg["layers"], collective_backend=f"{backend} ({world} ranks)")
        if direct is not None:
            for _ in range(5):
                direct.all_reduce(buf.numel())
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                direct.all_reduce(buf.numel())
            e1.record()
            torch.cuda.synchronize()
            extra.update(direct_all_reduce_us=round(e0.elapsed_time(e1) * 1e3 / 50, 2),
                         direct_all_reduce_timeouts=bool(direct.error()))
        extra.update(step_collectives=collective_note or "torch.distributed all-reduce (--collective rccl)",
                     all_reduces_per_step=2 * cfg["layers"] + int(eng.vocab_parallel),
                     lm_head=("vocabulary-parallel: every rank multiplies its V/N rows, the ranks' greedy candidates meet in "
                              "one more (32-byte-per-sequence-and-rank) sum all-reduce" if eng.vocab_parallel
                              else "replicated on every rank"))

    # ---- per-kernel timing + roofline ---------------------------------------------------------------------------------
    roof, roof_family = None, None
    if kernels is not None:
        step_k = [r for r in kernels if r["per_step"]]
        dom = max(step_k, key=lambda r: r["us"] * r["per_step"])          # the single kernel with the largest step share
        # HBM bytes per launch of that kernel from the L2 memory-side PMC counters: they need their own rocprofv3
        # passes (scripts/gpu_pmc.sh), so the committed measurement of the same kernel + shape is quoted here
        traffic, traffic_src = None, None
        try:
            import glob
            pdir = os.path.join(ROOT, "profiles")
            def _rank(path):     # newest round first: round2_c > round2_b > ... > r05 > r04 (file names, not mtimes: a
                b = os.path.basename(path)          # snapshot copy gives every file the same timestamp)
                return (1, b) if b.startswith("round") else (0, b)
            for cand in sorted(glob.glob(os.path.join(pdir, "*_pmc_traffic.json")), key=_rank, reverse=True):
                ents = json.load(open(cand))["kernels"]
                label = dom["kernel"].replace("decode_attention+quant[", "decode_attention[")   # (the PMC workload launches the plain op)
                ent = ents.get(label) or next((v for k, v in ents.items() if k.split(" L=")[0] == label.split(" L=")[0]), None)
                if ent:
                    traffic = ent["hbm_bytes"]
                    traffic_src = (f"profiles/{os.path.basename(cand)} ({ent['kernel_symbol']}, rocprofv3 --pmc, offline "
                                   f"pass{'' if label in ents else ' at a neighbouring context length'}"
                                   f"{'; the plain attention launch, the fused quantiser adds the 0.5 MB int8 row' if label != dom['kernel'] else ''})")
                    break
        except (OSError, ValueError, KeyError, IndexError):
            pass
        roof = dict(bound="hbm", kernel=dom["kernel"], achieved=round(dom["gbs"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(dom["gbs"] / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                    us_per_launch=round(dom["us"], 2), algorithmic_bytes=dom["bytes"],
                    step_share=round(dom["us"] * dom["per_step"] / (ms * 1e3), 3))
        if dom["family"] == "decode_attention_step":   # the same kernel without the fused quantiser (what rounds 1-3 quoted)
            plain = next(r for r in kernels if r["family"] == "decode_attention")
            roof.update(plain_attention_us=round(plain["us"], 2), plain_attention_frac=round(plain["gbs"] / HBM_PEAK_GBS, 4),
                        note="timed as the step launches it: attention + invoke_quant(_fuse_sum) of its output in one launch; "
                             "plain_attention_* = qs_single_query_attention alone (the launch rounds 1-3 quoted)")
            # the whole generation: plain-kernel fractions at the measured contexts and their average over
            # the steps prompt_len + 1 .. prompt_len + max_new (launch time interpolated linearly between the measured contexts;
            # bytes exact per step): total algorithmic bytes / total attention time of a layer over the generation
            pts = sorted((int(r["kernel"].split(" L=")[1].rstrip("]")), r["us"]) for r in kernels if r["family"] == "decode_attention")
            roof["plain_attention_frac_by_context"] = {
                str(L): round(attn_bytes(eng.B, eng.H, eng.Hkv, L - 1, eng.int4) / us / 1e3 / HBM_PEAK_GBS, 4) for L, us in pts}
            if len(pts) >= 2:
                tb = tu = 0.0
                for L in range(args.prompt_len + 1, args.prompt_len + args.max_new + 1):
                    lo = max([p_ for p_ in pts if p_[0] <= L] or [pts[0]], key=lambda p_: p_[0])
                    hi = min([p_ for p_ in pts if p_[0] >= L] or [pts[-1]], key=lambda p_: p_[0])
                    us = lo[1] if hi[0] == lo[0] else lo[1] + (hi[1] - lo[1]) * (L - lo[0]) / (hi[0] - lo[0])
                    tb += attn_bytes(eng.B, eng.H, eng.Hkv, L - 1, eng.int4)
                    tu += us
                roof["plain_attention_generation_average_frac"] = round(tb / tu / 1e3 / HBM_PEAK_GBS, 4)
        fam = [r for r in step_k if r["family"] == "w4a8_gemm"]
        fb, fu = sum(r["bytes"] for r in fam), sum(r["us"] for r in fam)
        roof_family = dict(bound="hbm", family="w4a8_gemm (the four decode GEMMs of a layer)", achieved=round(fb / fu / 1e3, 1),
                           peak=HBM_PEAK_GBS, unit="GB/s", frac=round(fb / fu / 1e3 / HBM_PEAK_GBS, 4), us_per_layer=round(fu, 2),
                           algorithmic_bytes=fb, step_share=round(fu * len(eng.layers) / (ms * 1e3), 3))
        for r in kernels:
            for k2, nd in (("us", 2), ("gbs", 1), ("tops", 1), ("tflops", 1), ("frac_of_hbm_peak", 4), ("frac_of_int8_mfma_peak", 4),
                           ("frac_of_fp16_mfma_peak", 4),
                           ("sustained_clock_ghz", 3), ("peak_tops_at_measured_clock", 1), ("frac_of_peak_at_measured_clock", 4)):
                if k2 in r:
                    r[k2] = round(r[k2], nd)
    headline = (args.model == "llama3-8b" and per_gpu_batch == 64 and args.group_size == -1 and not args.kv8
                and args.prompt_len == 1024 and args.max_new == 512)
    if headline and world == 1 and not args.no_extras and not args.op_by_op:
        extra["other_configs"] = other_configs(torch, D, dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, cfg, torch)
        cpu["value"] = round(cpu["value"], 3)

    if rank == 0:
        cfg_idx = next((i for i, c in CONFIGS.items() if all(getattr(args, k) == v for k, v in c.items()
                                                              if k not in ("gpus", "scaling", "batch")) and c["batch"] == per_gpu_batch), None)
        out = {
            "metric": "decode tokens/sec/GPU Llama-3-8B W4A8KV4 bs=64" if headline else f"decode tokens/sec {cfg['name']} bs={args.batch}",
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
                                   f"{f' ({per_gpu_batch} per GPU x tp{world})' if world > 1 and args.scaling == 'weak' else ''}"
                                   f", context {args.prompt_len}->+{args.max_new}"
                                   f"{f' (BASELINE.json configs[{cfg_idx - 1}])' if cfg_idx else ''}",
                       "global_batch": args.batch, "context_start": start_len + args.warmup,
                       "parallelism": f"tp{world}",
                       "hipgraph": ("piecewise (collectives issued eagerly between the pieces)"
                                    if graphed and world > 1 and not full_graph and direct is None else graphed),
                       "layers": cfg["layers"],
                       "op_sequence": "reference ops one by one" if args.op_by_op else
                       ("reference ops; (residual add, layer norm), (gate_up GEMM, silu_and_mul) and (attention, quant) "
                        "issued as bit-identical fused pairs (qserve_amd/fused.py)" +
                        ("; K-slice planes: " + ", ".join(f"{n}_proj leaves int32 partial sums per K slice ({p.size(0)}), the add + "
                                                          "norm + quant launch behind it finishes the GEMM" for n, p in eng.planes.items())
                         if getattr(eng, "planes", None) else "")),
                       "launches_per_layer": 8 if not args.op_by_op else 12,
                       "value_context": f"decode steps at context {start_len + args.warmup}..{start_len + args.warmup + args.steps - 1} "
                                        "(start of the 1024 -> +512 generation); mid / end of generation: "
                                        "decode_tokens_per_s_by_context, whole generation: e2e_decode_only_tokens_per_s",
                       **extra},
            "roofline": roof,
            "roofline_family": roof_family,
            "cpu_baseline": cpu,
            "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
