"""One decode step of a Llama-style W4A8KV4 model expressed with the `qserve_backend` ops -- the bench / smoke driver.

It issues exactly the op sequence of the reference's model code for `is_prompt=False`
(qserve/modeling/models/llama_w4a8_unpad.py:185-291, 330-361, 69-93):

    rms_norm_general_fuse_sum -> qkv GEMM -> single_query_attention -> invoke_quant_fuse_sum -> o_proj GEMM
    -> residual add -> rms_norm_general_fuse_sum -> gate_up GEMM -> silu_and_mul -> invoke_quant_fuse_sum
    -> down GEMM -> residual add;   then rms_norm, fp16 lm_head, greedy sampling.

With `fuse_pairs=True` (default) the adjacent pairs (attention, quant of its output), (residual add, layer norm) and
(gate_up GEMM, silu_and_mul) are issued as one launch each (qserve_amd/fused.py) - same arithmetic, same intermediate fp16 roundings, bit-identical tensors
(tests/test_fused_gpu.py, tests/test_decode_gpu.py); `fuse_pairs=False` issues the reference's ops one by one.

Weights are synthetic (random packed nibbles / scales of the right shapes, distinct per layer so nothing is
served from cache); the KV cache is filled by the prefill writer.  Tensor parallelism: rank r owns
H/tp query heads, Hkv/tp KV heads and the matching column / row shards; the partial outputs of o_proj and
down_proj are summed with one RCCL all-reduce each.
"""
import os

import torch

import qserve_backend.activation_ops as activation_ops
import qserve_backend.fused_attention as fused_attention
import qserve_backend.fused_kernels as fused_kernels
import qserve_backend.layernorm_ops as layernorm_ops
import qserve_backend.qgemm_w4a8_per_chn as gemm_chn
import qserve_backend.qgemm_w4a8_per_group as gemm_grp

from . import fused as fusedmod
from . import tp as tpmod
from ._lib import device_status as _device_status
from .backend._util import check as _check, lib as _lib, stream


def residual_add_(a, b):
    """a += b (fp16), the residual add the reference does with a torch add (llama_w4a8_unpad.py:348,360)."""
    _check(_lib.qs_residual_add(a.data_ptr(), b.data_ptr(), a.numel(), stream()), "residual_add")

def argmax_rows_(logits, out):
    """out[r] = argmax(logits[r]) (fp16 [rows, n] -> int64 [rows]); the greedy sampler of the benchmark step."""
    assert logits.dtype == torch.float16 and logits.dim() == 2 and logits.stride(1) == 1 and out.dtype == torch.int64
    if logits.size(1) < 8 or logits.stride(0) % 8 != 0:      # shapes the kernel's 16-byte rows cannot take (e.g. V = 32001)
        torch.argmax(logits, dim=1, out=out)
        return
    _check(_lib.qs_argmax_rows(logits.data_ptr(), out.data_ptr(), logits.size(0), logits.size(1), logits.stride(0), stream()),
           "argmax_rows")


LLAMA3_8B = dict(name="Llama-3-8B", hidden=4096, heads=32, kv_heads=8, inter=14336, layers=32, vocab=128256,
                 rope_theta=5e5, eps=1e-5)
QWEN15_72B = dict(name="Qwen1.5-72B", hidden=8192, heads=64, kv_heads=64, inter=24576, layers=80, vocab=152064,
                  rope_theta=1e6, eps=1e-6, qkv_bias=True)     # attention_bias: q/k/v projections carry a bias
LLAMA2_7B = dict(name="Llama-2-7B", hidden=4096, heads=32, kv_heads=32, inter=11008, layers=32, vocab=32000,
                 rope_theta=1e4, eps=1e-5)
LLAMA2_70B = dict(name="Llama-2-70B", hidden=8192, heads=64, kv_heads=8, inter=28672, layers=80, vocab=32000,
                  rope_theta=1e4, eps=1e-5)
TINY = dict(name="tiny-llama", hidden=256, heads=4, kv_heads=2, inter=512, layers=2, vocab=512, rope_theta=1e4,
            eps=1e-5)


class W4A8Linear:
    """Stand-in for W4A8OF16LinearDynamicInputScale (w4a8_linear.py:12-134): same buffers, same call, same bias rule
    (`output_buffer += bias` after the op, :116-118 / :132-134).  Built either from synthetic random tensors or from
    checkpoint tensors (`from_tensors`, fed by qserve_amd.loader).  `defer_bias`: row-parallel shard under tensor
    parallelism - the caller adds the bias once, after the all-reduce."""

    def __init__(self, n, k, group_size, device, gen, bias=False):
        self.n, self.k, self.group_size = n, k, group_size
        self.defer_bias = False
        self.qweight = torch.randint(-128, 128, (n, k // 2), dtype=torch.int8, device=device, generator=gen)
        self.s1_scales = (torch.rand((n,), device=device, generator=gen) * 0.004 + 0.001).half()
        if group_size == -1:
            z = torch.randint(0, 16, (n,), device=device, generator=gen).half()
            self.s1_szeros = (z * self.s1_scales).half()
        else:
            # small level-2 scales keep q*s2 <= 255; any bytes are well-defined input for the kernel
            self.s2_scales = torch.randint(1, 9, (k // 128, n), dtype=torch.int8, device=device, generator=gen)
            zz = torch.randint(0, 16, (k // 128, n), device=device, generator=gen).to(torch.int16)
            self.s2_zeros = (-(zz * self.s2_scales.to(torch.int16))).to(torch.int8)
        self.bias = ((torch.rand((n,), device=device, generator=gen) - 0.5) * 0.1).half() if bias else None

    @classmethod
    def from_tensors(cls, d, group_size, defer_bias=False):
        """d: {"qweight", "s1_scales", "s1_szeros" | ("s2_scales", "s2_zeros"), ["bias"]} already on the device."""
        self = cls.__new__(cls)
        self.n, self.k, self.group_size = d["qweight"].shape[0], d["qweight"].shape[1] * 2, group_size
        self.qweight, self.s1_scales = d["qweight"], d["s1_scales"]
        if group_size == -1:
            self.s1_szeros = d["s1_szeros"]
        else:
            self.s2_scales, self.s2_zeros = d["s2_scales"], d["s2_zeros"]
            assert tuple(self.s2_scales.shape) == (self.k // 128, self.n)
        self.bias = d.get("bias")
        self.defer_bias = defer_bias
        return self

    def __call__(self, x, input_scales, input_sum, out):
        if self.group_size == -1:   # forward_per_chn, w4a8_linear.py:105-118
            gemm_chn.gemm_forward_cuda(x, self.qweight, self.s1_scales, input_scales, self.s1_szeros, input_sum, out)
        else:                       # forward_per_group, :120-134
            gemm_grp.gemm_forward_cuda(x, self.qweight, self.s2_zeros, self.s2_scales, self.s1_scales, input_scales,
                                       out)
        if self.bias is not None and not self.defer_bias:
            out += self.bias

    def planes_slices(self, tokens):
        """K slices of the planes form of this projection at `tokens` rows (0: not available / has a bias: run the pair)."""
        if self.bias is not None:
            return 0
        return fusedmod.gemm_planes_plan(tokens, self.n, self.k, self.group_size != -1)

    def planes(self, x, planes):
        """The GEMM as K-slice planes (int32 [k_slices, T, n]); its epilogue runs inside the row kernel that follows
        (`add_norm_quant_planes`)."""
        if self.group_size == -1:
            fusedmod.gemm_planes(x, self.qweight, planes)
        else:
            fusedmod.gemm_planes(x, self.qweight, planes, self.s2_zeros, self.s2_scales)

    def add_norm_quant_planes(self, out, hidden, planes, input_scales, input_sum, weight, scaling, eps, out_sum):
        """hidden += this projection's output (from its planes) ; out / scaling (/ out_sum) = rms_norm_general(hidden)."""
        if self.group_size == -1:
            fusedmod.add_residual_rms_norm_general_planes(out, hidden, planes, self.s1_scales, input_scales, weight, scaling, eps,
                                                          w_szs=self.s1_szeros, a_ssums=input_sum, input_sum=out_sum)
        else:
            fusedmod.add_residual_rms_norm_general_planes(out, hidden, planes, self.s1_scales, input_scales, weight, scaling, eps,
                                                          input_sum=out_sum)

    def silu_mul(self, x, input_scales, input_sum, out_act, tmp):
        """gate_up projection + silu_and_mul as one op (qserve_amd.fused.gemm_silu_and_mul_*): out_act [T, n/2].  Only for
        a stacked gate_up weight without bias (the bias would have to be added between the two ops)."""
        assert self.bias is None
        if self.group_size == -1:
            fusedmod.gemm_silu_and_mul_per_chn(x, self.qweight, self.s1_scales, input_scales, self.s1_szeros, input_sum,
                                               out_act, tmp)
        else:
            fusedmod.gemm_silu_and_mul_per_group(x, self.qweight, self.s2_zeros, self.s2_scales, self.s1_scales,
                                                 input_scales, out_act, tmp)


class DecodeEngine:
    def __init__(self, cfg, batch, prompt_len, max_new, group_size=-1, int4_kv=True, device="cuda:0", seed=0,
                 tp_rank=0, tp_world=1, with_lm_head=True, fuse_pairs=True, weights=None, vocab_parallel=True, planes=None,
                 direct_allreduce=None):
        """weights: None = synthetic random-quantised tensors of the right shapes; otherwise this rank's tensors as
        qserve_amd.loader.load_llama_w4a8 returns them (checkpoint path)."""
        self.cfg, self.B, self.dev = cfg, batch, torch.device(device)
        # fuse_pairs: issue (residual add + layer norm) and (silu_and_mul + quant) as one launch each
        # (qserve_amd/fused.py: bit-identical to the op pairs; False = the reference's exact op-by-op sequence)
        self.fuse_pairs = fuse_pairs
        self.tp_rank, self.tp_world = tp_rank, tp_world
        self.group_size, self.int4 = group_size, int4_kv
        H, Hkv = cfg["heads"], cfg["kv_heads"]
        # KV heads: Hkv / tp per rank, or - beyond one rank per KV head - ONE head replicated over tp / Hkv neighbouring ranks
        # (the loader's rule, qserve_amd/loader.py: rank r then holds KV head r // (tp / Hkv))
        assert H % tp_world == 0 and (Hkv % tp_world == 0 or tp_world % Hkv == 0) and \
            cfg["inter"] % (tp_world * 128) == 0, \
            f"tp_world={tp_world} must divide heads={H}, divide or be a multiple of kv_heads={Hkv}, and divide inter/128={cfg['inter'] // 128}"
        self.H, self.Hkv = H // tp_world, max(1, Hkv // tp_world)
        hid, inter = cfg["hidden"], cfg["inter"] // tp_world
        self.hid, self.inter = hid, inter
        self.qkv_n = (self.H + 2 * self.Hkv) * 128
        gen = torch.Generator(device=self.dev).manual_seed(seed + 1000 * tp_rank)
        self.layers = []
        self.with_lm_head = with_lm_head
        if weights is None:
            for _ in range(cfg["layers"]):
                self.layers.append(dict(
                    ln1=(torch.rand((hid,), device=self.dev, generator=gen) + 0.5).half(),
                    ln2=(torch.rand((hid,), device=self.dev, generator=gen) + 0.5).half(),
                    qkv=W4A8Linear(self.qkv_n, hid, group_size, self.dev, gen, bias=bool(cfg.get("qkv_bias"))),
                    o=W4A8Linear(hid, self.H * 128, group_size, self.dev, gen),
                    gate_up=W4A8Linear(2 * inter, hid, group_size, self.dev, gen),
                    down=W4A8Linear(hid, inter, group_size, self.dev, gen),
                ))
            self.norm_w = (torch.rand((hid,), device=self.dev, generator=gen) + 0.5).half()
            if with_lm_head:
                self.embed = (torch.randn((cfg["vocab"], hid), device=self.dev, generator=gen) * 0.05).half()
                self.lm_head = (torch.randn((cfg["vocab"], hid), device=self.dev, generator=gen) * 0.02).half()
        else:
            to = lambda d: {k: v.to(self.dev).contiguous() for k, v in d.items()}   # noqa: E731
            for L in weights["layers"]:
                self.layers.append(dict(
                    ln1=L["ln1"].to(self.dev), ln2=L["ln2"].to(self.dev),
                    qkv=W4A8Linear.from_tensors(to(L["qkv"]), group_size),
                    o=W4A8Linear.from_tensors(to(L["o"]), group_size, defer_bias=tp_world > 1),
                    gate_up=W4A8Linear.from_tensors(to(L["gate_up"]), group_size),
                    down=W4A8Linear.from_tensors(to(L["down"]), group_size, defer_bias=tp_world > 1),
                ))
                assert self.layers[-1]["qkv"].n == self.qkv_n and self.layers[-1]["down"].k == inter
            self.norm_w = weights["norm"].to(self.dev)
            if with_lm_head:
                self.embed, self.lm_head = weights["embed"].to(self.dev), weights["lm_head"].to(self.dev)
        # ---- paged KV pools (cache_engine.py:59-115) and pointer tables (model_runner.py:396-414)
        self.max_len = prompt_len + max_new
        self.mb = (self.max_len + 63) // 64 + 1            # README.md:369 page budget rule (+1 page)
        dhb = 64 if int4_kv else 128
        self.size_per_token = self.Hkv * dhb
        self.page_bytes = self.Hkv * 64 * dhb + 64 * self.Hkv * 4
        nblocks = batch * self.mb
        perm = torch.randperm(nblocks, generator=torch.Generator().manual_seed(seed)).reshape(batch, self.mb)
        self.pools, self.tables = [], []
        for _ in range(cfg["layers"]):
            kp = torch.zeros((nblocks, self.page_bytes), dtype=torch.uint8, device=self.dev)
            vp = torch.zeros((nblocks, self.page_bytes), dtype=torch.uint8, device=self.dev)
            t = torch.empty((batch, 2, self.mb), dtype=torch.int64)
            t[:, 0] = kp.data_ptr() + perm * self.page_bytes
            t[:, 1] = vp.data_ptr() + perm * self.page_bytes
            self.pools.append((kp, vp))
            self.tables.append(t.to(self.dev))
        # ---- activation buffers (ActivationBuffer, input_metadata.py:71-109)
        B = batch
        f16, i8 = torch.float16, torch.int8
        self.hidden = torch.zeros((B, hid), dtype=f16, device=self.dev)
        self.q_act = torch.empty((B, hid), dtype=i8, device=self.dev)
        self.q_attn = torch.empty((B, self.H * 128), dtype=i8, device=self.dev)
        self.q_mlp = torch.empty((B, inter), dtype=i8, device=self.dev)
        self.q_scale = torch.empty((B,), dtype=f16, device=self.dev)
        self.q_sum = torch.empty((B,), dtype=f16, device=self.dev)
        self.qkv_buf = torch.empty((B, self.qkv_n), dtype=f16, device=self.dev)
        # row-parallel partial output / its sum over the ranks.  With the library's direct all-reduce (tp.DirectAllReduce)
        # the GEMMs write straight into the communicator's input buffer and the sum appears in its output buffer
        self.ar = direct_allreduce
        if self.ar is not None:
            assert tp_world > 1 and (B * hid) % (8 * tp_world) == 0
            self.proj_out, self.proj_res = self.ar.input((B, hid)), self.ar.output((B, hid))
        else:
            self.proj_out = torch.empty((B, hid), dtype=f16, device=self.dev)
            self.proj_res = self.proj_out
        self.gate_up_buf = torch.empty((B, 2 * inter), dtype=f16, device=self.dev)
        self.mlp_act = torch.empty((B, inter), dtype=f16, device=self.dev)
        self.final = torch.empty((B, hid), dtype=f16, device=self.dev)
        # K-slice planes (qserve_amd.fused.gemm_planes): the row-parallel projections named in `planes` (default: QS_PLANES or
        # "down") leave int32 partial sums per K slice and the add + norm + quant launch behind them finishes the GEMM - where
        # the pair fusions are on, on one GPU, for shapes the library has such a launch for
        if planes is None:
            planes = tuple(x for x in os.environ.get("QS_PLANES", "down").split(",") if x)
        bad = [x for x in planes if x not in ("o", "down")]
        if bad:
            raise ValueError(f"planes / QS_PLANES: {bad} - only the row-parallel projections 'o' and 'down' have a planes form")
        self.planes = {}
        if fuse_pairs and tp_world == 1 and hid <= 4096 and hid % 2048 == 0:
            for name in planes:
                # one buffer serves every layer: the form is taken only if EVERY layer's projection has it with the same number
                # of slices (a layer with a bias, or of another shape, would otherwise run the planes path and lose its bias)
                ks = {layer[name].planes_slices(B) for layer in self.layers}
                if len(ks) == 1 and 0 not in ks:
                    self.planes[name] = torch.empty((ks.pop(), B, hid), dtype=torch.int32, device=self.dev)
        self.lengths = torch.full((B,), prompt_len, dtype=torch.int32, device=self.dev)   # context incl. new token
        self.tokens = torch.randint(0, cfg["vocab"], (B,), device=self.dev, generator=gen)
        # Tensor parallel: the (un-quantised) lm_head is cut over the vocabulary - rank r multiplies rows [v0, v0 + V/N)
        # only and the ranks exchange one greedy candidate per sequence (value, global index) through the SAME fp16 sum
        # all-reduce as the row-parallel partials: every rank fills its own [B, 8] slot of a zeroed [N, B, 8] tensor
        # (index split into two fp16-exact integers < 2048), so the sum is an all-gather, exact.  A replicated head made
        # every rank stream the whole 1 GB matrix for the global batch: 0.52 ms of a 3.8 ms step at N = 8.
        V = cfg["vocab"]
        self.vocab_parallel = bool(with_lm_head and tp_world > 1 and vocab_parallel and V % tp_world == 0 and
                                   V // tp_world >= 8 and (V // tp_world) % 8 == 0 and V < (1 << 22) and 8 * tp_world <= hid)
        if self.vocab_parallel:
            self.v0 = tp_rank * (V // tp_world)
            self.lm_head = self.lm_head[self.v0:self.v0 + V // tp_world].contiguous()
            self.head_idx = torch.zeros((B,), dtype=torch.int64, device=self.dev)
            n = tp_world * B * 8
            self.head_cand = self.proj_out.view(-1)[:n].view(tp_world, B, 8)      # what this rank contributes
            self.head_cand_res = self.proj_res.view(-1)[:n].view(tp_world, B, 8)  # the sum over the ranks
        self.graph = None
        self.pieces = None

    # ---- fill the cache for positions [0, prompt_len) through the prefill writer (random K/V source) ----------
    def prefill_cache(self, prompt_len, chunk=8):
        B = self.B
        gen = torch.Generator(device=self.dev).manual_seed(99)
        seq = torch.full((B,), prompt_len, dtype=torch.int32, device=self.dev)
        cu = (torch.arange(0, B + 1, device=self.dev, dtype=torch.int32) * prompt_len)
        pad = fused_attention.compute_padding_offsets(cu, prompt_len, B * prompt_len)
        qkv = torch.randn((B * prompt_len, self.qkv_n), dtype=torch.float16, device=self.dev, generator=gen)
        for li in range(self.cfg["layers"]):
            work = qkv if li == 0 else (qkv * (1.0 + 0.01 * li)).half()
            fused_attention.apply_bias_rope_update_kv_cache(
                work, seq, pad, self.tables[li], self.H, self.Hkv, prompt_len, 64, self.size_per_token, 128,
                self.cfg["rope_theta"], 8192, True, self.int4, True)
        self.lengths.fill_(prompt_len + 1)
        torch.cuda.synchronize()

    # ---- real prefill: the reference's is_prompt=True path (llama_w4a8_unpad.py:199-243, 330-361) -------------
    def prefill(self, prompt_len, tokens=None):
        """Run the prompt through the model: per layer  norm+quant -> qkv GEMM -> apply_bias_rope_update_kv_cache
        (RoPE in place + quantised cache write) -> flash_attn_varlen_func (causal) -> quant -> o_proj -> residual+norm
        -> gate_up -> silu_and_mul+quant -> down -> residual.  All sequences have `prompt_len` tokens.  Leaves the
        cache filled, `hidden` = last-token states, `tokens` = first sampled token, lengths = prompt_len + 1."""
        from .flash import flash_attn_varlen_func
        cfg, B, dev = self.cfg, self.B, self.dev
        assert self.with_lm_head and prompt_len + 1 <= self.max_len
        T = B * prompt_len
        f16, i8 = torch.float16, torch.int8
        fuse_sum, fuse = self.group_size == -1, self.fuse_pairs
        if tokens is None:
            tokens = torch.randint(0, cfg["vocab"], (T,), device=dev,
                                   generator=torch.Generator(device=dev).manual_seed(7))
        h = torch.index_select(self.embed, 0, tokens)
        qa = torch.empty((T, self.hid), dtype=i8, device=dev)
        qo = torch.empty((T, self.H * 128), dtype=i8, device=dev)
        q_mlp = torch.empty((T, self.inter), dtype=i8, device=dev)
        q_scale = torch.empty((T,), dtype=f16, device=dev)
        q_sum = torch.empty((T,), dtype=f16, device=dev)
        qkv = torch.empty((T, self.qkv_n), dtype=f16, device=dev)
        proj = torch.empty((T, self.hid), dtype=f16, device=dev)
        gate_up = torch.empty((T, 2 * self.inter), dtype=f16, device=dev)
        mlp_act = torch.empty((T, self.inter), dtype=f16, device=dev)
        seq = torch.full((B,), prompt_len, dtype=torch.int32, device=dev)
        cu = torch.arange(0, B + 1, device=dev, dtype=torch.int32) * prompt_len
        pad = fused_attention.compute_padding_offsets(cu, prompt_len, T)
        sums = q_sum if fuse_sum else None

        def norm_quant(x, w):
            if fuse_sum:
                layernorm_ops.rms_norm_general_fuse_sum(qa, x, w, q_sum, q_scale, cfg["eps"], True)
            else:
                layernorm_ops.rms_norm_general(qa, x, w, q_scale, cfg["eps"], True)

        def add_norm_quant(x, delta, w):
            if fuse:
                fusedmod.add_residual_rms_norm_general(qa, x, delta, w, q_scale, cfg["eps"], sums)
            else:
                residual_add_(x, delta)
                norm_quant(x, w)

        nl = len(self.layers)
        for li, L in enumerate(self.layers):
            if li == 0:
                norm_quant(h, L["ln1"])
            L["qkv"](qa, q_scale, q_sum, qkv)
            fused_attention.apply_bias_rope_update_kv_cache(
                qkv, seq, pad, self.tables[li], self.H, self.Hkv, prompt_len, 64, self.size_per_token, 128,
                cfg["rope_theta"], 8192, True, self.int4, True)
            q, k, v = qkv.split([self.H * 128, self.Hkv * 128, self.Hkv * 128], dim=-1)
            attn = flash_attn_varlen_func(q.reshape(T, self.H, 128), k.reshape(T, self.Hkv, 128),
                                          v.reshape(T, self.Hkv, 128), cu, cu, prompt_len, prompt_len, dropout_p=0.0,
                                          causal=True).reshape(T, -1)
            if fuse_sum:
                fused_kernels.invoke_quant_fuse_sum(qo, attn, q_sum, q_scale)
            else:
                fused_kernels.invoke_quant(qo, attn, q_scale)
            L["o"](qo, q_scale, q_sum, proj)
            tpmod.all_reduce_sum_(proj)
            if L["o"].defer_bias and L["o"].bias is not None:
                proj += L["o"].bias
            add_norm_quant(h, proj, L["ln2"])
            if fuse and L["gate_up"].bias is None:     # gate_up GEMM with the silu * mul epilogue, then the quantiser
                L["gate_up"].silu_mul(qa, q_scale, q_sum, mlp_act, gate_up)
            else:
                L["gate_up"](qa, q_scale, q_sum, gate_up)
            if fuse and L["gate_up"].bias is not None:
                fusedmod.silu_and_mul_quant(q_mlp, gate_up, q_scale, sums)
            else:
                if not fuse:
                    activation_ops.silu_and_mul(mlp_act, gate_up)
                if fuse_sum:
                    fused_kernels.invoke_quant_fuse_sum(q_mlp, mlp_act, q_sum, q_scale)
                else:
                    fused_kernels.invoke_quant(q_mlp, mlp_act, q_scale)
            L["down"](q_mlp, q_scale, q_sum, proj)
            tpmod.all_reduce_sum_(proj)
            if L["down"].defer_bias and L["down"].bias is not None:
                proj += L["down"].bias
            if li + 1 < nl:
                add_norm_quant(h, proj, self.layers[li + 1]["ln1"])
            else:
                residual_add_(h, proj)
        last = (cu[1:] - 1).to(torch.int64)
        torch.index_select(h, 0, last, out=self.hidden)
        layernorm_ops.rms_norm(self.final, self.hidden, self.norm_w, cfg["eps"])
        if self.vocab_parallel:
            tpmod.all_reduce_sum_(self._head_local())                # in place, through the process group
            self._head_finish(self.head_cand)
        else:
            logits = torch.matmul(self.final, self.lm_head.t())
            argmax_rows_(logits, self.tokens)
        self.lengths.fill_(prompt_len + 1)

    # ---- one decode step (llama_w4a8_unpad.py:330-361 per layer) --------------------------------------------
    def _segments(self):
        """The step as a generator: yields the row-parallel partial output wherever tensor parallelism needs its sum
        all-reduce (2 per layer), so that the caller decides how the collective is issued (eagerly between hipGraph
        pieces, inside one graph, or not at all at world size 1)."""
        cfg, B = self.cfg, self.B
        fuse_sum = self.group_size == -1
        if self.with_lm_head:
            torch.index_select(self.embed, 0, self.tokens, out=self.hidden)
        h = self.hidden
        qa, qo = self.q_act, self.q_attn
        fuse = self.fuse_pairs
        sums = self.q_sum if fuse_sum else None

        def norm_quant(x, w):
            if fuse_sum:
                layernorm_ops.rms_norm_general_fuse_sum(qa, x, w, self.q_sum, self.q_scale, cfg["eps"], True)
            else:
                layernorm_ops.rms_norm_general(qa, x, w, self.q_scale, cfg["eps"], True)

        def add_norm_quant(x, delta, w):
            if fuse:
                fusedmod.add_residual_rms_norm_general(qa, x, delta, w, self.q_scale, cfg["eps"], sums)
            else:
                residual_add_(x, delta)
                norm_quant(x, w)

        # (row-parallel GEMM, add + norm + quant) as K-slice planes: the GEMM leaves int32 partial sums per K slice, the row
        # kernel that follows sums them and applies the GEMM's epilogue (bit-identical pair fusion; single GPU only - under
        # tensor parallelism the all-reduce sits between the two)
        def proj_add_norm_quant(lin, name, xq, x, w):
            pl = self.planes.get(name) if fuse and self.tp_world == 1 else None
            if pl is not None:
                lin.planes(xq, pl)
                lin.add_norm_quant_planes(qa, x, pl, self.q_scale, self.q_sum, w, self.q_scale, cfg["eps"], sums)
                return True
            return False

        nl = len(self.layers)
        for li, L in enumerate(self.layers):
            if li == 0:
                norm_quant(h, L["ln1"])
            L["qkv"](qa, self.q_scale, self.q_sum, self.qkv_buf)
            q, k, v = self.qkv_buf.split([self.H * 128, self.Hkv * 128, self.Hkv * 128], dim=-1)
            if fuse:         # attention + invoke_quant(_fuse_sum) of its output in one call (bit-identical pair fusion)
                fusedmod.single_query_attention_quant(
                    q.reshape(B, self.H, 128), k.reshape(B, self.Hkv, 128), v.reshape(B, self.Hkv, 128), self.tables[li],
                    self.lengths, qo, self.q_scale, 8192, 64, self.size_per_token, self.max_len, 128, cfg["rope_theta"],
                    True, self.int4, True, quant_sum=sums)
            else:
                attn = fused_attention.single_query_attention(
                    q.reshape(B, self.H, 128), k.reshape(B, self.Hkv, 128), v.reshape(B, self.Hkv, 128), self.tables[li],
                    self.lengths, None, 8192, 64, self.size_per_token, self.max_len, 128, cfg["rope_theta"], True,
                    self.int4, True)
                attn = attn.reshape(B, -1)
                if fuse_sum:
                    fused_kernels.invoke_quant_fuse_sum(qo, attn, self.q_sum, self.q_scale)
                else:
                    fused_kernels.invoke_quant(qo, attn, self.q_scale)
            if not proj_add_norm_quant(L["o"], "o", qo, h, L["ln2"]):
                L["o"](qo, self.q_scale, self.q_sum, self.proj_out)
                res = self.proj_out
                if self.tp_world > 1:
                    yield self.proj_out
                    res = self.proj_res
                    if L["o"].defer_bias and L["o"].bias is not None:
                        res += L["o"].bias                    # once, after the reduce
                add_norm_quant(h, res, L["ln2"])
            if fuse and L["gate_up"].bias is None:     # gate_up GEMM with the silu * mul epilogue, then the quantiser
                L["gate_up"].silu_mul(qa, self.q_scale, self.q_sum, self.mlp_act, self.gate_up_buf)
            else:
                L["gate_up"](qa, self.q_scale, self.q_sum, self.gate_up_buf)
            if fuse and L["gate_up"].bias is not None:
                fusedmod.silu_and_mul_quant(self.q_mlp, self.gate_up_buf, self.q_scale, sums)
            else:
                if not fuse:
                    activation_ops.silu_and_mul(self.mlp_act, self.gate_up_buf)
                if fuse_sum:
                    fused_kernels.invoke_quant_fuse_sum(self.q_mlp, self.mlp_act, self.q_sum, self.q_scale)
                else:
                    fused_kernels.invoke_quant(self.q_mlp, self.mlp_act, self.q_scale)
            if li + 1 < nl and proj_add_norm_quant(L["down"], "down", self.q_mlp, h, self.layers[li + 1]["ln1"]):
                continue                                             # (next layer's input norm done from the planes)
            L["down"](self.q_mlp, self.q_scale, self.q_sum, self.proj_out)
            res = self.proj_out
            if self.tp_world > 1:
                yield self.proj_out
                res = self.proj_res
                if L["down"].defer_bias and L["down"].bias is not None:
                    res += L["down"].bias
            if li + 1 < nl:
                add_norm_quant(h, res, self.layers[li + 1]["ln1"])   # next layer's input norm
            else:
                residual_add_(h, res)
        layernorm_ops.rms_norm(self.final, h, self.norm_w, cfg["eps"])
        if self.with_lm_head and self.vocab_parallel:
            yield self._head_local()                                 # candidates of the ranks meet in the sum all-reduce
            self._head_finish(self.head_cand_res)
        elif self.with_lm_head:
            logits = torch.matmul(self.final, self.lm_head.t())      # un-quantised fp16 lm_head (:392,476)
            argmax_rows_(logits, self.tokens)                        # greedy sampler
        self.lengths.add_(1)


    def _head_local(self):
        """This rank's greedy candidate per sequence -> its slot of `head_cand` (the other slots zero)."""
        logits = torch.matmul(self.final, self.lm_head.t())          # [B, V/N]
        argmax_rows_(logits, self.head_idx)
        val = logits.gather(1, self.head_idx.unsqueeze(1)).squeeze(1)
        gi = self.head_idx + self.v0
        self.head_cand.zero_()
        row = self.head_cand[self.tp_rank]
        row[:, 0] = val
        row[:, 1] = (gi & 2047).to(torch.float16)
        row[:, 2] = (gi >> 11).to(torch.float16)
        return self.head_cand.view(-1)

    def _head_finish(self, gathered):
        """tokens = the first maximum over the ranks' candidates (ranks own ascending vocabulary ranges, torch.argmax
        returns the first maximal entry: the same tie rule as an argmax over the whole row)."""
        c = gathered.float()                                          # [N, B, 8]
        best = c[:, :, 0].argmax(dim=0)
        sel = c.gather(0, best.view(1, -1, 1).expand(1, c.size(1), 8)).squeeze(0)
        self.tokens.copy_((sel[:, 1] + sel[:, 2] * 2048.0).to(torch.int64))

    def _reduce(self, partial):
        if self.ar is not None:
            self.ar.all_reduce(partial.numel())       # a kernel on the current stream (capturable)
        else:
            tpmod.all_reduce_sum_(partial)

    def step(self):
        for partial in self._segments():
            self._reduce(partial)

    def capture(self, piecewise=None):
        """Capture one step in hipGraph(s) (removes ~400 launches of host overhead per step).

        World size 1: one graph.  Tensor parallel (default `piecewise`): one graph per segment between the all-reduces,
        the collectives themselves are issued eagerly between the replays - nothing depends on the communication
        library supporting stream capture, and the host only issues ~2 calls per layer and rank.
        `piecewise=False` captures the collectives too (needs a capturable backend)."""
        if piecewise is None:
            piecewise = self.tp_world > 1 and self.ar is None   # the direct all-reduce is an ordinary kernel: one graph
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.step()                       # warm-up outside capture (allocator, lazy init)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.pieces = None
        if not piecewise:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            self.graph = g
            return g
        pool = torch.cuda.graph_pool_handle()
        pieces, gen = [], self._segments()
        while True:
            g = torch.cuda.CUDAGraph()
            partial = None
            # thread-local error mode: the process group's watchdog thread may poll its events while a piece is captured
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                try:
                    partial = next(gen)
                except StopIteration:
                    pass
            pieces.append((g, partial))
            if partial is None:
                break
            self._reduce(partial)             # keeps the data flow of the capture pass identical to a real step
        self.pieces = pieces
        self.graph = None
        return pieces

    def check(self):
        """Raise if a bounded in-launch wait gave up since the last call (the library's direct all-reduce waiting for a
        peer): the tensors of that step are undefined.  Synchronises the device - call it once per batch of steps, not
        per step."""
        if self.ar is not None and self.ar.error():
            raise RuntimeError("direct all-reduce: a wait for a peer rank timed out (ranks more than a few seconds apart, or "
                               "a peer died); the communicators' epochs no longer match - re-create them")
        bits = _device_status()                        # K-slice seam of the GEMMs / attention + quant hand-over
        if bits:
            raise RuntimeError(f"a bounded in-launch wait of libqserve_amd gave up (error bits {bits}: 1 = K-slice seam of a W4A8 "
                               "GEMM, 2 = attention + quant hand-over): the tensors of that step are undefined; call "
                               "qserve_amd._lib.lib.qs_device_reset() before the next step")

    def run(self):
        if getattr(self, "pieces", None):
            for g, partial in self.pieces:
                g.replay()
                if partial is not None:
                    self._reduce(partial)
        elif self.graph is not None:
            self.graph.replay()
        else:
            self.step()
