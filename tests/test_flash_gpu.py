"""GPU parity of the prefill attention provider (flash_attn_varlen_func shim) vs the float64 oracle."""
import numpy as np
import pytest
import torch

from _helpers import dev
from oracle import flash as oflash

pytestmark = pytest.mark.gpu
TOL = 2e-3   # fp16 inputs / probabilities rounded to fp16 for the matrix cores; outputs are O(1)


@pytest.fixture(params=[0, 1], ids=["round6", "rounds2to5"], autouse=True)
def flash_variant(request):
    """Every case runs through both key loops of the provider (include/qserve_amd.h qs_debug_flash_variant): the default
    block-pipelined one with the lazy running maximum (round 6) and the per-tile loop of rounds 2-5."""
    from qserve_amd._lib import lib
    assert lib.qs_debug_flash_variant(request.param) == 0
    yield request.param
    lib.qs_debug_flash_variant(0)


def _case(gpu, lens_q, lens_k, H, Hkv, causal, seed, packed=True):
    from flash_attn.flash_attn_interface import flash_attn_varlen_func
    r = np.random.default_rng(seed)
    Tq, Tk = int(sum(lens_q)), int(sum(lens_k))
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lens_k)]).astype(np.int32)
    if packed and Tq == Tk:
        # q, k, v as strided views of one packed qkv buffer, exactly as the reference passes them
        qkv = dev(r.standard_normal((Tq, (H + 2 * Hkv) * 128)).astype(np.float16))
        q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
        q, k, v = q.reshape(Tq, H, 128), k.reshape(Tk, Hkv, 128), v.reshape(Tk, Hkv, 128)
    else:
        q = dev(r.standard_normal((Tq, H, 128)).astype(np.float16))
        k = dev(r.standard_normal((Tk, Hkv, 128)).astype(np.float16))
        v = dev(r.standard_normal((Tk, Hkv, 128)).astype(np.float16))
    out = flash_attn_varlen_func(q, k, v, dev(cu_q), dev(cu_k), int(max(lens_q)), int(max(lens_k)), dropout_p=0.0,
                                 causal=causal)
    assert out.shape == (Tq, H, 128) and out.dtype == torch.float16 and out.is_contiguous()
    ref = oflash.attention_varlen(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), cu_q, cu_k, causal=causal)
    err = np.abs(out.cpu().numpy().astype(np.float32) - ref)
    assert np.isfinite(out.cpu().numpy()).all()
    assert err.max() <= TOL, f"max abs err {err.max():.2e}"


@pytest.mark.parametrize("H,Hkv", [(8, 2), (4, 4), (8, 1)])
def test_causal_ragged_batch(gpu, H, Hkv):
    # lengths around the 64-key tile and 128-row workgroup boundaries, incl. a 1-token sequence
    _case(gpu, [1, 63, 64, 65, 127, 128, 129, 300], [1, 63, 64, 65, 127, 128, 129, 300], H, Hkv, True, seed=H + Hkv)


def test_full_attention_and_unequal_lengths(gpu):
    _case(gpu, [5, 130, 64], [40, 200, 64], 4, 2, False, seed=3, packed=False)          # cross lengths, no mask
    _case(gpu, [5, 130, 64], [40, 200, 64], 4, 2, True, seed=4, packed=False)           # bottom-right aligned causal


def test_llama_shape_prefill(gpu):
    _case(gpu, [1024, 700], [1024, 700], 32, 8, True, seed=9)


def test_config5_prompt_8k_sampled_rows(gpu):
    """BASELINE config 5's prompt: one 8 192-token sequence next to a 5 000-token one, Llama-3-8B heads, causal, q / k / v as
    views of the packed qkv buffer (llama_w4a8_unpad.py:232-242).  The float64 oracle on sampled (query row, head) pairs - rows
    at both ends, around the 64-key tile and 128-row workgroup boundaries, deep into the sequence - within 2e-3; every output
    finite; and a size-independent property over ALL rows: the first row of each sequence attends to one key, so it equals v[0]."""
    from flash_attn.flash_attn_interface import flash_attn_varlen_func
    H, Hkv = 32, 8
    lens = [8192, 5000]
    T = sum(lens)
    g = torch.Generator(device=gpu).manual_seed(58)
    qkv = torch.randn((T, (H + 2 * Hkv) * 128), generator=g, device=gpu, dtype=torch.float16)
    q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
    q, k, v = q.reshape(T, H, 128), k.reshape(T, Hkv, 128), v.reshape(T, Hkv, 128)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    out = flash_attn_varlen_func(q, k, v, dev(cu), dev(cu), max(lens), max(lens), dropout_p=0.0, causal=True)
    torch.cuda.synchronize()
    assert out.shape == (T, H, 128) and torch.isfinite(out.float()).all()
    rows = [0, 1, 63, 64, 127, 128, 129, 2999, 3000, 4095, 4096, 6000, 8127, 8190, 8191,
            8192, 8192 + 1, 8192 + 2999, 8192 + 4096, 8192 + 4999]
    heads = [0, 5, 17, 31]
    ref = oflash.attention_rows(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), cu, cu, rows, heads)
    got = out[torch.tensor(rows, device=gpu)][:, torch.tensor(heads, device=gpu)].cpu().numpy().astype(np.float32)
    err = np.abs(got - ref)
    assert err.max() <= TOL, f"max abs err {err.max():.2e} at {np.unravel_index(err.argmax(), err.shape)}"
    for b in range(2):
        t0 = int(cu[b])
        assert torch.equal(out[t0].reshape(Hkv, H // Hkv, 128), v[t0][:, None, :].expand(Hkv, H // Hkv, 128))


def test_rejects_unsupported(gpu):
    from flash_attn.flash_attn_interface import flash_attn_varlen_func
    q = torch.zeros((4, 2, 128), dtype=torch.float16, device=gpu)
    cu = torch.tensor([0, 4], dtype=torch.int32, device=gpu)
    with pytest.raises(NotImplementedError):
        flash_attn_varlen_func(q, q, q, cu, cu, 4, 4, dropout_p=0.1)
    with pytest.raises(NotImplementedError):
        flash_attn_varlen_func(q[:, :, :64].contiguous(), q[:, :, :64].contiguous(), q[:, :, :64].contiguous(), cu, cu, 4, 4)
    with pytest.raises(RuntimeError):
        flash_attn_varlen_func(q.float(), q, q, cu, cu, 4, 4)
