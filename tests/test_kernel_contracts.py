"""Code-generation contracts of the hand-scheduled kernels, checked on the gfx950 assembly hipcc produces (CPU-only:
hipcc cross-compiles; ~1 minute).  These are the properties the measured performance rests on and that a harmless-looking
source edit silently destroys (each one was found the hard way):

* the unit / page loop of the KV4 / KV8 decode attention kernels waits on the vector-memory queue ONLY through the
  hand-placed counted waits - as soon as the compiler's own waitcnt pass sees an LDS-DMA in flight at loop entry it puts
  `s_waitcnt vmcnt(0)` in front of every LDS read, which serialises page fetch and compute;
* no kernel of the decode / prefill hot path uses scratch memory (register spills);
* the occupancy the launch geometry assumes (two 512-thread workgroups per CU for decode attention) is reachable.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qserve_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = {}
    d = tmp_path_factory.mktemp("asm")
    for name in ("attention_mfma", "attention_mfma8", "gemm_w4a8_ring", "gemm_w4a8_tiled", "gemm_w4a8_wide", "flash_prefill"):
        dst = d / (name + ".s")
        r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(dst),
                            os.path.join(CSRC, name + ".hip")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = open(dst).read()
    return out


def kernels(text):
    """{mangled name: body} of every kernel in an assembly file."""
    res = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        res[m.group(1)] = m.group(2)
    return res


def meta(text, name, key):
    m = re.search(re.escape(name) + r".*?;\s*" + key + r":\s*(\d+)", text, re.S)
    return int(m.group(1))


def test_kv4_attention_page_loop_has_only_the_hand_placed_vmcnt_waits(asm):
    ks = {n: b for n, b in kernels(asm["attention_mfma"]).items() if "decode_attention_mfma_kernel" in n}
    product = {n: b for n, b in ks.items() if re.search(r"kernelILi\dELi0E", n)}       # EXP = 0 instantiations, G = 1..8
    assert len(product) == 8
    for name, body in product.items():
        i0, i1 = body.index("QS_LOOP_BEGIN"), body.index("QS_LOOP_END")
        loop = body[i0:i1]
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)
        # loop top: group A of the unit landed - vmcnt(9) while the wave owns a further unit, vmcnt(3) on its last one; then the
        # wait for group B: vmcnt(9) / (6) / (0) by the number of units still to come (scalar branches inside the asm statements)
        assert waits == ["9", "3", "9", "6", "0"], f"{name}: vector-memory waits inside the unit loop: {waits}"
        # between the flag poll and the loop nothing may drain the queue either
        pre = body[body.index("s_sleep"):i0]
        assert "vmcnt(0)" not in pre.split("global_load_lds")[-1], f"{name}: vmcnt(0) between the unit DMA issue and the loop"
        assert "global_load_lds_dwordx4" in body and " nt" in body, "unit DMA must be issued from asm with the nt hint"
        # the request groups the counted waits rely on: 3 VMEM instructions each (1 x 4-byte meta + 2 x 16-byte data)
        assert "flat_load" not in loop and "scratch_" not in loop and "buffer_load" not in loop


def test_kv8_attention_page_loop_has_only_the_hand_placed_vmcnt_waits(asm):
    ks = {n: b for n, b in kernels(asm["attention_mfma8"]).items() if "decode_attention_mfma8_kernel" in n}
    assert len(ks) == 8
    for name, body in ks.items():
        head = body.index("s_waitcnt vmcnt(9)")
        loop = body[head:]
        loop = loop[:loop.index("s_barrier")]
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)
        # loop top vmcnt(9) [K landed], V wait vmcnt(0) (last page) / vmcnt(9); then the drain in front of the merge barrier
        assert waits == ["9", "0", "9", "0"], f"{name}: {waits}"


def test_flash_prefill_key_loop_waits_and_copies(asm):
    """The prefill attention's key loop (round-6 kernel, VAR = 1: flash_fwd_kernelILb?ELi1E): (a) between the loop header and its
    back edge the vector-memory queue is waited on ONLY by the asm `s_waitcnt vmcnt(0)` of tiles_landed() - rounds 2-5 carried
    eight compiler-placed counted waits (for the Q fragments, loaded before the loop) in front of the Q.K^T MFMAs of every tile,
    which also drained the LDS-DMA of the next tile; (b) the loop holds no register copies of the O accumulators (32 v_mov_b64 per
    tile and wave before); (c) both tiles of the unrolled loop hold their 32 MFMAs."""
    ks = {n: b for n, b in kernels(asm["flash_prefill"]).items() if "flash_fwd_kernel" in n}
    new = {n: b for n, b in ks.items() if re.search(r"flash_fwd_kernelILb[01]ELi1E", n)}
    assert len(ks) == 4 and len(new) == 2
    for name, body in new.items():
        lines = body.splitlines()
        head = next(i for i, l in enumerate(lines) if "Inner Loop Header" in l)
        label = lines[head].split(":")[0].strip()
        back = max(i for i, l in enumerate(lines) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b", l))
        loop = lines[head:back]
        ins = _instructions("\n".join(loop))
        compiler_waits = [ops for mn, ops, inside in ins if mn == "s_waitcnt" and "vmcnt" in ops and not inside]
        assert not compiler_waits, f"{name}: compiler-placed vector-memory waits inside the key loop: {compiler_waits}"
        asm_waits = [ops for mn, ops, inside in ins if mn == "s_waitcnt" and "vmcnt" in ops and inside]
        assert asm_waits == ["vmcnt(0)", "vmcnt(0)"], f"{name}: {asm_waits}"
        assert not [1 for mn, _, _ in ins if mn.startswith("v_mov_b64")], f"{name}: accumulator copies inside the key loop"
        assert sum(1 for mn, _, _ in ins if mn.startswith("v_mfma_f32_32x32x16_f16")) == 64
        assert not [1 for mn, _, _ in ins if mn.startswith("scratch_")]


@pytest.mark.parametrize("unit", ["attention_mfma", "attention_mfma8", "gemm_w4a8_ring", "gemm_w4a8_tiled", "gemm_w4a8_wide", "flash_prefill"])
def test_hot_path_kernels_do_not_spill(asm, unit):
    text = asm[unit]
    names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M)
    scratch = [int(x) for x in re.findall(r"; ScratchSize: (\d+)", text)]
    assert len(names) == len(scratch) and names
    bad = [(n, s) for n, s in zip(names, scratch) if s]
    assert not bad, f"kernels with scratch (spills): {bad}"


def test_decode_attention_fits_two_workgroups_per_cu(asm):
    text = asm["attention_mfma"]
    for name in kernels(text):
        if "decode_attention_mfma_kernel" not in name or not re.search(r"kernelILi\dELi0E", name):
            continue
        vg = meta(text, name, "NumVgprs")
        lds = meta(text, name, "LDSByteSize")
        alloc = (vg + 7) // 8 * 8
        assert 512 // alloc >= 4, f"{name}: {vg} VGPRs -> fewer than 4 waves per SIMD (two 8-wave workgroups per CU)"
        assert 2 * lds <= 160 * 1024, f"{name}: {lds} B of LDS per workgroup"


def _instructions(body):
    """(mnemonic, operand text, inside_asm) of every instruction of a kernel body, in order."""
    out, inside = [], False
    for line in body.splitlines():
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inside = True
            continue
        if t.startswith(";;#ASMEND"):
            inside = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        out.append((parts[0], parts[1] if len(parts) > 1 else "", inside))
    return out


def _vregs(tok):
    """register numbers of a v[a:b] / vN operand token"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def test_wide_gemm_owns_the_accumulator_file(asm):
    """gemm_w4a8_wide.hip addresses a[0:255] from inline-asm MFMAs with fixed register numbers.  The compiler must (a) allocate
    the whole accumulator file and leave 256 VGPRs' worth of room for it, (b) never touch an accumulator register itself (a spill
    into the file or a v_accvgpr_* of its own would be silent corruption), (c) never write an MFMA source operand with a VALU
    instruction fewer than two instructions ahead of the MFMA (inside an asm statement nothing pads that hazard), and every stage
    instantiation must contain its 64 MFMAs.  The hazard rule also covers the MFMA that sits INSIDE a DMA statement behind the
    s_add_u32 that sets M0 (an SALU instruction: no VGPR hazard)."""
    text = asm["gemm_w4a8_wide"]
    ks = {n: b for n, b in kernels(text).items() if "w4a8_gemm_wide" in n}
    assert len(ks) == 6
    for name, body in ks.items():
        assert meta(text, name, "NumAgprs") == 256, name
        assert meta(text, name, "NumVgprs") <= 200, name
        ins = _instructions(body)
        mf = [i for i, (op, _, _) in enumerate(ins) if op.startswith("v_mfma")]
        # 12 stage instantiations (first pair, six unrolled slots, run-time-slot pair, drain pair) of 64 MFMAs; the stages with
        # run-time prefetch conditions carry the MFMA of a DMA m-tile twice (wrapped into the DMA statement / plain)
        assert 12 * 64 <= len(mf) <= 12 * 64 + 6 * 11, f"{name}: {len(mf)} MFMAs"
        for i, (op, args, inside) in enumerate(ins):
            if not inside:
                assert not op.startswith("v_accvgpr") and not re.search(r"\ba\[?\d", args), \
                    f"{name}: compiler-generated accumulator access: {op} {args}"
            else:
                assert not op.startswith("v_accvgpr_write") and not op.startswith("v_accvgpr_mov"), (name, op, args)
        for i in mf:
            op, args, inside = ins[i]
            assert inside, f"{name}: an MFMA outside inline asm"
            toks = [t.strip() for t in args.split(",")]
            assert re.fullmatch(r"a\[\d+:\d+\]", toks[0]) and toks[3] in (toks[0], "0"), (name, args)
            srcs = _vregs(toks[1]) | _vregs(toks[2])
            assert len(srcs) == 8, (name, args)
            for j in (i - 1, i - 2):
                pop, pargs, _ = ins[j]
                if pop.startswith("v_") and not pop.startswith("v_mfma") and not pop.startswith("v_cmp"):
                    dst = _vregs(pargs.split(",")[0].strip())
                    assert not (dst & srcs), f"{name}: {pop} {pargs} writes a source of the MFMA {args} {i - j} instruction(s) later"
