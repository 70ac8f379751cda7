"""Which kernel would the W4A8 GEMM dispatcher pick for a problem?  (`qs_w4a8_gemm_plan`; runs without a GPU.)"""
import ctypes as C

from ._lib import check, lib

FAMILIES = {1: "splitk", 2: "pair", 3: "ring", 4: "tiled", 5: "wide"}


def gemm_plan(M, N, K, per_group=False):
    """-> dict(family=..., **geometry).  ring: m_tiles, units, token_blocks, k_slices; tiled: tile_tokens;
    splitk: m_tiles, waves, slices, xcd_map."""
    buf = (C.c_int * 5)()
    check(lib.qs_w4a8_gemm_plan(int(bool(per_group)), M, N, K, C.cast(buf, C.c_void_p)), "w4a8 gemm plan")
    fam = FAMILIES.get(buf[0], "none")
    if fam == "ring":
        return dict(family=fam, m_tiles=buf[1], units=buf[2], token_blocks=buf[3], k_slices=buf[4])
    if fam in ("tiled", "wide"):
        return dict(family=fam, tile_tokens=32 * buf[1])
    if fam == "splitk":
        return dict(family=fam, m_tiles=buf[1], waves=buf[2], slices=buf[3], xcd_map=bool(buf[4]))
    return dict(family=fam)


ATTN_FAMILIES = {1: "mfma_kv4", 2: "mfma_kv8", 3: "valu"}


def attention_plan(batch, num_heads, num_kv_heads, max_blocks, timestep, int4_kv_cache=True):
    """-> dict(family, kv_splits, waves): the decode attention dispatcher's choice (`qs_attention_plan`)."""
    buf = (C.c_int * 3)()
    check(lib.qs_attention_plan(batch, num_heads, num_kv_heads, max_blocks, timestep, int(bool(int4_kv_cache)),
                                C.cast(buf, C.c_void_p)), "attention plan")
    return dict(family=ATTN_FAMILIES.get(buf[0], "none"), kv_splits=buf[1], waves=buf[2])
