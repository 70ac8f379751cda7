"""The GEMM dispatcher's choices, checked without a GPU through the plan-only entry (qs_w4a8_gemm_plan).  These pin the
measured heuristics of DESIGN.md 5.1 - a change here must come with new measurements (scripts/bench_gemm_shard.py)."""
import pytest

from qserve_amd.plan import attention_plan, gemm_plan


def ring(mt, wn, mb, ks=1):
    return dict(family="ring", m_tiles=mt, units=wn, token_blocks=mb, k_slices=ks)


LLAMA3_DECODE = [   # bs = 64 step of BASELINE configs[1]
    ((64, 6144, 4096), ring(2, 1, 2)),          # qkv: 192 workgroups of 32 tokens x 64 channels
    ((64, 4096, 4096), ring(1, 1, 4)),          # o: 256 workgroups of 16 tokens
    ((64, 28672, 4096), ring(4, 2, 1)),         # gate_up: 224 workgroups of 64 tokens x 128 channels
    ((64, 4096, 14336), ring(2, 1, 2, 2)),      # down: 2 token blocks x 2 K slices meeting in the workspace (measured override of
                                                # the byte model's <2,2> x 4 slices: 14.4-14.7 vs 15.3-15.8 us, round 4)
]


@pytest.mark.parametrize("shape,expect", LLAMA3_DECODE)
def test_llama3_decode_shapes(shape, expect):
    assert gemm_plan(*shape) == expect


def test_small_batches_keep_whole_tokens_per_workgroup():
    assert gemm_plan(16, 6144, 4096) == ring(1, 1, 1)
    assert gemm_plan(16, 28672, 4096) == ring(2, 2, 1)        # padded 32-token tile, one round of 224 workgroups
    assert gemm_plan(16, 4096, 14336) == ring(1, 1, 1, 4)     # under-filled grid: 4 K slices fill the chip
    assert gemm_plan(32, 28672, 4096) == ring(2, 2, 1)


def test_compute_bound_shapes_take_the_tiled_kernel():
    assert gemm_plan(4096, 4096, 4096) == dict(family="tiled", tile_tokens=256)       # configs[0]
    assert gemm_plan(65536, 28672, 4096) == dict(family="tiled", tile_tokens=256)     # prefill gate_up
    # per-group, 256-token tiles: the four-wave kernel of round 5 (gemm_w4a8_wide.hip: one level-2 dequant per weight byte for 256
    # tokens); per-channel the two tiles measure the same and the eight-wave one stays
    assert gemm_plan(4096, 4096, 4096, per_group=True) == dict(family="wide", tile_tokens=256)
    assert gemm_plan(65536, 28672, 4096, per_group=True) == dict(family="wide", tile_tokens=256)
    assert gemm_plan(1024, 4096, 4096) == dict(family="tiled", tile_tokens=128)
    assert gemm_plan(512, 3584, 4096)["family"] == "ring"                              # too few tiles: ring 64 x 128


def test_tiled_kernel_needs_real_tokens():
    """Regression: N/256 >= 192 alone used to select the 256-token tile at M = 64 (every N >= 49152 ran at 2 TB/s)."""
    for N in (49152, 57344, 65536):                          # one round of four-unit workgroups instead of 1.5-2 of two-unit ones
        assert gemm_plan(64, N, 8192) == ring(4, 4, 1)          # (measured 48.9 vs 58.4 us at N = 49152, 55.1 vs 66.1 at 57344)
    assert gemm_plan(191, 57344, 8192)["family"] == "ring"
    assert gemm_plan(192, 57344, 8192) == dict(family="tiled", tile_tokens=256)


def test_k_slices_only_for_under_filled_grids():
    assert gemm_plan(64, 49152, 4096)["k_slices"] == 1
    assert gemm_plan(64, 8192, 8192)["k_slices"] == 1
    assert gemm_plan(64, 8192, 24576) == ring(4, 2, 1, 4)       # Qwen1.5-72B down_proj
    assert gemm_plan(128, 4096, 14336) == ring(4, 2, 2, 4)       # measured best for both modes (19.6 / 25.8 us)


def test_awkward_k_falls_back_or_uses_two_unit_workgroups():
    assert gemm_plan(64, 4096, 11008)["family"] == "splitk"      # Llama-2-7B down_proj: 172 stages, 16-token workgroups win
    assert gemm_plan(64, 5120, 13824)["family"] == "ring"        # Llama-2-13B down_proj: 216 stages = 4 groups x 54
    assert gemm_plan(64, 4096, 512)["family"] == "splitk"        # short K at decode batch: fixed costs
    assert gemm_plan(512, 4096, 1792)["family"] in ("ring", "pair")


@pytest.mark.parametrize("tp", [2, 4, 8])
def test_tensor_parallel_shards_use_the_ring_kernel(tp):
    M = 64 * tp
    for N, K in ((6144 // tp, 4096), (4096, 4096 // tp), (28672 // tp, 4096)):
        assert gemm_plan(M, N, K)["family"] == "ring", (M, N, K)


def test_per_group_follows_the_same_model():
    assert gemm_plan(64, 4096, 14336, per_group=True) == ring(2, 1, 2, 2)     # (19.6-19.9 vs 20.6-20.9 us)
    assert gemm_plan(32, 4096, 14336) == ring(2, 1, 1, 4)                       # M = 32 keeps four slices (11.35 vs 11.87 us)
    # g128 at 65 .. 128 tokens, many channels: ONE 128-token block per workgroup (round 5: the level-2 dequant of a weight byte
    # serves all 128 tokens; 30.6 us against 35.6 for two 64-token blocks of four units); per-channel has nothing to share
    assert gemm_plan(128, 28672, 4096, per_group=True) == ring(8, 2, 1)
    assert gemm_plan(96, 28672, 4096, per_group=True) == ring(8, 2, 1)
    assert gemm_plan(128, 28672, 4096) == ring(4, 4, 2)                     # 224 workgroups, one round
    assert gemm_plan(129, 28672, 4096, per_group=True)["m_tiles"] == 4
    assert gemm_plan(2048, 4096, 4096, per_group=True)["family"] in ("tiled", "ring", "pair")
    # the level-2 dequant is VALU work per streamed weight byte: at equal bytes per CU the geometry with fewer channels per
    # workgroup wins per-group (g128 qkv at M = 128: 16.2 vs 18.3 us), while per-channel keeps (2,2) (12.1 vs 13.9 us)
    assert gemm_plan(128, 6144, 4096, per_group=True) == ring(4, 1, 2)
    assert gemm_plan(128, 6144, 4096) == ring(2, 2, 4)
    assert gemm_plan(64, 6144, 4096, per_group=True) == ring(2, 1, 2)       # unchanged by the per-group term
    assert gemm_plan(64, 28672, 4096, per_group=True) == ring(4, 2, 1)


def test_rejected_shapes_raise():
    with pytest.raises(RuntimeError):
        gemm_plan(64, 4000, 4096)       # N % 64
    with pytest.raises(RuntimeError):
        gemm_plan(64, 4096, 4000)       # K % 128


# ---- decode attention ----------------------------------------------------------------------------------------------
def test_attention_headline_config_is_one_workgroup_per_sequence_and_kv_head():
    assert attention_plan(64, 32, 8, 24, 1536) == dict(family="mfma_kv4", kv_splits=1, waves=8)     # 512 workgroups
    assert attention_plan(128, 32, 8, 24, 1536)["kv_splits"] == 1


def test_attention_small_batches_split_the_context():
    # round 5: the split factor comes from a cost model fitted to a sweep on the MI355X (profiles/round5_split_sweep*.txt):
    # workgroups land on the 256 CUs round-robin and a CU streams at its own request budget, so the model fills the chip ONCE
    # (64 pairs -> 3-4 splits) instead of twice (the round-2 rule: 8 splits, 3-15 % slower at these sizes)
    p = attention_plan(8, 32, 8, 129, 8192, int4_kv_cache=False)        # BASELINE configs[4]: 64 workgroups -> split
    assert p["family"] == "mfma_kv8" and p["waves"] == 4 and p["kv_splits"] == 3
    assert attention_plan(8, 32, 8, 121, 7700)["kv_splits"] == 4       # its KV4 twin: measured 19.7 us (8 splits: 22.7)
    p = attention_plan(1, 32, 8, 129, 8192)
    assert p["family"] == "mfma_kv4" and p["kv_splits"] == 8            # 8 pairs: the merge's cost caps the factor
    assert attention_plan(1, 32, 8, 4, 200)["kv_splits"] == 1          # too short
    assert attention_plan(16, 32, 8, 65, 4096)["kv_splits"] == 2       # 128 pairs x 2 = 256 workgroups
    assert attention_plan(32, 32, 8, 65, 4096)["kv_splits"] == 1       # 256 pairs: one round, no merge (was 2: +11 %)
    assert attention_plan(24, 32, 8, 121, 7700)["kv_splits"] == 1      # 192 pairs: a second workgroup on some CUs doubles their time
    assert attention_plan(48, 32, 8, 121, 7700)["kv_splits"] == 2      # 384 pairs = 1.5 rounds -> 768 = 3 full rounds (-9 %)
    assert attention_plan(48, 32, 8, 17, 1030)["kv_splits"] == 1


def test_attention_wide_page_tables_take_the_valu_kernel():
    assert attention_plan(4, 32, 8, 193, 12000)["family"] == "valu"
    assert attention_plan(4, 32, 8, 192, 12000)["family"] == "mfma_kv4"


def test_attention_rejects_bad_head_counts():
    with pytest.raises(RuntimeError):
        attention_plan(4, 30, 8, 24, 1000)          # heads not a multiple of kv heads
    with pytest.raises(RuntimeError):
        attention_plan(4, 72, 8, 24, 1000)          # group of 9 query heads
