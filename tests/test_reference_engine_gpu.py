"""The reference's OWN, UNCHANGED engine on the MI355X.

`scripts/stage_reference.sh` copies /root/reference/{qserve, qserve_benchmark.py} byte for byte into the git-ignored
oracle/_ref/ (gpurun ships git-ignored files, so the tree reaches the GPU box; it is test infrastructure and nothing in the
product imports it).  `scripts/run_reference_engine.py --mode ragged` then stands the engine up exactly as the reference's
drivers do - EngineArgs -> LLMEngine.from_engine_args (config.py, arg_utils.py, worker.py, model_runner.py, cache_engine.py,
scheduler.py, block_manager.py) -> add_request -> engine.step() until done (qserve_benchmark.py:40-67, llm_engine.py:525) -
over a small Llama checkpoint in the reference's own format, with an IN-FLIGHT ragged batch: five prompts of 5 / 64 / 65 /
150 / 31 tokens that generate 9 / 3 / 12 / 5 / 12 tokens, so that sequences leave the batch at different steps, block
tables are padded (model_runner.py:386-392) and the decode batch shrinks while it runs.

Checked here:
  * the run completes over the COMPILED extension (`qserve_backend_ext.install()`, the form of the boundary a maintainer of
    the reference would link) and over the ctypes mirror, each in its own process;
  * both produce the same sampled token at every step of every sequence (the engine's scheduling included);
  * every logit the unchanged model's lm_head produced is finite;
  * the batch really was ragged: prompt step of 5 sequences, then a shrinking decode batch, requests finishing at the
    steps their generation lengths dictate.
Skipped where oracle/_ref/ is absent (a checkout that never ran the staging script).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
staged = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "qserve", "engine")),
                            reason="oracle/_ref/ not staged (scripts/stage_reference.sh, authoring container)")


def run_engine(*flags):
    env = dict(os.environ)
    env.pop("QS_AMD_LIBRARY", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_reference_engine.py"), *flags],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, f"reference engine run failed:\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _record(name, rec):
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        fn = os.path.join(d, "round4_reference_engine.json")
        allrec = json.load(open(fn)) if os.path.exists(fn) else {}
        allrec[name] = rec
        json.dump(allrec, open(fn, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


@staged
def test_staged_reference_is_the_reference_byte_for_byte():
    """The staged tree carries the digests taken from /root/reference when it was staged; nothing edited it since."""
    import hashlib
    sums = open(os.path.join(REF, "SHA256SUMS")).read().split("\n")
    n = 0
    for line in sums:
        if not line.strip():
            continue
        digest, rel = line.split()
        assert hashlib.sha256(open(os.path.join(REF, rel), "rb").read()).hexdigest() == digest, rel
        n += 1
    assert n >= 40


@staged
@pytest.mark.parametrize("group_size", [-1, 128], ids=["per_chn", "g128"])
def test_unchanged_reference_engine_ragged_inflight_batch(gpu, group_size):
    ext = run_engine("--mode", "ragged", "--backend", "ext", "--group-size", str(group_size))
    mir = run_engine("--mode", "ragged", "--backend", "ctypes", "--group-size", str(group_size))
    assert "qserve_backend_ext" in ext["backend_module"] and "qserve_amd/backend" in mir["backend_module"], \
        (ext["backend_module"], mir["backend_module"])
    for rec in (ext, mir):
        assert rec["all_logits_finite"] and rec["lm_head_calls"] == rec["engine_steps"]
        gens = rec["generation_lengths"]
        # one prompt step for all five, then decode steps with a shrinking batch; a request leaves after its last token
        assert rec["engine_steps"] == max(gens)
        assert rec["batch_size_per_step"][0] == 5 and rec["batch_size_per_step"][-1] == sum(g == max(gens) for g in gens)
        assert rec["batch_size_per_step"] == [sum(g >= s for g in gens) for s in range(1, max(gens) + 1)]
        assert rec["finished_at_step"] == {str(i): g for i, g in enumerate(gens)}
        assert [len(t) for t in rec["sampled_tokens_per_step"]] == rec["batch_size_per_step"]
    assert ext["sampled_tokens_per_step"] == mir["sampled_tokens_per_step"], "compiled extension and ctypes mirror disagree"
    _record(f"ragged_{'per_chn' if group_size == -1 else 'g128'}",
            dict(engine_steps=ext["engine_steps"], batch_size_per_step=ext["batch_size_per_step"],
                 prompts=ext["prompts"], generation_lengths=ext["generation_lengths"],
                 sampled_tokens_per_step=ext["sampled_tokens_per_step"], all_logits_finite=True,
                 ext_equals_ctypes_mirror=True, transformers_compat=ext["transformers_compat"]))
