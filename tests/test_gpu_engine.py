"""The one-sequence persistent decode step (csrc/decode_engine.hip) against the CPU oracle and against the multi-launch
HIP path it replaces at batch 1.

Reference semantics: swiftllm/worker/model.py:228-249 (layer loop) over swiftllm/worker/layers/transformer_layer.py:31-130.
The engine keeps the reference's rounding points in both dtypes (no deferred norm), so it is held to oracle/ref_model.py
(exact scores) at the bar of tests/test_gpu_parity_fullwidth.py: logits within 3 ulps of the row scale, a greedy id may
differ only where the oracle's own top-2 gap is within twice that row's logit distance. Geometries: a small one the engine
is laid out for (hidden 2048, 16 q / 8 kv heads of 128, FFN 2048, 3 layers) with contexts that cross every split boundary
(1 ... 51 tokens: most of the 32 context splits empty; ~600 tokens: ragged last split), and the Llama-3-8B width is covered by
tests/test_gpu_parity_fullwidth.py (its `decode_engine` variant at configs1).
Every run is wrapped in pytest-timeout: the kernel's waits are bounded (50 ms), a hang here is a bug, not a stall.
"""
import pytest
import torch

from oracle import synth
from oracle.ref_model import RefLlamaModel

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

SMALL = dict(num_hidden_layers=3, hidden_size=2048, num_attention_heads=16, num_key_value_heads=8,
             intermediate_size=2048, vocab_size=1024, max_position_embeddings=2048, rope_theta=500000.0)


def _ulp(x, dtype):
    mant = 10 if dtype == torch.float16 else 7
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14))) - mant)


def _kw(dtype, **opts):
    return dict(use_dummy=False, block_size=16, gpu_mem_utilization=0.5, num_cpu_blocks=0, max_seqs_in_block_table=8,
                max_blocks_per_seq=64, max_batch_size=4, max_tokens_in_batch=1024, dtype=dtype, **opts)


def _run(model, prompt, steps, forced=None):
    model.post_layer.logits_tap = []
    tap = model.post_layer.logits_tap
    toks, logits = [model.forward([prompt], [0], [])], [tap[-1].float().cpu()]
    n = len(prompt)
    for s in range(steps):
        n += 1
        feed = forced[s] if forced is not None else toks[-1]
        toks.append(model.forward([[feed[0]]], [0], [n]))
        logits.append(tap[-1].float().cpu())
    return toks, logits


def _check(toks, logits, want_toks, want_logits, tdtype, what, ulp_bar=3.0):
    worst = 0.0
    for s, (a, b) in enumerate(zip(logits, want_logits)):
        d = (a - b).abs()
        row = _ulp(b.abs().amax(dim=1, keepdim=True), tdtype)
        worst = max(worst, float((d / row).max()))
        if toks[s] != want_toks[s]:
            top2 = b[0].topk(2).values
            assert float(top2[0] - top2[1]) <= 2 * float(d.max()), (what, s, toks[s], want_toks[s])
    assert worst <= ulp_bar, f"{what}: {worst:.2f} ulps of the row scale"
    return worst


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("prompt_len,steps", [(1, 50), (590, 24)], ids=["ctx_1_to_51", "ctx_590_to_614"])
def test_engine_steps_match_the_oracle_and_the_multi_launch_path(tmp_path, dtype, prompt_len, steps):
    from swiftllm_amd import EngineConfig, LlamaModel, LlamaModelConfig
    tdtype = torch.float16 if dtype == "float16" else torch.bfloat16
    cfg = synth.make_config(**SMALL)
    sd = synth.make_state_dict(cfg, seed=5, dtype=tdtype)
    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(0, cfg["vocab_size"], (prompt_len,), generator=g).tolist()
    kw = _kw(dtype)

    teacher = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(model_path="", **kw), sd, tdtype, score_dtype="fp32")
    teacher.init_kvcache_and_swap(64)
    want_toks, want_logits = [teacher.forward([prompt], [0], [])], [teacher.last_logits.clone()]
    n = prompt_len
    for _ in range(steps):
        n += 1
        want_toks.append(teacher.forward([[want_toks[-1][0]]], [0], [n]))
        want_logits.append(teacher.last_logits.clone())
    del teacher

    synth.write_model_dir(str(tmp_path), cfg, sd)
    results = {}
    for name, opts in (("engine_graph", dict(tuning=dict(decode_engine=True))),
                       ("engine_eager", dict(use_hip_graph=False, tuning=dict(decode_engine=True))), ("multi_launch", dict())):
        model = LlamaModel(EngineConfig(model_path=str(tmp_path), **_kw(dtype, **opts)))
        model.load_weights()
        model.init_kvcache_and_swap(64)
        assert (model._engine is not None) == (name != "multi_launch"), "the engine must take this geometry on an MI355X"
        toks, logits = _run(model, prompt, steps, forced=want_toks)
        assert model.engine_fallbacks == 0 and (model._engine is not None) == (name != "multi_launch")
        results[name] = (toks, logits, _check(toks, logits, want_toks, want_logits, tdtype, name))
        del model
        torch.cuda.empty_cache()
    # graph replay launches the same kernel on the same inputs: bit-equal logits
    for a, b in zip(results["engine_graph"][1], results["engine_eager"][1]):
        assert torch.equal(a, b)
    print(f"\n[engine] {dtype} ctx {prompt_len}+{steps}: ulps of the row scale from the exact oracle — engine "
          f"{results['engine_graph'][2]:.2f}, multi-launch {results['multi_launch'][2]:.2f}")


def test_engine_free_running_generation_equals_teacher_forced_tokens(tmp_path):
    """Look-ahead + graph replay + engine: a free-running generation feeds each sampled token back on the device."""
    from swiftllm_amd import EngineConfig, LlamaModel
    cfg = synth.make_config(**SMALL)
    sd = synth.make_state_dict(cfg, seed=9, dtype=torch.bfloat16)
    synth.write_model_dir(str(tmp_path), cfg, sd)
    prompt = list(range(40, 75))
    outs = []
    for opts in (dict(tuning=dict(decode_engine=True)), dict(use_hip_graph=False, tuning=dict(decode_engine=True))):
        model = LlamaModel(EngineConfig(model_path=str(tmp_path), **_kw("bfloat16", **opts)))
        model.load_weights()
        model.init_kvcache_and_swap(64)
        assert model._engine is not None
        toks, _ = _run(model, prompt, 40)
        outs.append(toks)
        assert model.engine_fallbacks == 0
        del model
        torch.cuda.empty_cache()
    assert outs[0] == outs[1]


def test_a_poisoned_engine_falls_back_to_the_multi_launch_path(tmp_path):
    """The error protocol: a workspace whose error word is set makes the step return at once with that code; the model then
    re-runs the step on the multi-launch HIP path and stays there."""
    from swiftllm_amd import EngineConfig, LlamaModel
    cfg = synth.make_config(**SMALL)
    sd = synth.make_state_dict(cfg, seed=9, dtype=torch.bfloat16)
    synth.write_model_dir(str(tmp_path), cfg, sd)
    prompt = list(range(100, 140))

    def build(**opts):
        m = LlamaModel(EngineConfig(model_path=str(tmp_path), **_kw("bfloat16", **opts)))
        m.load_weights()
        m.init_kvcache_and_swap(64)
        return m
    ref = build()
    want, _ = _run(ref, prompt, 6)
    del ref
    model = build(tuning=dict(decode_engine=True))
    assert model._engine is not None
    first = model.forward([prompt], [0], [])
    with torch.inference_mode():
        model._engine.ws[1] = 3 | (17 << 8)     # "barrier wait timed out on CU 17"
    toks = [first]
    n = len(prompt)
    for _ in range(6):
        n += 1
        toks.append(model.forward([[toks[-1][0]]], [0], [n]))
    assert model._engine is None and model.engine_fallbacks == 1
    assert toks == want
