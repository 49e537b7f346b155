"""The decisive-logit checkpoint (oracle/synth.py: make_decisive_state_dict) on the CPU oracle.

Before tests/test_gpu_parity_decisive.py may assert bit-exact greedy ids at Llama-3-8B geometry on it, the construction
itself is held to its specification here, at a geometry the CPU oracle walks in seconds: the greedy stream of
oracle/ref_model.py — exact scores and the reference kernel's storage-dtype score rounding (paged_attn.py:72-73), float16 and
bfloat16 — equals the closed form `token after p = perm[token(p - offset)]`, with a top-2 gap hundreds of ulps wide, through
prefill attention (first token), decode attention over prompt K/V (steps < offset) and over K/V the decode steps stored
themselves (steps >= offset).
"""
import pytest
import torch

from oracle import synth
from oracle.ref_model import RefLlamaModel

CFG = dict(num_hidden_layers=3, hidden_size=1024, num_attention_heads=16, num_key_value_heads=4, intermediate_size=2048,
           vocab_size=2048, max_position_embeddings=2048, rope_theta=500000.0)
OFFSET, STEPS = 19, 26


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("score", ["fp32", "ref"])
def test_decisive_checkpoint_walks_its_closed_form_on_the_oracle(dtype, score):
    from swiftllm_amd import EngineConfig, LlamaModelConfig
    cfg = synth.make_config(**CFG)
    sd, perm, info = synth.make_decisive_state_dict(cfg, seed=5, dtype=dtype, offset=OFFSET, max_context=300)
    assert info["rotary_pairs"] <= 16 and info["peak_score_nats"] < 64
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(0, cfg["vocab_size"], (60 + 7 * i,), generator=g).tolist() for i in range(4)]
    want = synth.decisive_expected_tokens(prompts, perm, OFFSET, STEPS)
    seq_ids = list(range(len(prompts)))
    model = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(
        model_path="", use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0, max_seqs_in_block_table=8,
        max_blocks_per_seq=16, max_batch_size=4, max_tokens_in_batch=1024), sd, dtype, score_dtype=score)
    model.init_kvcache_and_swap(40)
    ulp = 2.0 ** (-1 - (10 if dtype == torch.float16 else 7))       # spacing of the storage dtype in [0.5, 1)
    toks, cur = [model.forward(prompts, seq_ids, [])], [len(p) for p in prompts]
    gaps = [model.last_logits.topk(2).values]
    for _ in range(STEPS):
        cur = [n + 1 for n in cur]
        toks.append(model.forward([[t] for t in toks[-1]], seq_ids, list(cur)))
        gaps.append(model.last_logits.topk(2).values)
    top2 = torch.stack(gaps)
    assert toks == want
    assert float((top2[..., 0] - top2[..., 1]).min()) > 50 * ulp      # measured: ~0.2 = 400 fp16 / 50 bf16 ulps at this width
    assert float(top2[..., 0].max()) < 2.0
    # generated tokens feed later decisions (steps >= offset read positions past the prompt): the stream is not a lookup
    assert STEPS > OFFSET
