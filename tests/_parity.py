"""Shared plumbing of the full-depth parity tests (tests/test_gpu_parity_fulldepth.py, tests/test_gpu_parity_decisive.py).

Three parties take part in those tests:
  * the COMPILED REFERENCE (oracle/ref_triton.py, always its own process), optionally with its reference-vs-itself
    control — the same script re-run under other legal execution plans (`variants`, oracle/ref_triton.py: cmd_forward);
  * the PRODUCT (swiftllm_amd.LlamaModel, default path unless options say otherwise), with its sequences placed in the
    high blocks of a 24 GB pool so that every pool offset of the run is beyond 2^31 elements;
  * the CPU ORACLE with exact scores (oracle/ref_model.py), where a test arbitrates between the two.
All comparisons between logits run on the GPU (a 129-step batch-32 run is 1 GB of 16-bit logits per party).
"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py"))

POOL_BLOCKS = 12288          # 24 GB of KV pool on our side (Llama-3-8B KV dims)
HIGH_BLOCK = 4096            # block id from which a pool offset exceeds 2^31 elements (1 MiB = 2^19 elements per block)


def ulp(x: torch.Tensor, dtype) -> torch.Tensor:
    """Spacing of the 16-bit `dtype` at magnitude |x|."""
    mant = 10 if dtype in (torch.float16, "float16") else 7
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14))) - mant)


def decode_script(prompts, gen):
    """prefill + `gen` self-feeding decode steps in oracle/ref_triton.py's job format."""
    batch = len(prompts)
    seq_ids = list(range(batch))
    script, cur = [dict(input_ids=prompts, seq_ids=seq_ids, dec_lens=[])], [len(p) for p in prompts]
    for _ in range(gen):
        cur = [n + 1 for n in cur]
        script.append(dict(input_ids=None, seq_ids=seq_ids, dec_lens=list(cur)))
    return script


def run_reference(tmp_path, cfg, path, dtype, prompts, gen, variants=None, logits_steps=None, timeout=1500, tag="ref"):
    """The compiled reference on (cfg, checkpoint dir): returns (per-step list of dict(tokens, logits-or-None), variants
    summary or None). Logits come back in the storage dtype on the CPU."""
    batch = len(prompts)
    longest = max(len(p) for p in prompts)
    pad = max([int(v.get("pad", 0)) for v in (variants or [])] + [0])      # dummy sequences of the widest `pad` plan
    num_blocks = (batch + pad) * (-(-(longest + gen + 1) // 16) + 1) + 4
    job, out = tmp_path / f"{tag}_job.pt", tmp_path / f"{tag}.pt"
    torch.save(dict(config=cfg, model_path=path, num_blocks=num_blocks, max_len=longest + gen + 16,
                    steps=decode_script(prompts, gen), dtype=dtype, logits="storage", logits_steps=logits_steps,
                    variants=variants or []), job)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("TRITON_INTERPRET", None)
    r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "forward", str(job), str(out)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = torch.load(out, weights_only=False)
    summary = None
    if variants:
        with open(str(out) + ".variants.json", encoding="utf-8") as f:
            summary = json.load(f)
    os.remove(out)
    return res, summary


def our_model(path, dtype, batch, prompt_len, gen, high_blocks=True, **opts):
    """The product on the checkpoint; filler sequences (ids batch .. batch+k) take the lowest block ids so that the test
    sequences land at >= HIGH_BLOCK."""
    from swiftllm_amd import EngineConfig, LlamaModel
    model = LlamaModel(EngineConfig(model_path=path, use_dummy=False, block_size=16, gpu_mem_utilization=0.9,
                                    num_cpu_blocks=0, max_seqs_in_block_table=batch + 8, max_blocks_per_seq=8192,
                                    max_batch_size=batch, max_tokens_in_batch=batch * (prompt_len + 16), dtype=dtype, **opts))
    model.load_weights()
    model.init_kvcache_and_swap(POOL_BLOCKS)
    if high_blocks:
        need = batch * (-(-(prompt_len + gen + 1) // 16))
        spare, sid = POOL_BLOCKS - need - 2, batch
        while spare > 0:
            n = min(spare, 8192)
            model.gpu_block_manager.allocate_blocks_for_seqs([sid], [n * 16])
            spare -= n
            sid += 1
    model.post_layer.logits_tap = []
    return model


def generate(model, prompts, gen, forced=None, logits_steps=None):
    """prefill + `gen` greedy steps (teacher-forced with `forced` tokens when given). Returns (tokens per step, logits per
    step ON THE GPU in the storage dtype — None for steps outside `logits_steps` —, (lowest, highest) block id used)."""
    batch = len(prompts)
    seq_ids = list(range(batch))
    tap = model.post_layer.logits_tap

    def grab(step):
        lg = tap[-1].clone() if (logits_steps is None or step in logits_steps) else None
        del tap[:]
        return lg
    toks, logits = [model.forward(prompts, seq_ids, [])], []
    logits.append(grab(0))
    cur = [len(p) for p in prompts]
    for s in range(gen):
        cur = [n + 1 for n in cur]
        feed = forced[s] if forced is not None else toks[-1]
        toks.append(model.forward([[t] for t in feed], seq_ids, list(cur)))
        logits.append(grab(s + 1))
    blocks = [b for s in seq_ids for b in model.gpu_block_manager.host.seq_blocks[s]]
    model.free_seqs_resources(seq_ids)
    return toks, logits, (min(blocks), max(blocks))


def compare_to_reference(toks, logits, ref_toks, ref_logits, tdtype):
    """Teacher-forced comparison of one party against the reference's logits (both histories identical by construction).
    Returns dict(max_abs_dlogit, max_ulp_of_row, token_mismatches, mismatches_not_on_a_near_tie, mismatches[:64],
    per_step). A mismatch is "on a near-tie" when the reference's own top-2 gap in that row is within twice that row's
    logit distance."""
    worst_abs = worst_ulp = 0.0
    mism, bad, per_step = [], 0, []
    for s, (a, b) in enumerate(zip(logits, ref_logits)):
        if a is None or b is None:
            continue
        a, b = a.cuda().float(), b.cuda().float()
        row_abs = (a - b).abs().amax(dim=1)
        row_ulp = row_abs / ulp(b.abs().amax(dim=1), tdtype)
        worst_abs, worst_ulp = max(worst_abs, float(row_abs.max())), max(worst_ulp, float(row_ulp.max()))
        per_step.append(dict(step=s, max_abs=float(row_abs.max()), max_ulp_of_row=float(row_ulp.max())))
        top2 = b.topk(2, dim=1).values
        for i, (x, y) in enumerate(zip(toks[s], ref_toks[s])):
            if x != y:
                gap = float(top2[i, 0] - top2[i, 1])
                mism.append(dict(step=s, seq=i, ref_top2_gap=gap, row_max_abs=float(row_abs[i])))
                bad += int(gap > 2 * float(row_abs[i]))
    return dict(max_abs_dlogit=worst_abs, max_ulp_of_row=worst_ulp, token_mismatches=len(mism),
                tokens_compared=sum(len(t) for t, a, b in zip(toks, logits, ref_logits) if a is not None and b is not None),
                mismatches_not_on_a_near_tie=bad, mismatches=mism[:64], per_step=per_step)


def first_divergences(toks, ref_toks):
    """Per sequence: the first step at which a free-running stream leaves the reference's (None = identical to the end)."""
    batch = len(toks[0])
    return [next((s for s in range(len(ref_toks)) if toks[s][i] != ref_toks[s][i]), None) for i in range(batch)]


def exact_oracle(cfg, sd, tdtype, batch, max_len):
    """The CPU oracle with EXACT (fp32) decode scores at full depth, set up to be affordable there: the projection weights
    are converted to fp32 ONCE (eager_ops.linear computes F.linear(a.float(), w.float()) — for an fp32 `w` the conversion
    is the identity, so every product and every rounding is what the 16-bit weights give; per-call conversion of 8 G
    parameters was most of a forward's time), decode attention as one dense softmax per sequence (paged_attention_dense:
    the same value up to fp32 reassociation, no Python loop per KV block). The embedding stays in the storage dtype."""
    from oracle.ref_model import RefLlamaModel
    from swiftllm_amd import EngineConfig, LlamaModelConfig
    blocks_per_seq = -(-max_len // 16) + 1
    oracle = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(
        model_path="", use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0,
        max_seqs_in_block_table=max(2, batch), max_blocks_per_seq=blocks_per_seq + 2, max_batch_size=batch,
        max_tokens_in_batch=batch * max_len), sd, tdtype, score_dtype="fp32", dense_decode_attention=True)
    for layer in oracle.layers:
        for name in ("q_proj", "k_proj", "v_proj", "o_proj", "up_gate_proj", "down_proj"):
            setattr(layer, name, getattr(layer, name).float())
    oracle.lm_head = oracle.lm_head.float()
    oracle.init_kvcache_and_swap(batch * blocks_per_seq + 2)
    return oracle


def write_report(name: str, report: dict) -> str:
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, name)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    return path
