"""Parity at the REAL geometry of BASELINE configs[1] / configs[2] (SURVEY.md §8c Tier 2):

32-layer Llama-3-8B (random-init, full 128 256-token vocabulary), 1024-token prompts, **128 free-running greedy steps**,
the compiled reference (its own Triton kernels on this MI355X, oracle/ref_triton.py) against the product's default path
— float16 (the reference's only precision) and bfloat16 (the headline dtype, against the mechanically patched
float16 -> bfloat16 twin of the reference, oracle/make_ref.py).

For configs[1] (batch 1) the CPU oracle with EXACT scores (oracle/ref_model.py, fp32 accumulation everywhere, the
reference's rounding points) additionally runs the prompt + 4 teacher-forced decode steps at full depth and arbitrates:
both implementations' distance to it is reported, and ours must not be the larger one. (Two 16-bit implementations of a
32-layer random-init network differ by far more than at 2 layers — every layer amplifies the 1-ulp differences of the
one before — so "how far apart" only means something next to "how far from exact".)

Three runs per case:
  * reference, free-running (feeds itself);
  * ours, free-running: the token streams must be identical up to each sequence's first divergence, and a divergence is
    only acceptable where the reference's own top-2 gap in that row is within the PER-ROW logit distance;
  * ours, teacher-forced with the reference's tokens: logit distance at every one of the 129 steps.

On OUR side the KV pool is 12 288 blocks (24 GB) and filler sequences hold the low block ids, so the test sequences
live in blocks >= 4096: every pool offset of prefill store, decode store and paged attention is beyond 2^31 elements
(VERDICT r02 item 2 — a 32-bit wrap anywhere would change the tokens). Block placement is result-invariant, the
reference side uses its own (low) blocks.

The report (first divergence step per sequence, the reference's top-2 gap there, logit distances) goes to
gpurun_out/parity_fulldepth_<case>_<dtype>.json; the copies under profiles/ are the ones the docs cite.
"""
import json
import os
import shutil
import subprocess
import sys

import pytest
import torch

from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py"))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1800),
              pytest.mark.skipif(not STAGED, reason="oracle/_ref not staged (python -m oracle.make_ref)")]

GEN = 128
PROMPT = 1024
POOL_BLOCKS = 12288          # 24 GB of KV pool on our side
HIGH_BLOCK = 4096            # block id from which a pool offset exceeds 2^31 elements (1 MiB = 2^19 elements per block)
CASES = {"configs1_batch1": 1, "configs2_batch32": 32}


def _ulp(x: torch.Tensor, dtype) -> torch.Tensor:
    mant = 10 if dtype == torch.float16 else 7
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14))) - mant)


@pytest.fixture(scope="module")
def checkpoints(tmp_path_factory):
    """ONE 32-layer Llama-3-8B checkpoint (float16 values, 16 GB), written once and removed at module teardown. The bfloat16
    case loads the same file: both sides round the float16 weights to bfloat16 on load (the reference: weight.py:50
    `.to(item.dtype)`; ours: the per-tensor loader path), to the same bits."""
    made = {}

    def get(dtype):
        if "ckpt" not in made:
            cfg = synth.make_config(**synth.LLAMA3_8B)
            path = str(tmp_path_factory.mktemp("llama3_8b"))
            sd = synth.make_state_dict_on_gpu(cfg, seed=2024, dtype=torch.float16)
            synth.write_model_dir(path, cfg, sd)
            del sd
            made["ckpt"] = (cfg, path)
        return made["ckpt"]
    yield get
    for _, path in made.values():
        shutil.rmtree(path, ignore_errors=True)


def _run_reference(tmp_path, cfg, path, dtype, prompts, batch):
    seq_ids = list(range(batch))
    script, cur = [dict(input_ids=prompts, seq_ids=seq_ids, dec_lens=[])], [len(p) for p in prompts]
    for _ in range(GEN):
        cur = [n + 1 for n in cur]
        script.append(dict(input_ids=None, seq_ids=seq_ids, dec_lens=list(cur)))
    num_blocks = batch * (-(-(PROMPT + GEN + 1) // 16) + 1) + 4
    torch.save(dict(config=cfg, model_path=path, num_blocks=num_blocks, max_len=PROMPT + GEN + 16, steps=script,
                    dtype=dtype, logits="storage"), tmp_path / "job.pt")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("TRITON_INTERPRET", None)
    r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "forward", str(tmp_path / "job.pt"),
                        str(tmp_path / "ref.pt")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(tmp_path / "ref.pt", weights_only=False)


def _our_model(path, dtype, batch):
    from swiftllm_amd import EngineConfig, LlamaModel
    model = LlamaModel(EngineConfig(model_path=path, use_dummy=False, block_size=16, gpu_mem_utilization=0.9,
                                    num_cpu_blocks=0, max_seqs_in_block_table=batch + 8, max_blocks_per_seq=8192,
                                    max_batch_size=batch, max_tokens_in_batch=batch * (PROMPT + 16), dtype=dtype))
    model.load_weights()
    model.init_kvcache_and_swap(POOL_BLOCKS)
    # filler sequences (ids batch .. batch+k) take the lowest block ids: the test sequences land at >= HIGH_BLOCK
    need = batch * (-(-(PROMPT + GEN + 1) // 16))
    spare, sid = POOL_BLOCKS - need - 2, batch
    while spare > 0:
        n = min(spare, 8192)
        model.gpu_block_manager.allocate_blocks_for_seqs([sid], [n * 16])
        spare -= n
        sid += 1
    model.post_layer.logits_tap = []
    return model


def _generate(model, prompts, batch, forced=None, keep_logits=True):
    seq_ids = list(range(batch))
    tap = model.post_layer.logits_tap
    toks, logits = [model.forward(prompts, seq_ids, [])], []
    if keep_logits:
        logits.append(tap[-1].cpu())
    cur = [len(p) for p in prompts]
    for s in range(GEN):
        cur = [n + 1 for n in cur]
        feed = forced[s] if forced is not None else toks[-1]
        toks.append(model.forward([[t] for t in feed], seq_ids, list(cur)))
        if keep_logits:
            logits.append(tap[-1].cpu())
        del tap[:]
    blocks = [b for s in seq_ids for b in model.gpu_block_manager.host.seq_blocks[s]]
    model.free_seqs_resources(seq_ids)
    return toks, logits, (min(blocks), max(blocks))


@pytest.mark.parametrize("case,dtype", [("configs1_batch1", "float16"), ("configs2_batch32", "float16"),
                                        ("configs2_batch32", "bfloat16")])
def test_llama3_8b_full_depth_128_free_running_steps_vs_compiled_reference(tmp_path, checkpoints, case, dtype):
    batch = CASES[case]
    tdtype = torch.float16 if dtype == "float16" else torch.bfloat16
    cfg, path = checkpoints(dtype)
    g = torch.Generator().manual_seed(77)
    prompts = [torch.randint(0, cfg["vocab_size"], (PROMPT,), generator=g).tolist() for _ in range(batch)]

    ref = _run_reference(tmp_path, cfg, path, dtype, prompts, batch)
    ref_toks = [r["tokens"] for r in ref]
    ref_logits = [r["logits"] for r in ref]

    model = _our_model(path, dtype, batch)
    free_toks, free_logits, (blk_lo, blk_hi) = _generate(model, prompts, batch)
    forced_toks, forced_logits, _ = _generate(model, prompts, batch, forced=ref_toks)
    del model
    torch.cuda.empty_cache()
    assert blk_lo >= HIGH_BLOCK, (blk_lo, blk_hi)       # every offset of the run was beyond 2^31 elements

    # ---- teacher-forced: logit distance at all 129 steps, per-row near-tie rule for token differences -------------
    worst_abs = worst_ulp = 0.0
    forced_mism, bad = [], []
    per_step = []
    for s in range(GEN + 1):
        a, b = forced_logits[s].float(), ref_logits[s].float()
        d = (a - b).abs()
        row_abs = d.amax(dim=1)
        row_ulp = row_abs / _ulp(b.abs().amax(dim=1), tdtype)
        worst_abs, worst_ulp = max(worst_abs, float(row_abs.max())), max(worst_ulp, float(row_ulp.max()))
        per_step.append(dict(step=s, max_abs=float(row_abs.max()), max_ulp_of_row=float(row_ulp.max())))
        for i, (x, y) in enumerate(zip(forced_toks[s], ref_toks[s])):
            if x != y:
                top2 = b[i].topk(2).values
                gap = float(top2[0] - top2[1])
                forced_mism.append(dict(step=s, seq=i, ref_top2_gap=gap, row_max_abs=float(row_abs[i])))
                if gap > 2 * float(row_abs[i]):
                    bad.append(("teacher-forced", s, i, gap, float(row_abs[i])))
    # ---- free-running: identical streams up to each sequence's first divergence ----------------------------------
    first_div = []
    for i in range(batch):
        step = next((s for s in range(GEN + 1) if free_toks[s][i] != ref_toks[s][i]), None)
        if step is None:
            first_div.append(dict(seq=i, step=None))
            continue
        # both sides saw the same history up to `step`: their logits there are comparable
        b = ref_logits[step][i].float()
        dist = float((free_logits[step][i].float() - b).abs().max())
        top2 = b.topk(2).values
        gap = float(top2[0] - top2[1])
        first_div.append(dict(seq=i, step=step, ref_top2_gap=gap, row_max_abs=dist))
        if gap > 2 * dist:
            bad.append(("free-running", step, i, gap, dist))
    diverged = [d for d in first_div if d["step"] is not None]
    report = dict(case=case, dtype=dtype, model="Llama-3-8B dims, 32 layers, vocab 128256, random init", batch=batch,
                  prompt_len=PROMPT, free_running_steps=GEN,
                  reference="compiled reference Triton path" + (" (float16 -> bfloat16 patched)" if dtype == "bfloat16" else ""),
                  our_kv_pool_blocks=POOL_BLOCKS, our_block_ids=[blk_lo, blk_hi],
                  our_min_pool_element_offset=blk_lo * 32 * 8 * 16 * 128,
                  teacher_forced=dict(max_abs_dlogit=worst_abs, max_ulp_of_row=worst_ulp,
                                      token_mismatches=len(forced_mism), tokens_compared=(GEN + 1) * batch,
                                      mismatches=forced_mism[:64]),
                  free_running=dict(sequences=batch, identical_to_the_end=batch - len(diverged),
                                    first_divergence=first_div,
                                    earliest_divergence_step=min((d["step"] for d in diverged), default=None)),
                  per_step=per_step, violations=bad)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_fulldepth_{case}_{dtype}.json"), "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    print("\n[full-depth Tier-2]", case, dtype, json.dumps({k: report[k] for k in ("teacher_forced", "our_block_ids")})[:600],
          "diverged:", len(diverged), "earliest:", report["free_running"]["earliest_divergence_step"])
    # ---- arbitration by the exact-score CPU oracle at full depth (batch 1 only: a 32-layer CPU forward per step) ----
    if batch == 1 and dtype == "float16":
        from safetensors.torch import load_file
        from oracle.ref_model import RefLlamaModel
        from swiftllm_amd import EngineConfig, LlamaModelConfig
        n_dec = 4
        sd = load_file(os.path.join(path, "model.safetensors"))
        oracle = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(
            model_path="", use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0,
            max_seqs_in_block_table=2, max_blocks_per_seq=80, max_batch_size=1, max_tokens_in_batch=PROMPT + 16), sd,
            tdtype, score_dtype="fp32")
        oracle.init_kvcache_and_swap(80)
        exact = []
        oracle.forward(prompts, [0], [])
        exact.append(oracle.last_logits.clone())
        for s in range(n_dec):
            oracle.forward([[ref_toks[s][0]]], [0], [PROMPT + 1 + s])     # teacher-forced like the forced run
            exact.append(oracle.last_logits.clone())
        del oracle, sd
        ours_d = max(float((forced_logits[s].float() - exact[s]).abs().max()) for s in range(n_dec + 1))
        ref_d = max(float((ref_logits[s].float() - exact[s]).abs().max()) for s in range(n_dec + 1))
        scale = float(_ulp(torch.stack(exact).abs().amax(dim=2).max(), tdtype))
        report["exact_oracle_arbitration"] = dict(
            steps=n_dec + 1, ours_vs_exact_max_abs=ours_d, reference_vs_exact_max_abs=ref_d,
            ours_vs_exact_ulp_of_row=ours_d / scale, reference_vs_exact_ulp_of_row=ref_d / scale,
            ours_token_mismatches_vs_exact=sum(int(forced_logits[s][0].float().argmax()) != int(exact[s][0].argmax())
                                               for s in range(n_dec + 1)),
            reference_token_mismatches_vs_exact=sum(int(ref_logits[s][0].float().argmax()) != int(exact[s][0].argmax())
                                                    for s in range(n_dec + 1)))
        with open(os.path.join(out_dir, f"parity_fulldepth_{case}_{dtype}.json"), "w", encoding="utf-8") as f:
            json.dump(report, f, indent=1)
        print("[full-depth arbitration by the exact oracle]", json.dumps(report["exact_oracle_arbitration"]))
        assert ours_d <= 1.25 * ref_d, report["exact_oracle_arbitration"]
    # Every token difference sits on a near-tie of the reference (per row: gap <= 2 x that row's logit distance) ...
    assert not bad, bad
    # ... and the logits stay within the band measured between the two implementations at this depth (r03: 33.5 / 37.9
    # fp16 ulps of the row scale at batch 1 / 32, 40.3 bf16 ulps against the bf16-patched reference; at 2 layers the same
    # pair is 7 ulps apart): the reference rounds its decode scores to the storage dtype (paged_attn.py:72-73), its
    # q/k/v are three hipBLASLt calls where ours is one fused MFMA kernel on packed weights, and 32 random-init layers
    # amplify every 1-ulp difference. A regression guard, not a precision claim — that is the arbitration above.
    assert worst_ulp <= 64.0, (worst_ulp, worst_abs)
