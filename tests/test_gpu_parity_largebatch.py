"""End-to-end parity at the LARGE-BATCH operating point the 264 GB pool exists for (VERDICT r04 "missing" 2): Llama-3-8B
width (hidden 4096, 32 q / 8 kv heads of 128, FFN 14336; 2 layers, 8k vocabulary), decode batches of 128 and 256
sequences at contexts ~1.1k, the product's default path (packed weights, gemm_wide / hipBLASLt mix of kernels/linear.py,
slab-fed attention for up to 256 sequences, hipGraph replay with batch buckets), sequences in pool blocks whose offsets
exceed 2^31 elements.

The reference path it must match: swiftllm/worker/layers/transformer_layer.py:54-77,101-128 and kernels/paged_attn.py:152-222
(grid (Bd, H, nsb) at Bd = 256), kernels/linear.py:3-12.

Parties (tests/_parity.py):
  * the PRODUCT: one prompt pass (ragged 1000..1100-token prompts) + 2 teacher-forced decode steps;
  * the CPU ORACLE with exact scores, for the DECODE steps only: a prompt pass of 256 x 1.1k tokens through a CPU model
    is ~2.5e14 flop, so the oracle decodes on a copy of the KV the product's prompt pass stored (block for block) — the
    prompt pass itself is held to the oracle at this width by tests/test_gpu_parity_fullwidth.py; what this test adds is
    the large-batch DECODE composition;
  * the COMPILED REFERENCE (oracle/ref_triton.py, its own process): its own prompt pass + the same two decode steps,
    teacher-forced with the oracle's tokens — with SWIFTLLM_PARITY_FULL_CONTROL=1 only (10 minutes at batch 256; the r05 run
    is committed as profiles/r05k_parity_largebatch_256_bfloat16.json).
Bars (written here): decode logits within 3 ulps of the storage dtype at the row's scale of the exact oracle; every greedy
id that differs from the oracle's sits on a near-tie (the oracle's top-2 gap within twice that row's logit distance);
against the compiled reference: ours no farther from it than 3 ulps + its own distance from the exact oracle, id differences
on its near-ties only. A JSON report goes to gpurun_out/parity_largebatch_<batch>_<dtype>.json."""
import os

import pytest
import torch

from oracle import synth
from tests import _parity as P

pytestmark = pytest.mark.gpu

CFG = dict(num_hidden_layers=2, hidden_size=4096, num_attention_heads=32, num_key_value_heads=8,
           intermediate_size=14336, vocab_size=8192, max_position_embeddings=2048, rope_theta=500000.0)
STEPS = 2


@pytest.mark.parametrize("batch", [128, 256])
def test_large_batch_decode_at_llama3_8b_width(tmp_path, batch):
    import subprocess
    import sys
    from swiftllm_amd import _hip
    dtype, tdtype = "bfloat16", torch.bfloat16
    cfg = synth.make_config(**CFG)
    sd = synth.make_state_dict(cfg, seed=41, dtype=tdtype)
    g = torch.Generator().manual_seed(12 + batch)
    lens = [1000 + int(x) for x in torch.randint(0, 101, (batch,), generator=g)]
    lens[0], lens[-1] = 1100, 1000
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in lens]
    seq_ids = list(range(batch))
    path = str(tmp_path / "model")
    synth.write_model_dir(path, cfg, sd)

    # ---- the product: prompt pass, then STEPS decode steps (teacher-forced below, once the oracle has spoken) -------------
    calls = []
    orig_call = _hip.call

    def spy(name, *a):
        calls.append(name)
        return orig_call(name, *a)
    from swiftllm_amd import EngineConfig, LlamaModel
    model = LlamaModel(EngineConfig(model_path=path, use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0,
                                    max_seqs_in_block_table=batch + 16, max_blocks_per_seq=8192, max_batch_size=batch,
                                    max_tokens_in_batch=batch * (max(lens) + 16), dtype=dtype))
    model.load_weights()
    # a 2-layer pool block is 2^15 elements per cache: block ids >= 2^16 put every offset of the run beyond 2^31 elements
    block_elems = CFG["num_hidden_layers"] * CFG["num_key_value_heads"] * 16 * (CFG["hidden_size"] // CFG["num_attention_heads"])
    high = (1 << 31) // block_elems + 1
    need = sum(-(-(n + STEPS + 1) // 16) for n in lens)
    model.init_kvcache_and_swap(high + need + 8)
    spare, sid = high, batch
    while spare > 0:                    # filler sequences take the lowest block ids
        n = min(spare, 8192)
        model.gpu_block_manager.allocate_blocks_for_seqs([sid], [n * 16])
        spare -= n
        sid += 1
    model.post_layer.logits_tap = []
    tap = model.post_layer.logits_tap
    first = model.forward(prompts, seq_ids, [])
    del tap[:]
    blocks = {s: list(model.gpu_block_manager.host.seq_blocks[s]) for s in seq_ids}
    assert min(b for bl in blocks.values() for b in bl) >= high      # every pool offset of the run > 2^31 elements

    # ---- the exact oracle on a copy of that KV ---------------------------------------------------------------------------
    oracle = P.exact_oracle(cfg, sd, tdtype, batch, max(lens) + STEPS + 2)
    oracle.gpu_block_manager.allocate_blocks_for_seqs(torch.tensor(seq_ids, dtype=torch.int32),
                                                      torch.tensor(lens, dtype=torch.int32))
    bt = oracle.gpu_block_manager.block_table
    for s in seq_ids:
        mine = torch.tensor(blocks[s], dtype=torch.long)
        theirs = bt[s, : len(blocks[s])].long()
        oracle.k_cache[theirs] = model.k_cache[mine.cuda()].cpu()
        oracle.v_cache[theirs] = model.v_cache[mine.cuda()].cpu()
    want_toks, want_logits, feed, cur = [], [], first, list(lens)
    script = [dict(input_ids=prompts, seq_ids=seq_ids, dec_lens=[])]
    for s in range(STEPS):
        cur = [n + 1 for n in cur]
        script.append(dict(input_ids=[[t] for t in feed], seq_ids=seq_ids, dec_lens=list(cur)))
        want_toks.append(oracle.forward([[t] for t in feed], seq_ids, list(cur)))
        want_logits.append(oracle.last_logits.clone())
        feed = want_toks[-1]
    del oracle

    # ---- the product's decode steps, default path (hipGraph replay), every library call recorded --------------------------
    _hip.call = spy
    try:
        ours_toks, ours_logits, feed, cur = [], [], first, list(lens)
        for s in range(STEPS):
            cur = [n + 1 for n in cur]
            ours_toks.append(model.forward([[t] for t in feed], seq_ids, list(cur)))
            ours_logits.append(tap[-1].float().cpu())
            del tap[:]
            feed = want_toks[s]
    finally:
        _hip.call = orig_call
    assert model.graph_captures >= 1 and "swl_paged_attn_decode_qkv_rs" not in calls   # (> 32 sequences: exact norm)
    assert "swl_paged_attn_decode_qkv" in calls, sorted(set(calls))     # slab-fed attention serves up to 256 sequences
    assert any(c.startswith("swl_gemm_packed_wide") for c in calls), sorted(set(calls))
    del model
    torch.cuda.empty_cache()

    def compare(toks, logits, ref_toks, ref_logits):
        rows = []
        for s, (a, b) in enumerate(zip(logits, ref_logits)):
            d = (a - b).abs()
            row_abs = d.amax(dim=1)
            row_ulp = row_abs / P.ulp(b.abs().amax(dim=1), tdtype)
            mism = [i for i, (x, y) in enumerate(zip(toks[s], ref_toks[s])) if x != y]
            top2 = b.topk(2, dim=1).values
            rows.append(dict(step=s, max_abs=float(row_abs.max()), max_ulp_of_row=float(row_ulp.max()), mismatches=len(mism),
                             off_tie=sum(int(float(top2[i, 0] - top2[i, 1]) > 2 * float(row_abs[i])) for i in mism)))
        return rows

    report = dict(batch=batch, dtype=dtype, model=CFG, contexts=[min(lens), max(lens)], decode_steps=STEPS,
                  ours_vs_exact=compare(ours_toks, ours_logits, want_toks, want_logits),
                  kernels=sorted(set(c for c in calls if c.startswith("swl_gemm") or "attn" in c)))
    failures = []
    worst = max(r["max_ulp_of_row"] for r in report["ours_vs_exact"])
    if worst > 3.0:
        failures.append(f"ours is {worst:.2f} ulps of the row scale from the exact oracle (bar 3)")
    if any(r["off_tie"] for r in report["ours_vs_exact"]):
        failures.append(f"greedy ids differ from the exact oracle's off a near-tie: {report['ours_vs_exact']}")

    # The compiled reference needs ~10 MINUTES for this script at batch 256 (600 s measured in r05: its prompt pass of 256 x 1.1k
    # tokens), so its leg runs with SWIFTLLM_PARITY_FULL_CONTROL=1 only; the r05 run is profiles/r05k_parity_largebatch_256_bfloat16.json:
    # reference 8.4 ulps from the exact oracle, ours 2.0, ours-vs-reference 8.5, every differing id on a near-tie.
    if P.STAGED and os.environ.get("SWIFTLLM_PARITY_FULL_CONTROL") == "1":
        torch.save(dict(config=cfg, model_path=path, num_blocks=batch * 72 + 8, max_len=max(lens) + 8, steps=script,
                        dtype=dtype), tmp_path / "job.pt")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        env.pop("TRITON_INTERPRET", None)
        r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "forward", str(tmp_path / "job.pt"),
                            str(tmp_path / "ref.pt")], cwd=P.ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res = torch.load(tmp_path / "ref.pt", weights_only=False)[1:]        # (the prompt pass is not compared here)
        tri_toks, tri_logits = [x["tokens"] for x in res], [x["logits"].float() for x in res]
        report["reference_vs_exact"] = compare(tri_toks, tri_logits, want_toks, want_logits)
        report["ours_vs_reference"] = compare(ours_toks, ours_logits, tri_toks, tri_logits)
        ref_worst = max(r["max_ulp_of_row"] for r in report["reference_vs_exact"])
        vs_ref = max(r["max_ulp_of_row"] for r in report["ours_vs_reference"])
        if vs_ref > 3.0 + ref_worst:
            failures.append(f"ours is {vs_ref:.2f} ulps from the compiled reference, which is {ref_worst:.2f} from exact")
        if any(r["off_tie"] for r in report["ours_vs_reference"]):
            failures.append(f"greedy ids differ from the compiled reference's off a near-tie: {report['ours_vs_reference']}")
    P.write_report(f"parity_largebatch_{batch}_{dtype}.json", report)
    print("\n[large-batch parity]", batch, {k: v for k, v in report.items() if k.endswith("exact") or k.endswith("reference")})
    assert not failures, failures
