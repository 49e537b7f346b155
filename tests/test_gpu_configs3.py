"""Correctness at BASELINE configs[3] geometry: Llama-2-7B heads (32 query = 32 kv heads of 128:
G = 1, the VALU attend_block of csrc/paged_attn.hip, not the matrix-core one), batch 4, ~16.4k-token contexts,
multi-split flash-decoding — the launch geometry the product picks for it with eager launches (batch_plan) AND the one
hipGraph replay buckets it to (LlamaModel._graph_bucket) — against the CPU oracle, against the compiled reference
Triton kernel, and end to end (2 layers at Llama-2-7B width, KV pool filled directly: no 16k-token prompt pass on the
CPU side) with graph replay held bit-equal to eager launches at the replay geometry.

Tolerances: decode attention output is one storage-dtype rounding of an O(1) value (2e-3 fp16 / 1.6e-2 bf16 against the
exact-score oracle, as tests/test_gpu_kernels.py); <= 4e-3 against the reference kernel (it rounds scores to fp16,
paged_attn.py:72-73); whole forward: within 4 storage-dtype ulps of the row scale, greedy ids equal except on per-row
near-ties.
"""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

from oracle import eager_ops as ops
from oracle import synth
from oracle.ref_model import RefLlamaModel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py"))
NS = types.SimpleNamespace
H = KVH = 32
D = 128
LENS = [16416, 16405, 16384, 16385]     # BASELINE configs[3]: 4 x ~16k, around the 16384 boundary and the bench's window


def _geometries():
    """(name, seq_block_size, num_seq_blocks) the product launches this batch with: eager plan and replay bucket."""
    from swiftllm_amd.worker.batch_plan import plan_batch
    from swiftllm_amd.worker.model import LlamaModel
    plan = plan_batch([[0]] * len(LENS), list(range(len(LENS))), LENS, KVH, 256)
    eager = (int(plan.seq_block_size), int(plan.num_seq_blocks))
    bucket = LlamaModel._graph_bucket(None, plan)
    return [("eager_plan",) + eager, ("graph_bucket",) + tuple(int(x) for x in bucket)]


def _case(dtype, L=1, layer=0):
    g = torch.Generator().manual_seed(163)
    seq_ids = [3, 0, 2, 1]
    nblk = sum(-(-n // 16) for n in LENS) + 5
    kc = torch.randn(nblk, L, KVH, 16, D, generator=g).to(dtype)
    vc = torch.randn(nblk, L, KVH, 16, D, generator=g).to(dtype)
    perm = torch.randperm(nblk, generator=g).tolist()
    bt = torch.zeros(4, -(-max(LENS) // 16) + 2, dtype=torch.int32)
    for sid, n in zip(seq_ids, LENS):
        for j in range(-(-n // 16)):
            bt[sid, j] = perm.pop()
    q = torch.randn(len(LENS), H, D, generator=g).to(dtype)
    return q, kc, vc, bt, seq_ids


def _state(seq_ids, sbs, nsb, device):
    return NS(num_decoding_seqs=len(LENS), num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
              softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(LENS, dtype=torch.int32, device=device),
              seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device=device))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["float16", "bfloat16"])
def test_configs3_paged_attention_vs_oracle_and_compiled_reference(tmp_path, dtype):
    from swiftllm_amd.worker import kernels as K
    q, kc, vc, bt, seq_ids = _case(dtype)
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=1), NS(block_size=16)
    want = torch.zeros_like(q)
    ops.paged_attention(q, kc, vc, bt, mc, ec, _state(seq_ids, 2048, -(-max(LENS) // 2048), "cpu"), 0, want)
    tri = None
    if STAGED and dtype == torch.float16:
        # the reference's own kernel. (Its own heuristic gives a 2048-token split for this batch, model.py:305-324; the
        # kernel unrolls seq_block_size / 16 block iterations at compile time and the 128-fold body takes Triton ~100 s to
        # build on a fresh box — 512 tokens compile in a quarter of that and the result is split-invariant up to fp32
        # reassociation, which the four geometries of OUR kernel below demonstrate on the same data.)
        torch.save({"paged": dict(op="paged_attention", H=H, KVH=KVH, D=D, L=1, layer=0, lens=LENS, seq_ids=seq_ids,
                                  seq_block_size=512, q=q, k_cache=kc, v_cache=vc, block_table=bt)}, tmp_path / "in.pt")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        env.pop("TRITON_INTERPRET", None)
        r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "ops", str(tmp_path / "in.pt"),
                            str(tmp_path / "out.pt")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        tri = torch.load(tmp_path / "out.pt", weights_only=False)["paged"]["o"]
    qd, kd, vd, btd = q.cuda(), kc.cuda(), vc.cuda(), bt.cuda()
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    report = dict(lens=LENS, heads=[H, KVH, D], dtype=str(dtype))
    outs = {}
    for name, sbs, nsb in _geometries() + [("reference_split", 2048, -(-max(LENS) // 2048)), ("one_wave_blocks", 64, -(-max(LENS) // 64))]:
        o = torch.zeros_like(qd)
        K.paged_attention(qd, kd, vd, btd, mc, ec, _state(seq_ids, sbs, nsb, "cuda"), 0, o)
        outs[name] = o.cpu()
        err = (outs[name].float() - want.float()).abs().max().item()
        report[name] = dict(seq_block_size=sbs, num_seq_blocks=nsb, vs_oracle_max_abs=err)
        assert err <= tol, (name, sbs, nsb, err)
        if tri is not None:
            err_t = (outs[name].float() - tri.float()).abs().max().item()
            report[name]["vs_reference_triton_max_abs"] = err_t
            assert err_t <= 4e-3, (name, err_t)
    if tri is not None:
        report["reference_triton_vs_oracle_max_abs"] = (tri.float() - want.float()).abs().max().item()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"parity_configs3_paged_attention_{str(dtype).split('.')[-1]}.json"), "w",
              encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    print("\n[configs[3] paged attention]", json.dumps(report))


# float16 (the uneven K-split float16 path end to end, 32 s) runs with SWIFTLLM_PARITY_FULL_CONTROL=1: it held the same bars in
# r03 (profiles/r03_parity_configs3_*), the G = 1 attention kernel itself is tested in both dtypes above, and the suite has
# a time budget (kept behind a switch rather than removed)
@pytest.mark.parametrize("dtype", ["bfloat16"] + (["float16"] if os.environ.get("SWIFTLLM_PARITY_FULL_CONTROL") == "1" else []))
def test_configs3_decode_forward_vs_oracle_and_replay_equals_eager(tmp_path, dtype):
    """2 layers at Llama-2-7B width (hidden 4096, 32/32 heads, FFN 11008 — the uneven K-split projection), batch 4 at
    contexts ~16.4k: the KV pool is filled with the same N(0,1) data on both sides, then 2 decode steps run."""
    from swiftllm_amd import EngineConfig, LlamaModel, LlamaModelConfig
    tdtype = torch.float16 if dtype == "float16" else torch.bfloat16
    cfg = synth.make_config(num_hidden_layers=2, hidden_size=4096, num_attention_heads=32, num_key_value_heads=32,
                            intermediate_size=11008, vocab_size=4096, max_position_embeddings=4096, rope_theta=10000.0,
                            rope_scaling=5.0)
    sd = synth.make_state_dict(cfg, seed=43, dtype=tdtype)
    synth.write_model_dir(str(tmp_path), cfg, sd)
    batch, steps = len(LENS), 2
    seq_ids = list(range(batch))
    blocks_per_seq = -(-(max(LENS) + steps + 1) // 16)
    num_blocks = batch * blocks_per_seq + 4
    kw = dict(use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0, max_seqs_in_block_table=8,
              max_blocks_per_seq=blocks_per_seq + 4, max_batch_size=batch, max_tokens_in_batch=batch * 64, dtype=dtype)
    g = torch.Generator().manual_seed(5)
    pool_shape = (num_blocks, 2, KVH, 16, D)
    k0 = torch.randn(pool_shape, generator=g).to(tdtype)
    v0 = torch.randn(pool_shape, generator=g).to(tdtype)
    first = [n - 1 for n in LENS]      # lengths before the first decoded token
    toks0 = torch.randint(0, cfg["vocab_size"], (batch,), generator=g).tolist()

    ref = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(model_path="", **kw), sd, tdtype)
    ref.init_kvcache_and_swap(num_blocks)
    ref.k_cache.copy_(k0)
    ref.v_cache.copy_(v0)
    want_toks, want_logits, cur, feed = [], [], list(first), toks0
    for s in range(steps):
        cur = [n + 1 for n in cur]
        feed = ref.forward([[t] for t in feed], seq_ids, list(cur))
        want_toks.append(feed)
        want_logits.append(ref.last_logits.clone())
    del ref, sd

    def run(opts, bucketed=False):
        model = LlamaModel(EngineConfig(model_path=str(tmp_path), **kw, **opts))
        model.load_weights()
        model.init_kvcache_and_swap(num_blocks)
        with torch.inference_mode():
            model.k_cache.copy_(k0)
            model.v_cache.copy_(v0)
        model._eager_uses_graph_buckets = bucketed
        model.post_layer.logits_tap = []
        toks, logits, c, f = [], [], list(first), toks0
        for s in range(steps):
            c = [n + 1 for n in c]
            toks.append(model.forward([[t] for t in f], seq_ids, list(c)))
            logits.append(model.post_layer.logits_tap[-1].float().cpu())
            f = want_toks[s]       # teacher-forced with the oracle's tokens
        del model
        torch.cuda.empty_cache()
        return toks, logits

    eps = 2.0 ** -10 if dtype == "float16" else 2.0 ** -7
    results = {}
    for name, opts, bucketed in (("hipgraph", dict(), False), ("eager", dict(use_hip_graph=False), False),
                                 ("eager_bucketed", dict(use_hip_graph=False), True)):
        toks, logits = run(opts, bucketed)
        results[name] = (toks, logits)
        for s in range(steps):
            d = (logits[s] - want_logits[s]).abs()
            scale = want_logits[s].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
            assert (d <= 4 * eps * scale).all(), (name, s, float((d / scale).max()))
            for i, (x, y) in enumerate(zip(toks[s], want_toks[s])):
                if x != y:
                    top2 = want_logits[s][i].topk(2).values
                    assert float(top2[0] - top2[1]) <= 2 * float(d[i].max()), (name, s, i)
    assert results["hipgraph"][0] == results["eager_bucketed"][0]
    for s in range(steps):
        assert torch.equal(results["hipgraph"][1][s], results["eager_bucketed"][1][s]), s
