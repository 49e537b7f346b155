"""Tier-2 parity: the REFERENCE's own Triton kernels, JIT-compiled by Triton's
gfx950 backend on this MI355X, against our HIP kernels on the same inputs — per operator at Llama-3-8B geometry
and for the whole LlamaModel.forward (fp16, the reference's only precision).

The reference files are staged by `python -m oracle.make_ref` (build container) into oracle/_ref/ — git-ignored,
shipped to the GPU box by gpurun; the tests skip when they are absent. The reference always runs in its OWN
process (`python -m oracle.ref_triton ...`): its package is called `swiftllm` like this repo's import alias.

Tolerances (fp16): copies bit-exact; rmsnorm / silu <= 1-2 ulp (fp32 reduction order, exp implementation); rotary
within one rounding of the products (compiled Triton contracts fp16 mul+add into fma); decode attention <= 4e-3 (the
reference rounds scores to fp16, paged_attn.py:72-73 — ours is closer to the exact value); prefill attention
<= 2e-3; whole forward: greedy ids identical except at near-ties below the measured logit distance, logits within
8 ulp of the row scale at Llama-3-8B width — the reference's own fp16-score noise there is ~5 ulp — and the north star's
1e-3 where logits are O(0.1): the tiny model.
"""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

from oracle import synth
from conftest import ulp_diff_fp16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py"))
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not STAGED, reason="oracle/_ref not staged (python -m oracle.make_ref)")]
NS = types.SimpleNamespace


def _ref(cmd, inp, out):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("TRITON_INTERPRET", None)
    r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", cmd, str(inp), str(out)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out, weights_only=False)


def K():
    from swiftllm_amd.worker import kernels
    return kernels


def test_reference_triton_kernels_vs_hip_per_operator(tmp_path):
    g = torch.Generator().manual_seed(41)
    h, inter, H, KVH, D = 4096, 14336, 32, 8, 128
    f16 = torch.float16
    cases = {}
    x = torch.randn(33, h, generator=g).to(f16)
    r = torch.randn(33, h, generator=g).to(f16)
    w = (1 + 0.1 * torch.randn(h, generator=g)).to(f16)
    cases["rmsnorm"] = dict(op="rmsnorm", x=x, w=w, eps=1e-5)
    cases["fused_add_rmsnorm"] = dict(op="fused_add_rmsnorm", x=x, r=r, w=w, eps=1e-5)
    cases["silu"] = dict(op="silu_and_mul", x=(2 * torch.randn(19, 2 * inter, generator=g)).to(f16))
    ang = torch.rand(37, D // 2, generator=g) * 6.28
    cases["rotary"] = dict(op="rotary", q=torch.randn(37, H, D, generator=g).to(f16),
                           k=torch.randn(37, KVH, D, generator=g).to(f16), cos=torch.cos(ang).to(f16),
                           sin=torch.sin(ang).to(f16))
    # decode attention at configs[2]-like shapes: ragged contexts around 1k, sbs 256 (the reference's choice, §8 a2)
    lens = [1100, 1024, 17, 333]
    L, layer, bs = 2, 1, 16
    seq_ids = [1, 2, 3, 4]
    nblk = sum(-(-n // bs) for n in lens) + 3
    kc = torch.randn(nblk, L, KVH, bs, D, generator=g).to(f16)
    vc = torch.randn(nblk, L, KVH, bs, D, generator=g).to(f16)
    perm = torch.randperm(nblk, generator=g).tolist()
    bt = torch.zeros(6, 80, dtype=torch.int32)
    for sid, n in zip(seq_ids, lens):
        for j in range(-(-n // bs)):
            bt[sid, j] = perm.pop()
    q = torch.randn(len(lens), H, D, generator=g).to(f16)
    cases["paged"] = dict(op="paged_attention", H=H, KVH=KVH, D=D, L=L, layer=layer, lens=lens, seq_ids=seq_ids,
                          seq_block_size=256, q=q, k_cache=kc, v_cache=vc, block_table=bt)
    plens = [1024, 130, 5]
    P = sum(plens)
    pq = torch.randn(P, H, D, generator=g).to(f16)
    pk = torch.randn(P, KVH, D, generator=g).to(f16)
    pv = torch.randn(P, KVH, D, generator=g).to(f16)
    cases["prefill"] = dict(op="prefill_attention", H=H, KVH=KVH, D=D, lens=plens, q=pq, k=pk, v=pv)
    # KV store: 2 prefill sequences + 2 decoding ones into a pool with live content
    s_plens, s_dlens = [21, 16], [18, 49]
    sbt = torch.zeros(8, 16, dtype=torch.int32)
    blocks = iter(torch.randperm(14, generator=g).tolist())
    s_seq_ids = torch.tensor([3, 0, 5, 2], dtype=torch.int32)
    for sid, n in zip(s_seq_ids.tolist(), s_plens + s_dlens):
        for j in range(-(-n // bs)):
            sbt[sid, j] = next(blocks)
    T = sum(s_plens) + 2
    cases["store"] = dict(op="store_kvcache", L=3, KVH=KVH, D=D, layer=2, plens=s_plens, dlens=s_dlens,
                          seq_ids=s_seq_ids, block_table=sbt, k=torch.randn(T, KVH, D, generator=g).to(f16),
                          v=torch.randn(T, KVH, D, generator=g).to(f16),
                          k_cache=torch.randn(14, 3, KVH, bs, D, generator=g).to(f16),
                          v_cache=torch.randn(14, 3, KVH, bs, D, generator=g).to(f16))
    torch.save(cases, tmp_path / "in.pt")
    ref = _ref("ops", tmp_path / "in.pt", tmp_path / "out.pt")

    k = K()
    report = {}
    # rmsnorm
    gx = x.cuda()
    k.rmsnorm_inplace(gx, w.cuda(), 1e-5)
    report["rmsnorm_ulp"] = ulp_diff_fp16(gx.cpu(), ref["rmsnorm"]["x"])
    gx, gr = x.cuda(), r.cuda()
    k.fused_add_rmsnorm_inplace(gx, gr, w.cuda(), 1e-5)
    assert torch.equal(gr.cpu(), ref["fused_add_rmsnorm"]["r"])
    report["fused_add_rmsnorm_ulp"] = ulp_diff_fp16(gx.cpu(), ref["fused_add_rmsnorm"]["x"])
    # silu
    gs = cases["silu"]["x"].cuda()
    k.silu_and_mul_inplace(gs)
    report["silu_ulp"] = ulp_diff_fp16(gs.cpu()[:, :inter], ref["silu"]["x"][:, :inter])
    # rotary
    c = cases["rotary"]
    gq, gk = c["q"].cuda(), c["k"].cuda()
    k.rotary_embedding_inplace(gq, gk, NS(position_cos=c["cos"].cuda(), position_sin=c["sin"].cuda()))
    # compiled Triton contracts the fp16 multiply-add of the rotation into an fma (one rounding fewer than the
    # interpreter run our kernel reproduces bit for bit): the results differ by up to one rounding of the PRODUCTS,
    # which under cancellation is many ulps of the small difference — so the bound is on |delta| relative to the
    # operands, eps_fp16 * (|x0| + |x1|), not in ulps of the result
    def rot_excess(got, want, x):
        half = x.shape[-1] // 2
        mag = x[..., :half].float().abs() + x[..., half:].float().abs()
        bound = 2.0 ** -10 * torch.cat([mag, mag], dim=-1)
        return ((got.float() - want.float()).abs() - bound).max().item()
    report["rotary_q_excess"] = rot_excess(gq.cpu(), ref["rotary"]["q"], c["q"])
    report["rotary_k_excess"] = rot_excess(gk.cpu(), ref["rotary"]["k"], c["k"])
    report["rotary_q_max_abs"] = (gq.cpu().float() - ref["rotary"]["q"].float()).abs().max().item()
    # paged attention
    o = torch.zeros_like(q).cuda()
    st = NS(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=256, num_seq_blocks=-(-max(lens) // 256),
            softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device="cuda"),
            seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device="cuda"))
    k.paged_attention(q.cuda(), kc.cuda(), vc.cuda(), bt.cuda(),
                      NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16), st, layer, o)
    report["paged_attention_max_abs"] = (o.cpu().float() - ref["paged"]["o"].float()).abs().max().item()
    # prefill attention
    po = torch.zeros_like(pq).cuda()
    pl = torch.tensor(plens, dtype=torch.int32, device="cuda")
    starts = torch.cumsum(pl, 0, dtype=torch.int32) - pl
    pst = NS(num_prefill_seqs=len(plens), max_prefill_len=max(plens), softmax_scale=D ** -0.5,
             prefill_seq_start_locs=starts, prefill_seq_lens=pl, num_prefill_tokens=P,
             prefill_seq_start_locs_with_end=torch.cat([starts, torch.tensor([P], dtype=torch.int32, device="cuda")]))
    k.prefill_attention(pq.cuda(), pk.cuda(), pv.cuda(), po, NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D), None, pst)
    report["prefill_attention_max_abs"] = (po.cpu().float() - ref["prefill"]["o"].float()).abs().max().item()
    # KV store
    c = cases["store"]
    gkc, gvc = c["k_cache"].cuda(), c["v_cache"].cuda()
    spl = torch.tensor(s_plens, dtype=torch.int32, device="cuda")
    sst = NS(seq_ids=s_seq_ids.cuda(), num_prefill_seqs=2, num_prefill_tokens=sum(s_plens), max_prefill_len=max(s_plens),
             prefill_seq_lens=spl, prefill_seq_start_locs=torch.cumsum(spl, 0, dtype=torch.int32) - spl,
             num_decoding_seqs=2, decoding_seq_lens=torch.tensor(s_dlens, dtype=torch.int32, device="cuda"))
    k.store_kvcache(c["k"].cuda(), c["v"].cuda(), gkc, gvc, sbt.cuda(), NS(num_layers=3, num_kv_heads=KVH, head_dim=D),
                    NS(block_size=16, max_blocks_per_seq=16), sst, 2)
    assert torch.equal(gkc.cpu(), ref["store"]["k_cache"]) and torch.equal(gvc.cpu(), ref["store"]["v_cache"])
    print("\n[tier-2 per-op, compiled reference Triton vs HIP]", json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "tier2_per_op.json"), "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    assert report["rmsnorm_ulp"] <= 1 and report["fused_add_rmsnorm_ulp"] <= 1
    assert report["silu_ulp"] <= 2
    assert report["rotary_q_excess"] <= 0 and report["rotary_k_excess"] <= 0
    assert report["paged_attention_max_abs"] <= 4e-3
    assert report["prefill_attention_max_abs"] <= 2e-3


def _ulp16(x):
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14))) - 10)


@pytest.mark.parametrize("width", ["tiny", "llama3_8b_width"])
def test_reference_forward_vs_hip_forward(tmp_path, width):
    """The reference's whole forward (compiled Triton + F.linear) vs ours on the same checkpoint and prompts:
    prefill, decode steps (each side fed the REFERENCE's tokens), fp16."""
    from swiftllm_amd import EngineConfig, LlamaModel
    if width == "tiny":
        cfg = synth.make_config()
        lens, steps, max_len = [5, 9, 17, 120], 8, 160
    else:
        cfg = synth.make_config(num_hidden_layers=2, hidden_size=4096, num_attention_heads=32, num_key_value_heads=8,
                                intermediate_size=14336, vocab_size=8192, max_position_embeddings=2048,
                                rope_theta=500000.0)
        lens, steps, max_len = [1024, 1, 15, 16, 17, 100, 257, 640, 33, 1000, 511, 513, 64, 900, 31, 300], 5, 1040
    batch = len(lens)
    sd = synth.make_state_dict(cfg, seed=33, dtype=torch.float16)
    synth.write_model_dir(str(tmp_path / "model"), cfg, sd)
    del sd
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in lens]
    seq_ids = list(range(batch))
    num_blocks = sum(-(-(n + steps + 1) // 16) for n in lens) + 4
    script, cur = [dict(input_ids=prompts, seq_ids=seq_ids, dec_lens=[])], list(lens)
    for _ in range(steps):
        cur = [n + 1 for n in cur]
        script.append(dict(input_ids=None, seq_ids=seq_ids, dec_lens=list(cur)))
    torch.save(dict(config=cfg, model_path=str(tmp_path / "model"), num_blocks=num_blocks, max_len=max_len,
                    steps=script), tmp_path / "job.pt")
    ref = _ref("forward", tmp_path / "job.pt", tmp_path / "ref.pt")

    model = LlamaModel(EngineConfig(model_path=str(tmp_path / "model"), use_dummy=False, block_size=16,
                                    gpu_mem_utilization=0.9, num_cpu_blocks=0, max_seqs_in_block_table=max(8, batch),
                                    max_blocks_per_seq=max_len // 16 + 8, max_batch_size=batch,
                                    max_tokens_in_batch=batch * max_len))
    model.load_weights()
    model.init_kvcache_and_swap(num_blocks)
    model.post_layer.logits_tap = []
    worst_abs = worst_ulp = 0.0
    mism = []
    for s, step in enumerate(script):
        ids = step["input_ids"] if s == 0 else [[t] for t in ref[s - 1]["tokens"]]
        toks = model.forward(ids, step["seq_ids"], step["dec_lens"])
        a, b = model.post_layer.logits_tap[-1].float().cpu(), ref[s]["logits"]
        d = (a - b).abs()
        worst_abs = max(worst_abs, d.max().item())
        worst_ulp = max(worst_ulp, (d / _ulp16(b.abs().amax(dim=1, keepdim=True))).max().item())
        for i, (x, y) in enumerate(zip(toks, ref[s]["tokens"])):
            if x != y:
                top2 = b[i].topk(2).values
                mism.append((s, i, float(top2[0] - top2[1])))
    report = dict(width=width, batch=batch, steps=steps + 1, max_abs_dlogit=worst_abs, max_ulp_of_row=worst_ulp,
                  token_mismatches=len(mism), tokens_compared=(steps + 1) * batch, mismatch_top2_gaps=[m[2] for m in mism])
    print("\n[tier-2 forward, compiled reference vs HIP]", json.dumps(report))
    with open(os.path.join(ROOT, "gpurun_out", f"tier2_forward_{width}.json"), "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    if width == "tiny":
        assert worst_abs <= 1e-3, worst_abs          # the north star's bar, where logits are O(0.1)
        assert not mism, mism
    else:
        # At this width the REFERENCE is the noisy side: its fp16 score path (paged_attn.py:72-73) puts it 4.75-5.5 ulp
        # of the row scale from the exact-score oracle (measured by tests/test_gpu_parity_fullwidth.py on this same
        # model geometry), ours sits <= 2 ulp from it. Distance between the two <= the sum.
        assert worst_ulp <= 8.0, (worst_ulp, worst_abs)
        assert all(gap <= 2 * worst_abs for _, _, gap in mism), mism
