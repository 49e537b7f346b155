"""oracle/ref_triton.py — run the STAGED REFERENCE (its own Triton kernels, compiled by Triton's gfx950
backend) on the MI355X box (TEST INFRASTRUCTURE; "Tier 2" of SURVEY.md §8c).

    python -m oracle.make_ref                       # build container: stage /root/reference -> oracle/_ref
    python -m oracle.ref_triton probe               # GPU box: do the reference kernels JIT and run?
    python -m oracle.ref_triton bench --config c2   # time the reference decode step / prefill
    python -m oracle.ref_triton forward in.pt out.pt    # whole LlamaModel.forward on given weights/prompts
    python -m oracle.ref_triton ops in.pt out.pt        # per-operator outputs on given inputs

Always its own process: the reference package is called `swiftllm`, like this repo's import alias, so the two
must never meet in one interpreter. The harness only makes the reference importable — stub modules for the
three imports that are not installed (`ray`, `vllm_flash_attn`, `swiftllm_c`; the flash-attention stub
forwards to the reference's OWN Triton `prefill_attention`, the drop-in the reference shows commented out at
transformer_layer.py:97-100) — and changes none of its arithmetic. The reference is fp16-only
(model.py:70,147-148).
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
STAGED = os.path.join(HERE, "_ref")

MODEL_DIMS = {
    "llama3-8b": dict(num_hidden_layers=32, hidden_size=4096, num_attention_heads=32, num_key_value_heads=8,
                      intermediate_size=14336, vocab_size=128256, max_position_embeddings=8192,
                      rope_theta=500000.0),
    "llama2-7b": dict(num_hidden_layers=32, hidden_size=4096, num_attention_heads=32, num_key_value_heads=32,
                      intermediate_size=11008, vocab_size=32000, max_position_embeddings=4096,
                      rope_theta=10000.0),
}
# name: (model, batch, context at the first timed step, what BASELINE.json calls it)
BENCH_CONFIGS = {
    "c1": ("llama3-8b", 1, 1024, "configs[1]: batch 1 decode-only"),
    "c2": ("llama3-8b", 32, 1024, "configs[2]: batch 32, 1024-in/128-out"),
    "c3": ("llama2-7b", 4, 16384, "configs[3]: Llama-2-7B dims, batch 4 x 16k context"),
}


def staged_available() -> bool:
    return os.path.isfile(os.path.join(STAGED, "swiftllm", "worker", "model.py"))


def load_reference(dtype: str = "float16"):
    """Import the staged reference as `swiftllm` (compiled Triton: TRITON_INTERPRET must be unset). dtype "bfloat16"
    imports the mechanically patched twin oracle/_ref/bf16 (oracle/make_ref.py: float16 -> bfloat16 everywhere)."""
    global STAGED
    if dtype == "bfloat16":
        STAGED = os.path.join(HERE, "_ref", "bf16")
    if not staged_available():
        raise SystemExit(f"{STAGED}/swiftllm is absent: run `python -m oracle.make_ref` in the build container")
    if os.environ.get("TRITON_INTERPRET"):
        raise SystemExit("TRITON_INTERPRET is set: this harness is for the compiled gfx950 path")
    if "swiftllm" in sys.modules:
        raise SystemExit("a module named swiftllm is already imported (this repo's alias?): use a fresh process")
    import torch
    for name in ("ray", "vllm_flash_attn", "swiftllm_c"):
        m = types.ModuleType(name)
        if name == "ray":
            m.remote = lambda cls: cls
        sys.modules[name] = m
    sys.path.insert(0, STAGED)
    import swiftllm
    assert os.path.dirname(os.path.dirname(os.path.abspath(swiftllm.__file__))) == STAGED, swiftllm.__file__
    from swiftllm.worker.kernels.prefill_attn import prefill_attention as ref_prefill

    def flash_attn_varlen_func(q, k, v, cu_q, cu_k, max_q, max_k, softmax_scale=None, causal=True):
        assert causal
        # model.py:340-343 terminates the start-loc array with num_tokens (prefill + decode); clamp to the rows
        # of q as oracle/gen_golden.py does (the reference's scheduler never emits mixed batches).
        cu_q = cu_q.clone()
        cu_q[-1] = min(int(cu_q[-1]), q.shape[0])
        st = types.SimpleNamespace(
            num_prefill_seqs=cu_q.numel() - 1, max_prefill_len=max_q, softmax_scale=softmax_scale,
            prefill_seq_start_locs=cu_q[:-1].contiguous(),
            prefill_seq_lens=(cu_q[1:] - cu_q[:-1]).contiguous())
        mc = types.SimpleNamespace(num_q_heads=q.shape[1], num_kv_heads=k.shape[1], head_dim=q.shape[2])
        o = torch.empty_like(q)
        ref_prefill(q.contiguous(), k.contiguous(), v.contiguous(), o, mc, None, st)
        return o

    sys.modules["vllm_flash_attn"].flash_attn_varlen_func = flash_attn_varlen_func
    return swiftllm


def _config_dict(model: str, need_positions: int = 0, **over):
    cfg = dict(model_type="llama", hidden_act="silu", rms_norm_eps=1e-5, rope_scaling=None,
               tie_word_embeddings=False)
    cfg.update(MODEL_DIMS[model])
    cfg.update(over)
    have = cfg["max_position_embeddings"]
    if need_positions + 128 > have:
        cfg["rope_scaling"] = float(-(-(need_positions + 128) // have))
    return cfg


def _build_model(swiftllm, cfg: dict, model_path: str, use_dummy: bool, batch: int, max_len: int, num_blocks: int):
    import torch
    ec = swiftllm.EngineConfig(model_path=model_path, use_dummy=use_dummy, block_size=16, gpu_mem_utilization=0.9,
                               num_cpu_blocks=0, max_seqs_in_block_table=max(8, batch),
                               max_blocks_per_seq=max_len // 16 + 8, max_batch_size=batch,
                               max_tokens_in_batch=batch * max_len)
    model = swiftllm.LlamaModel(ec)
    model.load_weights()
    model.init_kvcache_and_swap(num_blocks)
    with torch.inference_mode():
        model.gpu_block_manager.block_table.zero_()     # (torch.empty in the reference: entries past the count)
    return model


def _sync_time(fn):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0


def cmd_probe(_args):
    import torch
    import triton
    swiftllm = load_reference()
    print(f"triton {triton.__version__}, torch {torch.__version__}, device {torch.cuda.get_device_name(0)}")
    cfg = _config_dict("llama3-8b", num_hidden_layers=2, vocab_size=4096)
    path = tempfile.mkdtemp(prefix="ref_probe_")
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f)
    model = _build_model(swiftllm, cfg, path, True, 4, 256, 4 * 20)
    prompts = [[(7 * i + j) % 4096 for j in range(100 + 10 * i)] for i in range(4)]
    toks = model.forward(prompts, [0, 1, 2, 3], [])
    lens = [len(p) for p in prompts]
    for _ in range(3):
        lens = [n + 1 for n in lens]
        toks = model.forward([[t] for t in toks], [0, 1, 2, 3], lens)
    print("reference Triton path ran: prefill + 3 decode steps, tokens", toks)


def cmd_bench(args):
    """Decode tokens/s (and one prefill) of the reference's forward at a BASELINE config: its Triton kernels +
    F.linear (hipBLASLt), fp16, dummy weights re-initialised to N(0, 0.02^2) like bench.py."""
    import torch
    swiftllm = load_reference(args.dtype)
    name, batch, ctx, label = BENCH_CONFIGS[args.config]
    steps, warmup = args.steps, args.warmup
    if args.batch > 0:
        batch = args.batch
    if args.first_context > 0:      # context (incl. the new token) of the first TIMED step
        ctx = args.first_context - warmup
    cfg = _config_dict(name, ctx + steps + warmup + 2)
    path = tempfile.mkdtemp(prefix="ref_bench_")
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f)
    max_len = ctx + steps + warmup + 2
    blocks_per_seq = -(-max_len // 16) + 1
    model = _build_model(swiftllm, cfg, path, True, batch, max_len, batch * blocks_per_seq + 8)
    g = torch.Generator(device="cuda").manual_seed(1234)
    w = model.weight
    tensors = [w.wte, w.lm_head, w.final_norm]
    for layer in w.layers:
        tensors += [t for t in vars(layer).values() if isinstance(t, torch.Tensor)]
    with torch.inference_mode():
        for t in tensors:
            if t.dim() == 1:
                t.normal_(0.0, 0.02, generator=g).add_(1.0)
            else:
                t.normal_(0.0, 0.02, generator=g)
    gp = torch.Generator().manual_seed(1)
    seq_ids = list(range(batch))
    out = dict(config=args.config, workload=label, model=name, batch=batch,
               dtype="f16" if args.dtype == "float16" else "bf16 (float16 -> bfloat16 patched reference)",
               path="reference swiftllm LlamaModel.forward: its Triton kernels (triton gfx950 backend) + F.linear")
    if args.prefill_len > 0:
        plen = args.prefill_len
        prompts = [torch.randint(0, cfg["vocab_size"], (plen,), generator=gp).tolist() for _ in range(batch)]
        model.forward(prompts, seq_ids, [])
        model.free_seqs_resources(seq_ids)
        _, s = _sync_time(lambda: model.forward(prompts, seq_ids, []))
        model.free_seqs_resources(seq_ids)
        out.update(prefill_len=plen, prefill_ms=round(s * 1e3, 2), prefill_tok_s=round(batch * plen / s, 1))
    # decode at context ctx: fill the pool directly (random K/V) instead of running a ctx-token prefill
    lens = [ctx - 1] * batch
    with torch.inference_mode():
        model.gpu_block_manager.allocate_blocks_for_seqs(
            torch.tensor(seq_ids, dtype=torch.int32, device="cuda"),
            torch.tensor(lens, dtype=torch.int32, device="cuda"))
        model.k_cache.normal_(0.0, 1.0, generator=g)
        model.v_cache.normal_(0.0, 1.0, generator=g)
    toks = [1] * batch

    def step():
        nonlocal toks, lens
        lens = [n + 1 for n in lens]
        toks = model.forward([[t] for t in toks], seq_ids, lens)

    for _ in range(warmup):
        step()
    first = lens[0] + 1
    _, s = _sync_time(lambda: [step() for _ in range(steps)])
    out.update(steps=steps, warmup=warmup, context_first=first, context_last=lens[0],
               ms_per_step=round(s / steps * 1e3, 4), decode_tok_s=round(batch * steps / s, 1))
    print(json.dumps(out), flush=True)


def _ulp_of(x, dtype_name: str):
    import torch
    mant = 10 if dtype_name == "float16" else 7
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14))) - mant)


def planned_step(model, logits_log, ids, seq_ids, dec_lens, split=1, reverse=False):
    """One scripted step as `split` forward() calls (sequence i goes to call i % split), optionally with the sequences of
    each group in reverse order; forward() wants prefill sequences before decoding ones, so the order is changed inside
    the two groups only. `logits_log` is the list the lm_head tap appends to. Returns (tokens, logits [B, V]) in the
    SCRIPT's sequence order whatever the plan."""
    import torch
    n, n_dec = len(seq_ids), len(dec_lens)
    n_pre = n - n_dec
    pre, dec = list(range(n_pre)), list(range(n_pre, n))
    if reverse:
        pre, dec = pre[::-1], dec[::-1]
    toks, logits = [None] * n, [None] * n
    for c in range(max(1, split)):
        chunk = pre[c::split] + dec[c::split] if split > 1 else pre + dec
        if not chunk:
            continue
        del logits_log[:]
        t = model.forward([ids[i] for i in chunk], [seq_ids[i] for i in chunk], [dec_lens[i - n_pre] for i in chunk if i >= n_pre])
        lg = logits_log[-1]
        for j, i in enumerate(chunk):
            toks[i], logits[i] = t[j], lg[j]
    return toks, torch.stack(logits)


def cmd_forward(args):
    """in.pt: dict(config=HF config dict, model_path=dir with safetensors, num_blocks, max_len,
    steps=[dict(input_ids, seq_ids, dec_lens)]). out.pt: per step tokens + pre-argmax logits (fp32, CPU).

    Optional keys: `logits` ("fp32" | "storage" | "none"), `logits_steps` (step indices whose full logits are returned; default
    all) and `variants` — the REFERENCE-VS-ITSELF control. Each variant re-runs the same script on the
    same weights under an execution plan that changes nothing in exact arithmetic:
        seq_block_size=N   the flash-decoding split width (model.py:305-324 picks one by heuristic; any power of two >= 64
                           is a legal value of the same kernel, paged_attn.py:35-108) — other partial-softmax merges;
        split=k            the batch as k forward() calls of batch/k sequences each — other GEMM row counts, hence other
                           hipBLASLt kernels / K-split orders (split = batch: every request served alone);
        pad=k              k dummy sequences ride along in every call (for a batch that cannot be split);
        reverse=True       the sequences in reverse order;
        max_steps=n        only the first n scripted steps, teacher-forced only (for plans that cost many calls);
        subset=[i, ...]    only these requests of the batch are served (subset=[0]: the first request alone — batch 1).
    For each variant: the script teacher-forced with the base run's tokens (logit distance and greedy-id mismatches at
    identical histories) and free-running (sequences identical to the end). Summaries go to <out>.variants.json; nothing of
    the reference's arithmetic is touched — only which legal plan it runs."""
    import torch
    job = torch.load(args.inp)
    dtype_name = job.get("dtype", "float16")
    swiftllm = load_reference(dtype_name)
    from swiftllm.worker.layers import post_layer as post_mod
    logits_log = []
    orig_linear = post_mod.linear

    keep = job.get("logits", "fp32")     # "fp32" | "storage" (exact, half the bytes) | "none"
    keep_steps = job.get("logits_steps")
    variants = job.get("variants") or []

    def tapped_linear(a, w):
        r = orig_linear(a, w)
        logits_log.append(r)
        return r
    post_mod.linear = tapped_linear     # the only linear() in post_layer.py is lm_head (:38)
    batch = max(len(s["seq_ids"]) for s in job["steps"])
    pad_max = max([int(v.get("pad", 0)) for v in variants] + [0])
    model = _build_model(swiftllm, job["config"], job["model_path"], False, batch + pad_max, job["max_len"], job["num_blocks"])
    plan = dict(seq_block_size=job.get("seq_block_size"))      # (a job may pin the split width of its BASE run too)
    inner_forward = model._forward
    # `dump_residual_steps`: at these scripted steps, the residual stream after every transformer layer (residual_buf + the
    # layer's FFN output, fp32, [layers, tokens, hidden]) goes to <out>.residual.pt — the per-layer attribution diagnostic
    # of tests/test_gpu_parity_attribution.py. Recording only: the layer's own forward is called unchanged.
    dump_steps = set(job.get("dump_residual_steps") or [])
    residual_log, cur_step = {}, [None]
    if dump_steps:
        for layer in model.transformer_layers:
            def hooked(input_embds, residual_buf, *a, _orig=layer.forward, **kw):
                out = _orig(input_embds, residual_buf, *a, **kw)
                if cur_step[0] in dump_steps:
                    residual_log.setdefault(cur_step[0], []).append((residual_buf.float() + out.float()).cpu())
                return out
            layer.forward = hooked

    def planned_forward(ids, st):
        sbs = plan["seq_block_size"]
        if sbs and st.num_decoding_seqs > 0:
            st.seq_block_size = sbs
            st.num_seq_blocks = (st.max_decoding_len + sbs - 1) // sbs
        return inner_forward(ids, st)
    model._forward = planned_forward

    def run_step(ids, seq_ids, dec_lens, split=1, reverse=False):
        return planned_step(model, logits_log, ids, seq_ids, dec_lens, split, reverse)

    all_seqs = sorted({i for s in job["steps"] for i in s["seq_ids"]})

    def run_script(forced=None, split=1, reverse=False, pad=0, max_steps=None, subset=None):
        """The scripted steps under a plan. `pad` > 0: that many extra DUMMY sequences ride along in every forward (copies
        of the step's first prompt with shifted ids, then their own greedy tokens; outputs dropped) — other GEMM row
        counts for a batch that cannot be split; pure-prefill / pure-decode steps only. `max_steps`: stop early.
        `subset`: only these positions of every step's batch are served (e.g. [0]: the first request alone, batch 1);
        tokens / logits come back for those rows only."""
        all_toks, all_logits = [], []
        dummy_ids = [max(all_seqs) + 1 + j for j in range(pad)]
        dummy_toks, dummy_lens = [], []
        vocab = job["config"]["vocab_size"]
        for si, s in enumerate(job["steps"][:max_steps]):
            ids = s["input_ids"]
            if ids is None:     # "feed back what you sampled": decode continuation
                if forced is not None:
                    prev = forced[si - 1] if subset is None else [forced[si - 1][i] for i in subset]
                else:
                    prev = all_toks[-1]
                ids = [[t] for t in prev]
            elif subset is not None:
                ids = [ids[i] for i in subset]
            seq_ids, dec_lens = list(s["seq_ids"]), list(s["dec_lens"])
            if subset is not None:
                n_pre = len(seq_ids) - len(dec_lens)
                dec_lens = [dec_lens[i - n_pre] for i in subset if i >= n_pre]
                seq_ids = [seq_ids[i] for i in subset]
            if pad:
                assert len(dec_lens) in (0, len(seq_ids)), "pad: pure prefill or pure decode steps only"
                if not dec_lens:
                    ids = list(ids) + [[(t + 1 + j) % vocab for t in ids[0]] for j in range(pad)]
                    dummy_lens = [len(ids[0])] * pad
                else:
                    dummy_lens = [n + 1 for n in dummy_lens]
                    ids = list(ids) + [[t] for t in dummy_toks]
                    dec_lens = dec_lens + dummy_lens
                seq_ids = seq_ids + dummy_ids
            cur_step[0] = si if forced is None and split == 1 and not reverse and not pad and subset is None else None
            t, lg = run_step(ids, seq_ids, dec_lens, split, reverse)
            cur_step[0] = None
            if pad:
                dummy_toks = t[-pad:]
                t, lg = t[:-pad], lg[:-pad]
            all_toks.append(t)
            all_logits.append(lg)
        model.free_seqs_resources(all_seqs + dummy_ids)
        return all_toks, all_logits

    base_toks, base_logits = run_script()
    res = []
    for si, (t, lg) in enumerate(zip(base_toks, base_logits)):
        want = keep != "none" and (keep_steps is None or si in keep_steps)
        res.append(dict(tokens=t, logits=None if not want else (lg.cpu() if keep == "storage" else lg.float().cpu())))
    torch.save(res, args.out)
    print(f"reference forward: {len(res)} steps -> {args.out}")
    if residual_log:
        torch.save({s: torch.stack(v) for s, v in residual_log.items()}, args.out + ".residual.pt")

    if variants:
        # the base run's own decisiveness: smallest top-2 gap over all rows
        gaps = torch.stack([lg.float().topk(2, dim=1).values for lg in base_logits])        # [S, B, 2]
        gap = (gaps[..., 0] - gaps[..., 1])
        summary = dict(dtype=dtype_name, steps=len(base_toks), batch=batch,
                       base=dict(min_top2_gap=float(gap.min()), median_top2_gap=float(gap.median()),
                                 max_abs_logit=float(max(float(lg.float().abs().max()) for lg in base_logits)),
                                 default_seq_block_size="model.py:305-324 heuristic"),
                       variants=[])
        for var in variants:
            rows = var.get("subset")
            kw = dict(split=int(var.get("split", 1)), reverse=bool(var.get("reverse", False)), pad=int(var.get("pad", 0)),
                      max_steps=var.get("max_steps"), subset=rows)
            rows = list(range(batch)) if rows is None else list(rows)       # positions of the base run this plan serves
            plan["seq_block_size"] = var.get("seq_block_size")
            f_toks, f_logits = run_script(forced=base_toks, **kw)
            worst_abs = worst_ulp = 0.0
            mism, not_near_tie, per_step = [], 0, []
            for si, (a, b) in enumerate(zip(f_logits, base_logits)):
                a, b = a.float(), b[rows].float()
                row_abs = (a - b).abs().amax(dim=1)
                row_ulp = row_abs / _ulp_of(b.abs().amax(dim=1), dtype_name)
                worst_abs, worst_ulp = max(worst_abs, float(row_abs.max())), max(worst_ulp, float(row_ulp.max()))
                per_step.append(dict(step=si, max_abs=float(row_abs.max()), max_ulp_of_row=float(row_ulp.max())))
                for j, i in enumerate(rows):
                    if f_toks[si][j] != base_toks[si][i]:
                        g2 = float(gap[si, i])
                        mism.append(dict(step=si, seq=i, ref_top2_gap=g2, row_max_abs=float(row_abs[j])))
                        not_near_tie += int(g2 > 2 * float(row_abs[j]))
            entry = dict(plan=var, teacher_forced=dict(
                max_abs_dlogit=worst_abs, max_ulp_of_row=worst_ulp, token_mismatches=len(mism),
                tokens_compared=len(f_toks) * len(rows), mismatches_not_on_a_near_tie=not_near_tie, mismatches=mism[:32],
                per_step=per_step))
            if kw["max_steps"] is None:     # (a truncated plan is a teacher-forced probe only)
                r_toks, _ = run_script(**kw)
                first_div = []
                for j, i in enumerate(rows):
                    first_div.append(next((s for s in range(len(base_toks)) if r_toks[s][j] != base_toks[s][i]), None))
                diverged = [d for d in first_div if d is not None]
                entry["free_running"] = dict(sequences=len(rows), identical_to_the_end=len(rows) - len(diverged),
                                             first_divergence_steps=first_div,
                                             earliest_divergence_step=min(diverged, default=None))
            summary["variants"].append(entry)
            del f_logits
        plan["seq_block_size"] = None
        with open(args.out + ".variants.json", "w", encoding="utf-8") as f:
            json.dump(summary, f, indent=1)
        print(f"reference self-distance: {len(variants)} plans -> {args.out}.variants.json")


def cmd_ops(args):
    """Per-operator Tier-2 outputs. in.pt: dict of cases as built by tests/test_gpu_reference.py; each case
    names an op of swiftllm/worker/kernels and its CPU input tensors. out.pt: outputs on CPU."""
    import torch
    swiftllm = load_reference()     # noqa: F841
    from swiftllm.worker.kernels.rmsnorm import rmsnorm_inplace, fused_add_rmsnorm_inplace
    from swiftllm.worker.kernels.silu_and_mul import silu_and_mul_inplace
    from swiftllm.worker.kernels.rotary_emb import rotary_embedding_inplace
    from swiftllm.worker.kernels.kvcache_mgmt import store_kvcache
    from swiftllm.worker.kernels.paged_attn import paged_attention
    from swiftllm.worker.kernels.prefill_attn import prefill_attention
    ns = types.SimpleNamespace

    def cu(t):
        return t.cuda() if isinstance(t, torch.Tensor) else t

    cases = torch.load(args.inp)
    out = {}
    for name, c in cases.items():
        op = c["op"]
        t = {k: cu(v) for k, v in c.items()}
        if op == "rmsnorm":
            rmsnorm_inplace(t["x"], t["w"], c["eps"])
            out[name] = dict(x=t["x"].cpu())
        elif op == "fused_add_rmsnorm":
            fused_add_rmsnorm_inplace(t["x"], t["r"], t["w"], c["eps"])
            out[name] = dict(x=t["x"].cpu(), r=t["r"].cpu())
        elif op == "silu_and_mul":
            silu_and_mul_inplace(t["x"])
            out[name] = dict(x=t["x"].cpu())
        elif op == "rotary":
            rotary_embedding_inplace(t["q"], t["k"], ns(position_cos=t["cos"], position_sin=t["sin"]))
            out[name] = dict(q=t["q"].cpu(), k=t["k"].cpu())
        elif op == "paged_attention":
            H, KVH, D, L = c["H"], c["KVH"], c["D"], c["L"]
            mc = ns(num_layers=L, num_q_heads=H, num_kv_heads=KVH, head_dim=D)
            ec = ns(block_size=16, max_blocks_per_seq=t["block_table"].shape[1])
            lens = c["lens"]
            sbs = c["seq_block_size"]
            st = ns(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs,
                    num_seq_blocks=(max(lens) + sbs - 1) // sbs, softmax_scale=D ** -0.5,
                    decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device="cuda"),
                    seq_ids=torch.tensor(c["seq_ids"], dtype=torch.int32, device="cuda"))
            o = torch.zeros_like(t["q"])
            paged_attention(t["q"], t["k_cache"], t["v_cache"], t["block_table"], mc, ec, st, c["layer"], o)
            out[name] = dict(o=o.cpu())
        elif op == "prefill_attention":
            H, KVH, D = c["H"], c["KVH"], c["D"]
            lens_t = torch.tensor(c["lens"], dtype=torch.int32, device="cuda")
            st = ns(num_prefill_seqs=len(c["lens"]), max_prefill_len=max(c["lens"]), softmax_scale=D ** -0.5,
                    prefill_seq_start_locs=(torch.cumsum(lens_t, 0, dtype=torch.int32) - lens_t),
                    prefill_seq_lens=lens_t)
            o = torch.zeros_like(t["q"])
            prefill_attention(t["q"], t["k"], t["v"], o, ns(num_q_heads=H, num_kv_heads=KVH, head_dim=D), None, st)
            out[name] = dict(o=o.cpu())
        elif op == "store_kvcache":
            mc = ns(num_layers=c["L"], num_kv_heads=c["KVH"], head_dim=c["D"])
            ec = ns(block_size=16, max_blocks_per_seq=t["block_table"].shape[1])
            pl = torch.tensor(c["plens"], dtype=torch.int32, device="cuda")
            st = ns(seq_ids=t["seq_ids"], num_prefill_seqs=len(c["plens"]), num_prefill_tokens=sum(c["plens"]),
                    max_prefill_len=max(c["plens"]) if c["plens"] else 0, prefill_seq_lens=pl,
                    prefill_seq_start_locs=(torch.cumsum(pl, 0, dtype=torch.int32) - pl),
                    num_decoding_seqs=len(c["dlens"]),
                    decoding_seq_lens=torch.tensor(c["dlens"], dtype=torch.int32, device="cuda"))
            store_kvcache(t["k"], t["v"], t["k_cache"], t["v_cache"], t["block_table"], mc, ec, st, c["layer"])
            out[name] = dict(k_cache=t["k_cache"].cpu(), v_cache=t["v_cache"].cpu())
        else:
            raise SystemExit(f"unknown op {op}")
    torch.cuda.synchronize()
    torch.save(out, args.out)
    print(f"reference ops: {len(out)} cases -> {args.out}")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    sub.add_parser("probe")
    b = sub.add_parser("bench")
    b.add_argument("--config", choices=sorted(BENCH_CONFIGS), default="c2")
    b.add_argument("--steps", type=int, default=20)
    b.add_argument("--warmup", type=int, default=5)
    b.add_argument("--prefill-len", type=int, default=0)
    b.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"],
                   help="bfloat16 = the mechanically patched twin under oracle/_ref/bf16")
    b.add_argument("--batch", type=int, default=0, help="override the config's batch size")
    b.add_argument("--first-context", type=int, default=0,
                   help="context (including the new token) of the first timed step (default: the config's + warmup)")
    for name in ("forward", "ops"):
        p = sub.add_parser(name)
        p.add_argument("inp")
        p.add_argument("out")
    args = ap.parse_args()
    {"probe": cmd_probe, "bench": cmd_bench, "forward": cmd_forward, "ops": cmd_ops}[args.cmd](args)


if __name__ == "__main__":
    main()
