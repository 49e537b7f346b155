"""oracle/synth.py — synthetic LLaMA checkpoints (TEST INFRASTRUCTURE).

There are no real weights on disk and no network, so every test and bench runs on random-init models
written by this helper: an HF-style config.json plus one safetensors file, seeded, matrices
~ N(0, 0.02^2), norm weights = 1 + N(0, 0.02^2) (the reference's own dummy init
U(-1e-3, 1e-3), weight.py:217, makes every logit ~0 and argmax meaningless).
"""
import json
import os

import torch

# BASELINE.json configs[0]: 2-layer / 128-dim LLaMA
TINY = dict(num_hidden_layers=2, hidden_size=128, num_attention_heads=4, num_key_value_heads=2,
            intermediate_size=256, vocab_size=256, max_position_embeddings=512)
# same width, head_dim 64 / 128 variants for kernel-shape coverage
SMALL64 = dict(num_hidden_layers=2, hidden_size=256, num_attention_heads=4, num_key_value_heads=2,
               intermediate_size=512, vocab_size=512, max_position_embeddings=1024)
SMALL128 = dict(num_hidden_layers=2, hidden_size=512, num_attention_heads=4, num_key_value_heads=1,
                intermediate_size=1024, vocab_size=512, max_position_embeddings=2048)
LLAMA3_8B = dict(num_hidden_layers=32, hidden_size=4096, num_attention_heads=32,
                 num_key_value_heads=8, intermediate_size=14336, vocab_size=128256,
                 max_position_embeddings=8192, rope_theta=500000.0, rms_norm_eps=1e-5)
LLAMA2_7B = dict(num_hidden_layers=32, hidden_size=4096, num_attention_heads=32,
                 num_key_value_heads=32, intermediate_size=11008, vocab_size=32000,
                 max_position_embeddings=4096, rope_theta=10000.0, rms_norm_eps=1e-5)


def make_config(**overrides) -> dict:
    cfg = dict(model_type="llama", hidden_act="silu", rms_norm_eps=1e-5, rope_theta=10000.0,
               rope_scaling=None, tie_word_embeddings=False)
    cfg.update(TINY)
    cfg.update(overrides)
    return cfg


def make_state_dict(cfg: dict, seed: int = 0, dtype=torch.float16, std: float = 0.02) -> dict:
    g = torch.Generator().manual_seed(seed)
    h, inter, v = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    kv = cfg.get("num_key_value_heads", cfg["num_attention_heads"]) * (h // cfg["num_attention_heads"])

    def mat(*shape):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    def norm(n):
        return (1.0 + torch.randn(n, generator=g) * std).to(dtype)

    sd = {"model.embed_tokens.weight": mat(v, h), "lm_head.weight": mat(v, h),
          "model.norm.weight": norm(h)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"] = norm(h)
        sd[p + "self_attn.q_proj.weight"] = mat(h, h)
        sd[p + "self_attn.k_proj.weight"] = mat(kv, h)
        sd[p + "self_attn.v_proj.weight"] = mat(kv, h)
        sd[p + "self_attn.o_proj.weight"] = mat(h, h)
        sd[p + "post_attention_layernorm.weight"] = norm(h)
        sd[p + "mlp.up_proj.weight"] = mat(inter, h)
        sd[p + "mlp.gate_proj.weight"] = mat(inter, h)
        sd[p + "mlp.down_proj.weight"] = mat(h, inter)
    return sd


def make_state_dict_on_gpu(cfg: dict, seed: int = 0, dtype=torch.float16, std: float = 0.02, lm_head_std: float = None) -> dict:
    """make_state_dict for full-size models (Llama-3-8B: 8 G parameters): the same tensor names, shapes and
    distributions, drawn on the GPU (a CPU generator needs minutes for 16 GB) and returned as CPU tensors. NOT the
    same values as make_state_dict(seed): both sides of a comparison must load the file this dict is written to."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    h, inter, v = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    kv = cfg.get("num_key_value_heads", cfg["num_attention_heads"]) * (h // cfg["num_attention_heads"])

    def mat(*shape, s=std):
        return (torch.randn(*shape, generator=g, device="cuda") * s).to(dtype).cpu()

    def norm(n):
        return (1.0 + torch.randn(n, generator=g, device="cuda") * std).to(dtype).cpu()

    sd = {"model.embed_tokens.weight": mat(v, h), "lm_head.weight": mat(v, h, s=lm_head_std or std),
          "model.norm.weight": norm(h)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"] = norm(h)
        sd[p + "self_attn.q_proj.weight"] = mat(h, h)
        sd[p + "self_attn.k_proj.weight"] = mat(kv, h)
        sd[p + "self_attn.v_proj.weight"] = mat(kv, h)
        sd[p + "self_attn.o_proj.weight"] = mat(h, h)
        sd[p + "post_attention_layernorm.weight"] = norm(h)
        sd[p + "mlp.up_proj.weight"] = mat(inter, h)
        sd[p + "mlp.gate_proj.weight"] = mat(inter, h)
        sd[p + "mlp.down_proj.weight"] = mat(h, inter)
    return sd


EXAMPLE_PROMPTS = ["Life blooms like a flower, far away", "one two three four five",
                   "A B C D E F G H I J K L M N O P Q R S T U V", "To be or not to be,"]


def write_tokenizer(path: str, vocab_size: int):
    """A word-level tokenizer (tokenizer.json + tokenizer_config.json, loadable by transformers.AutoTokenizer with no
    network) whose vocabulary holds the words of the reference's example prompts (examples/offline.py:47-52,
    examples/online.py:66-71) and filler words up to `vocab_size`: every id a `vocab_size`-token model can emit
    decodes to a word."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    words = ["<unk>", "<s>", "</s>"] + sorted(set(" ".join(EXAMPLE_PROMPTS).split()))
    assert len(words) <= vocab_size
    vocab = {w: i for i, w in enumerate(words)}
    for i in range(len(vocab), vocab_size):
        vocab[f"w{i}"] = i
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.decoder = decoders.WordPiece(prefix="##")       # joins words with single spaces
    os.makedirs(path, exist_ok=True)
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w", encoding="utf-8") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "unk_token": "<unk>", "bos_token": "<s>",
                   "eos_token": "</s>", "model_max_length": 1 << 20}, f)
    return path


def write_model_dir(path: str, cfg: dict, state_dict: dict = None, fmt: str = "safetensors"):
    """config.json (+ weights) in `path`. state_dict=None writes the config only (use_dummy runs)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f)
    if state_dict is None:
        return path
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in state_dict.items()},
                  os.path.join(path, "model.safetensors"))
    else:
        torch.save(state_dict, os.path.join(path, "pytorch_model.bin"))
    return path


def paged_setup(g, num_blocks, L, KVH, bs, D, seq_ids, lens, max_seqs=8, mbps=64):
    """Random KV pool + a block table that scatters each sequence's blocks (non-monotonic ids)."""
    k_cache = torch.randn(num_blocks, L, KVH, bs, D, generator=g).half()
    v_cache = torch.randn(num_blocks, L, KVH, bs, D, generator=g).half()
    perm = torch.randperm(num_blocks, generator=g).tolist()
    block_table = torch.zeros(max_seqs, mbps, dtype=torch.int32)
    for sid, ln in zip(seq_ids, lens):
        for j in range((ln + bs - 1) // bs):
            block_table[sid, j] = perm.pop()
    return k_cache, v_cache, block_table


def seeded_paged_case(seed, H, KVH, D, L, lens, bs=16, max_seqs=4, mbps=80):
    """K/V pool, block table and q of a decode-attention case as a pure function of `seed` (CPU generator): large
    cases are committed as seed + outputs only, tests regenerate the inputs and check `kv_checksum`."""
    g = torch.Generator().manual_seed(seed)
    seq_ids = list(range(1, 1 + len(lens)))
    nblk = sum((n + bs - 1) // bs for n in lens) + 3
    k_cache, v_cache, bt = paged_setup(g, nblk, L, KVH, bs, D, seq_ids, lens, max_seqs=max_seqs, mbps=mbps)
    q = torch.randn(len(lens), H, D, generator=g).half()
    checksum = float(k_cache.double().sum() + 3.0 * v_cache.double().sum() + 7.0 * bt.double().sum())
    return seq_ids, k_cache, v_cache, bt, q, checksum


# ---- a checkpoint whose greedy decisions are DECISIVE ------------------------------------------
def _rope_inv_freq(cfg: dict) -> torch.Tensor:
    """worker/model.py:177-225, scalar-scaling branch (eager_ops.rope_tables): angle of pair j at position p = p * inv_freq[j]
    / scaling."""
    dim = cfg["hidden_size"] // cfg["num_attention_heads"]
    scaling = cfg.get("rope_scaling") or 1.0
    assert not isinstance(scaling, dict)
    return (1.0 / (cfg.get("rope_theta", 10000.0) ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))) / scaling


def make_decisive_state_dict(cfg: dict, seed: int = 0, dtype=torch.float16, offset: int = 19, margin: float = 4.0,
                             peak_logit: float = 0.85, max_context: int = 1400, device: str = "cpu",
                             body_std: float = 0.02, final_norm_scale: float = 1.0 / 16):
    """A LLaMA checkpoint of ANY geometry whose greedy token stream is a known function of its input and whose top-2
    logit gap is hundreds of storage-dtype ulps wide — the model on which "greedy token ids bit-exact" is a fair
    assert for two 16-bit implementations (a random-init network's top-2 gap is below one ulp on a few rows out of every
    few thousand, whatever the implementation).

    Construction — every operator of the forward is on the decisive path or perturbs the logits through it:
      * hidden = [ A: token code (h/2 dims, N(0,1)) | B: zero (h/2 - 1 dims) | one constant "bias" dim ];
      * layer 0 attention is a positional COPY head: W_q / W_k read only the bias dim and are shaped so that, after the
        reference's rotate-half rotary (rotary_emb.py:26-42), q_p . k_s = c * sum_j cos((p - s - offset) * theta_j) over the
        highest-frequency pairs — a softmax peak `margin` nats above its neighbours at s = p - offset; W_v is a random
        projection of the A dims, W_o writes it into the B dims. The token at position p - offset therefore arrives, as a
        random code u(token), in the B dims of position p — through rotary, the paged KV store, prefill attention (prompt)
        and paged decode attention (generated tokens; `offset` = 19 crosses a 16-token block and, after 19 steps, reads
        K/V the decode path stored itself);
      * lm_head[perm[v]] = tau * u(v) on the B dims: the greedy token after position p is perm[token(p - offset)], with
        the runner-up at ~0.2 of the peak logit;
      * every other weight (layer 0's MLP, layers 1..L-1) is N(0, body_std^2) with o_proj / down_proj scaled by
        1/sqrt(2L) (a residual stream that grows like a trained model's instead of being re-randomised by every
        layer): they move every logit, not the decision.
    `peak_logit` < 1 keeps every fp16 logit in the range where the north star's absolute 1e-3 is two ulps. The final norm
    weight is ~`final_norm_scale` (and lm_head correspondingly larger) so that lm_head's entries (~5e-3) stay clear of float16's
    subnormal range (< 6.1e-5), where a CPU GEMM and a matrix-core GEMM may legitimately treat inputs differently.

    Returns (state_dict of CPU tensors, perm as a CPU int64 tensor, info dict)."""
    import math
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    h, inter, v, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    H = cfg["num_attention_heads"]
    KVH = cfg.get("num_key_value_heads", H)
    D = h // H
    hA, bias = h // 2, h - 1
    hB = h - 1 - hA
    beta = 0.25 * math.sqrt(h)
    res_scale = 1.0 / math.sqrt(2 * L)

    def rnd(*shape, s=1.0):
        return torch.randn(*shape, generator=g, device=dev) * s

    def out(t):
        return t.to(dtype).cpu()

    def norm_w(n):
        return 1.0 + rnd(n, s=0.02)

    # ---- positional copy head: which rotary pairs, and how sharp ------------------------------------------------------
    theta = _rope_inv_freq(cfg)
    d = torch.arange(-max_context, max_context + 1, dtype=torch.float64)
    best = None
    for J in range(4, D // 2 + 1):       # the J highest-frequency pairs; fewest that leave no alias within the context
        f = torch.cos(d[:, None] * theta[None, :J]).sum(1)
        f0 = float(J)
        f[max_context] = -1e9            # d = 0 itself
        gap1 = f0 - float(torch.cos(theta[:J]).sum())        # to the neighbours d = +-1
        gap_far = f0 - float(f.max())
        if gap_far >= 0.999 * gap1:
            best = (J, gap1)
            break
    assert best is not None, "no alias-free set of rotary pairs for this geometry"
    J, gap1 = best
    amp2 = margin * math.sqrt(D) / gap1          # (a * gamma)^2: neighbours sit `margin` nats below the peak
    embed = torch.zeros(v, h, device=dev)
    embed[:, :hA] = rnd(v, hA)
    embed[:, bias] = beta
    n0 = norm_w(h)
    # what layer 0 sees: rmsnorm(embedding) * n0 (the residual entering layer 0 is the embedding itself, model.py:228-249)
    emb_r = embed.to(dtype).float()
    n0_r = n0.to(dtype).float()
    rstd = torch.rsqrt((emb_r * emb_r).mean(dim=1, keepdim=True) + cfg.get("rms_norm_eps", 1e-5))
    x0 = (emb_r * rstd * n0_r).to(dtype).float()                     # [v, h], rounded as the kernels round it
    gamma = float(x0[:, bias].mean())
    a = math.sqrt(amp2) / gamma
    qv = torch.zeros(D, dtype=torch.float64)
    kv_ = torch.zeros(D, dtype=torch.float64)
    qv[:J] = torch.cos(offset * theta[:J])
    qv[D // 2:D // 2 + J] = -torch.sin(offset * theta[:J])
    kv_[:J] = 1.0
    wq = torch.zeros(H * D, h, device=dev)
    wk = torch.zeros(KVH * D, h, device=dev)
    wq[:, bias] = (a * qv).float().to(dev).repeat(H)
    wk[:, bias] = (a * kv_).float().to(dev).repeat(KVH)
    wv = torch.zeros(KVH * D, h, device=dev)
    wv[:, :hA] = rnd(KVH * D, hA, s=1.0 / math.sqrt(hA))
    wo = torch.zeros(h, H * D, device=dev)
    u_std = 2.0
    v_std = float((x0[:256, :hA] @ wv[:, :hA].to(dtype).float().T).std())
    wo[hA:hA + hB] = rnd(hB, H * D, s=u_std / (math.sqrt(H * D) * v_std))
    # u(token): the code layer 0 deposits for a token, through the ROUNDED weights, q-head i reading kv-head i // (H/KVH)
    wv_r, wo_r = wv.to(dtype).float(), wo.to(dtype).float()
    code_v = (x0 @ wv_r.T).to(dtype).float()                                  # [v, KVH*D]
    rep = code_v.view(v, KVH, 1, D).expand(v, KVH, H // KVH, D).reshape(v, H * D)
    u = (rep @ wo_r[hA:hA + hB].T)                                            # [v, hB]
    # final hidden ~ embedding + u + body noise (variance ~1 per dim): tau puts the peak logit at `peak_logit`
    rms_f = math.sqrt((hA * 2.0 + hB * (u_std ** 2 + 1.0) + beta ** 2) / h)
    tau = peak_logit * rms_f / float((u * u).sum(1).mean()) / final_norm_scale
    perm = torch.randperm(v, generator=g, device=dev)
    lm_head = torch.zeros(v, h, device=dev)
    lm_head[perm, hA:hA + hB] = tau * u
    del u, rep, code_v, x0, emb_r

    sd = {"model.embed_tokens.weight": out(embed), "lm_head.weight": out(lm_head),
          "model.norm.weight": out(norm_w(h) * final_norm_scale)}
    del embed, lm_head
    kvd = KVH * D
    for i in range(L):
        p = f"model.layers.{i}."
        if i == 0:
            sd[p + "input_layernorm.weight"] = out(n0)
            sd[p + "self_attn.q_proj.weight"] = out(wq)
            sd[p + "self_attn.k_proj.weight"] = out(wk)
            sd[p + "self_attn.v_proj.weight"] = out(wv)
            sd[p + "self_attn.o_proj.weight"] = out(wo)
        else:
            sd[p + "input_layernorm.weight"] = out(norm_w(h))
            sd[p + "self_attn.q_proj.weight"] = out(rnd(h, h, s=body_std))
            sd[p + "self_attn.k_proj.weight"] = out(rnd(kvd, h, s=body_std))
            sd[p + "self_attn.v_proj.weight"] = out(rnd(kvd, h, s=body_std))
            sd[p + "self_attn.o_proj.weight"] = out(rnd(h, h, s=body_std * res_scale))
        sd[p + "post_attention_layernorm.weight"] = out(norm_w(h))
        sd[p + "mlp.up_proj.weight"] = out(rnd(inter, h, s=body_std))
        sd[p + "mlp.gate_proj.weight"] = out(rnd(inter, h, s=body_std))
        sd[p + "mlp.down_proj.weight"] = out(rnd(h, inter, s=body_std * res_scale))
    info = dict(offset=offset, margin_nats=margin, rotary_pairs=J, peak_score_nats=amp2 * J / math.sqrt(D),
                peak_logit_target=peak_logit, tau=tau, res_scale=res_scale)
    return sd, perm.cpu(), info


def decisive_expected_tokens(prompts, perm, offset: int, steps: int):
    """The greedy stream make_decisive_state_dict's model must produce: token after position p = perm[token(p - offset)]
    (prompts must be longer than `offset`). Returns steps+1 lists (the token after the prompt, then one per decode step)."""
    seqs = [list(p) for p in prompts]
    outs = []
    for _ in range(steps + 1):
        new = [int(perm[s[len(s) - 1 - offset]]) for s in seqs]
        outs.append(new)
        for s, t in zip(seqs, new):
            s.append(t)
    return outs
