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
