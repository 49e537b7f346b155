"""LlamaModelConfig — architecture hyper-parameters read from a HuggingFace `config.json`.

Mirrors swiftllm/model_config.py:5-46 (attribute names, defaults, kv-slot size formula)."""
import json
import os

import torch


class LlamaModelConfig:
    """LLaMA 1/2/3 configuration. Attribute names follow the reference so layer/kernel code and
    user scripts that poke at `model.model_config` keep working."""

    def __init__(self, model_config: dict):
        if model_config["model_type"] != "llama":
            raise AssertionError(f"unsupported model_type {model_config['model_type']!r}")
        if model_config["hidden_act"] != "silu":
            raise AssertionError(f"unsupported hidden_act {model_config['hidden_act']!r}")
        self.num_layers = model_config["num_hidden_layers"]
        self.num_q_heads = model_config["num_attention_heads"]
        self.num_kv_heads = model_config.get("num_key_value_heads", self.num_q_heads)
        self.hidden_size = model_config["hidden_size"]
        self.head_dim = self.hidden_size // self.num_q_heads
        self.vocab_size = model_config["vocab_size"]
        self.max_position_embeddings = model_config["max_position_embeddings"]
        self.ffn_inter_dim = model_config["intermediate_size"]
        self.rope_theta = model_config.get("rope_theta", 10000)
        self.rotary_base = model_config.get("rope_theta", model_config.get("rotary_base", 10000))
        self.rms_norm_eps = model_config["rms_norm_eps"]
        scaling = model_config.get("rope_scaling", 1.0)
        self.rope_scaling = 1.0 if scaling is None else scaling

    def get_kvslot_size(self, dtype: torch.dtype = torch.float16) -> int:
        """Bytes of KV cache one token occupies (K and V, all layers)."""
        return 2 * self.num_layers * self.num_kv_heads * self.head_dim * dtype.itemsize

    @staticmethod
    def load_from_model_path(model_path: str) -> "LlamaModelConfig":
        with open(os.path.join(model_path, "config.json"), "r", encoding="utf-8") as f:
            return LlamaModelConfig(json.load(f))
