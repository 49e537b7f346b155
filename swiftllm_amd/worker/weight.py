"""Weight containers and checkpoint readers for LLaMA.

Behaviour follows swiftllm/worker/weight.py:10-272: the same HuggingFace tensor names, the same
shape checks, safetensors (single file or `model.safetensors.index.json`) preferred over
`pytorch_model.bin` (single or indexed, mmap), dummy weights ~ U(-1e-3, 1e-3) for profiling runs,
`up_proj`/`gate_proj` concatenated into `up_gate_proj` = [up ; gate] (weight.py:133), and the
"rope_scaling is a dict => Llama-3.2 => lm_head tied to the embedding" sniff (weight.py:199-213).
Differences: dtype and device are parameters (the reference hard-codes fp16 / "cuda"), and q/k/v can
additionally be concatenated into one `qkv_proj` for a single fused projection GEMM.
"""
import json
import os
from typing import Callable, Dict, List, NamedTuple, Tuple

import torch


class WeightSpec(NamedTuple):
    attr: str               # attribute the tensor is stored under
    key: str                # tensor name in the checkpoint
    shape: Tuple[int, ...]


def layer_weight_specs(cfg, layer_id: int) -> List[WeightSpec]:
    h, kv = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim
    inter = cfg.ffn_inter_dim
    pre = f"model.layers.{layer_id}."
    return [
        WeightSpec("attn_norm", pre + "input_layernorm.weight", (h,)),
        WeightSpec("q_proj", pre + "self_attn.q_proj.weight", (h, h)),
        WeightSpec("k_proj", pre + "self_attn.k_proj.weight", (kv, h)),
        WeightSpec("v_proj", pre + "self_attn.v_proj.weight", (kv, h)),
        WeightSpec("o_proj", pre + "self_attn.o_proj.weight", (h, h)),
        WeightSpec("ffn_norm", pre + "post_attention_layernorm.weight", (h,)),
        WeightSpec("up_proj", pre + "mlp.up_proj.weight", (inter, h)),
        WeightSpec("gate_proj", pre + "mlp.gate_proj.weight", (inter, h)),
        WeightSpec("down_proj", pre + "mlp.down_proj.weight", (h, inter)),
    ]


def global_weight_specs(cfg, model_version: str) -> List[WeightSpec]:
    v, h = cfg.vocab_size, cfg.hidden_size
    head_key = "model.embed_tokens.weight" if model_version == "llama3.2" else "lm_head.weight"
    return [
        WeightSpec("wte", "model.embed_tokens.weight", (v, h)),
        WeightSpec("lm_head", head_key, (v, h)),
        WeightSpec("final_norm", "model.norm.weight", (h,)),
    ]


Getter = Callable[[WeightSpec], torch.Tensor]


def _fill(obj, specs: List[WeightSpec], getter: Getter, dtype: torch.dtype, device: torch.device):
    for spec in specs:
        t = getter(spec)
        assert isinstance(t, torch.Tensor), f"Weight {spec.key} is not a tensor"
        assert tuple(t.shape) == tuple(spec.shape), \
            f"Shape of weight {spec.key} does not match: {tuple(t.shape)} vs {spec.shape}"
        setattr(obj, spec.attr, t.to(device=device, dtype=dtype).contiguous())


class LlamaTransformerLayerWeight:
    """Weights of one transformer block: attn_norm, q/k/v/o_proj, ffn_norm, up_gate_proj, down_proj
    (and qkv_proj = [q ; k ; v] when fused)."""

    def __init__(self, layer_id: int, model_config, dtype: torch.dtype):
        self.layer_id = layer_id
        self.model_config = model_config
        self.dtype = dtype
        self.qkv_proj = None

    def load(self, getter: Getter, device: torch.device, fuse_qkv: bool):
        _fill(self, layer_weight_specs(self.model_config, self.layer_id), getter, self.dtype, device)
        self.up_gate_proj = torch.cat((self.up_proj, self.gate_proj), dim=0).contiguous()
        del self.up_proj, self.gate_proj
        if fuse_qkv:
            self.qkv_proj = torch.cat((self.q_proj, self.k_proj, self.v_proj), dim=0).contiguous()
            del self.q_proj, self.k_proj, self.v_proj


class LlamaWeight:
    """All weights of the model: wte, lm_head, final_norm and `layers`."""

    def __init__(self, model_config, dtype: torch.dtype, model_version: str = "llama"):
        self.model_config = model_config
        self.dtype = dtype
        self.model_version = model_version
        self.layers = [LlamaTransformerLayerWeight(i, model_config, dtype)
                       for i in range(model_config.num_layers)]

    def load(self, getter: Getter, device: torch.device, fuse_qkv: bool = False):
        _fill(self, global_weight_specs(self.model_config, self.model_version), getter, self.dtype,
              device)
        if self.model_version == "llama3.2":
            self.lm_head = self.wte     # tied: one copy in HBM
        for layer in self.layers:
            layer.load(getter, device, fuse_qkv)


# ---- checkpoint readers --------------------------------------------------------------------------
def _dummy_getter(dtype: torch.dtype, device: torch.device) -> Getter:
    def get(spec: WeightSpec) -> torch.Tensor:
        return torch.empty(spec.shape, dtype=dtype, device=device).uniform_(-0.001, 0.001)
    return get


def _safetensors_getter(model_path: str, files: List[str], device: torch.device) -> Getter:
    import safetensors
    index_path = os.path.join(model_path, "model.safetensors.index.json")
    if os.path.exists(index_path):
        with open(index_path, "r", encoding="utf-8") as f:
            where: Dict[str, str] = json.load(f)["weight_map"]
        locate = where.__getitem__
    else:
        assert len(files) == 1, \
            "model.safetensors.index.json not found, but there are multiple .safetensors files"
        locate = lambda key: files[0]   # noqa: E731
    dev = str(device)

    def get(spec: WeightSpec) -> torch.Tensor:
        # opening a safetensors file only parses its header: cheap enough to do per tensor
        with safetensors.safe_open(os.path.join(model_path, locate(spec.key)), framework="pt",
                                   device=dev) as f:
            return f.get_tensor(spec.key)
    return get


def _torch_bin_getter(model_path: str, device: torch.device) -> Getter:
    index_path = os.path.join(model_path, "pytorch_model.bin.index.json")
    if os.path.exists(index_path):
        with open(index_path, "r", encoding="utf-8") as f:
            where: Dict[str, str] = json.load(f)["weight_map"]
        locate = where.__getitem__
    else:
        locate = lambda key: "pytorch_model.bin"    # noqa: E731
    opened: Dict[str, dict] = {}    # unpickling is slow: each shard is opened once (mmap)

    def get(spec: WeightSpec) -> torch.Tensor:
        name = locate(spec.key)
        if name not in opened:
            opened[name] = torch.load(os.path.join(model_path, name), map_location="cpu", mmap=True,
                                      weights_only=True)
        return opened[name][spec.key]
    return get


def detect_model_version(model_path: str) -> str:
    """'llama3.2' when config.json carries a dict-valued rope_scaling, else 'llama'."""
    cfg_path = os.path.join(model_path, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path, "r", encoding="utf-8") as f:
            if isinstance(json.load(f).get("rope_scaling"), dict):
                return "llama3.2"
    return "llama"


def load_weights(model_config, dtype: torch.dtype, model_path: str, use_dummy: bool = False,
                 model_version: str = "auto", device="cuda", fuse_qkv: bool = False) -> LlamaWeight:
    """Read (or synthesise) every weight of the model onto `device` in `dtype`."""
    device = torch.device(device)
    if model_version == "auto":
        model_version = detect_model_version(model_path)
    if use_dummy:
        getter = _dummy_getter(dtype, device)
    else:
        st_files = sorted(n for n in os.listdir(model_path) if n.endswith(".safetensors"))
        getter = (_safetensors_getter(model_path, st_files, device) if st_files
                  else _torch_bin_getter(model_path, device))
    weight = LlamaWeight(model_config, dtype, model_version)
    weight.load(getter, device, fuse_qkv)
    return weight
