#!/usr/bin/env python3
"""weight_load_bench.py — checkpoint -> HBM load rate (§8f rank 4): the reference's tensor-by-tensor safetensors
path vs the streaming loader, on a Llama-3-8B-shaped checkpoint with --layers layers written to --dir.
Both are timed with the file already in the page cache (second read); reports GB/s."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from swiftllm_amd.model_config import LlamaModelConfig
from swiftllm_amd.worker.weight import load_weights

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--dir", default="/tmp/swl_ckpt")
a = ap.parse_args()
cfg = bench.model_config_dict("llama3-8b")
cfg["num_hidden_layers"] = a.layers
os.makedirs(a.dir, exist_ok=True)
with open(os.path.join(a.dir, "config.json"), "w") as f:
    json.dump(cfg, f)
h, kv, inter, v = 4096, 1024, 14336, cfg["vocab_size"]
g = torch.Generator().manual_seed(0)
def rnd(*shape):
    return torch.randint(-30000, 30000, shape, dtype=torch.int16, generator=g).view(torch.bfloat16)
from safetensors.torch import save_file
t0 = time.perf_counter()
shards, per = {}, 4
sd = {"model.embed_tokens.weight": rnd(v, h), "lm_head.weight": rnd(v, h), "model.norm.weight": rnd(h)}
files = {"model-00000.safetensors": sd}
for l in range(a.layers):
    p = f"model.layers.{l}."
    d = files.setdefault(f"model-{1 + l // per:05d}.safetensors", {})
    d.update({p + "input_layernorm.weight": rnd(h), p + "post_attention_layernorm.weight": rnd(h),
              p + "self_attn.q_proj.weight": rnd(h, h), p + "self_attn.k_proj.weight": rnd(kv, h),
              p + "self_attn.v_proj.weight": rnd(kv, h), p + "self_attn.o_proj.weight": rnd(h, h),
              p + "mlp.up_proj.weight": rnd(inter, h), p + "mlp.gate_proj.weight": rnd(inter, h),
              p + "mlp.down_proj.weight": rnd(h, inter)})
index = {"weight_map": {k: name for name, d in files.items() for k in d}}
total = 0
for name, d in files.items():
    save_file(d, os.path.join(a.dir, name))
    total += sum(t.numel() * 2 for t in d.values())
with open(os.path.join(a.dir, "model.safetensors.index.json"), "w") as f:
    json.dump(index, f)
keep = {k: files["model-00001.safetensors"][k].clone() for k in list(files["model-00001.safetensors"])[:3]}
del files, sd
print(f"wrote {total / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s", flush=True)

mc = LlamaModelConfig.load_from_model_path(a.dir)
res = {"GB": round(total / 1e9, 2), "layers": a.layers}
for name, streaming in (("warm", True), ("per_tensor", False), ("streaming", True)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w = load_weights(mc, torch.bfloat16, a.dir, device="cuda", fuse_qkv=True, streaming=streaming)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[name + "_s"] = round(dt, 3)
    res[name + "_GBps"] = round(total / dt / 1e9, 2)
    q = keep["model.layers.0.self_attn.q_proj.weight"]
    assert torch.equal(w.layers[0].qkv_proj[:h].cpu().view(torch.int16), q.view(torch.int16))   # (bit patterns: NaNs inside)
    del w
    torch.cuda.empty_cache()
print(json.dumps(res), flush=True)
