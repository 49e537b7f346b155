#!/usr/bin/env python3
"""Upper bound for fusing the three latency-bound kernels of a decode layer into their neighbours:
run the default bench with those launches simply SKIPPED (outputs are garbage, timing only).
   SKIP=norm,rope python tools/probe/skip_small.py [bench args]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from swiftllm_amd.worker.layers import transformer_layer as tl

skip = set(os.environ.get("SKIP", "norm,rope").split(","))
_bufs = {}

def _buf(key, shape, dtype, device):
    t = _bufs.get(key)
    if t is None or t.shape != shape:
        t = torch.zeros(shape, dtype=dtype, device=device)
        _bufs[key] = t
    return t

if "norm" in skip:
    def fake_norm(part, residual, w, eps):
        return _buf("norm", part.shape, part.dtype, part.device)
    tl.fused_add_rmsnorm_from_splitk = fake_norm
if "rope" in skip:
    def fake_rope(part, k_cache, v_cache, block_table, cfg, ecfg, st, layer_id):
        m = part.shape[0]
        q = _buf("q", (m, cfg.num_q_heads, cfg.head_dim), part.dtype, part.device)
        k = _buf("k", (m, cfg.num_kv_heads, cfg.head_dim), part.dtype, part.device)
        return q, k, k
    tl.rotary_embedding_and_store_kvcache_decode_from_splitk = fake_rope
sys.argv = [sys.argv[0]] + sys.argv[1:] + ["--no-cpu-baseline"]
bench.main()
