#!/usr/bin/env python3
"""swap_micro.py — KV block swap throughput (§8f rank 4): LlamaModel.swap_out_seqs / swap_in_seqs of `seqs`
sequences of `len` tokens on a Llama-3-8B-shaped pool (1 MiB per block and pool), pinned vs pageable swap pool."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--seqs", type=int, default=16)
ap.add_argument("--len", type=int, default=2048)
a = ap.parse_args()
from swiftllm_amd import EngineConfig, LlamaModel
import json as _json, tempfile
cfg = bench.model_config_dict("llama3-8b")
cfg["num_hidden_layers"] = 32
blocks = a.seqs * (a.len // 16) + 8
for pinned in (True, False):
    path = tempfile.mkdtemp(prefix="swl_swap_")
    with open(os.path.join(path, "config.json"), "w") as f:
        _json.dump(cfg, f)
    ec = EngineConfig(model_path=path, use_dummy=True, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=blocks,
                      max_seqs_in_block_table=64, max_blocks_per_seq=a.len // 16 + 8, max_batch_size=a.seqs,
                      max_tokens_in_batch=4096, dtype="bfloat16", tuning=dict(pin_swap_memory=pinned))
    model = LlamaModel(ec)
    model.init_kvcache_and_swap(blocks)      # no weights needed: only the pools and the block managers
    ids = list(range(a.seqs))
    model.gpu_block_manager.allocate_blocks_for_seqs(ids, [a.len] * a.seqs)
    nbytes = 2 * a.seqs * (a.len // 16) * model.k_cache[0].numel() * model.k_cache.element_size()
    res = {"pinned": pinned, "GB": round(nbytes / 1e9, 3)}
    for name, fn in (("swap_out", model.swap_out_seqs), ("swap_in", model.swap_in_seqs), ("swap_out2", model.swap_out_seqs),
                     ("swap_in2", model.swap_in_seqs)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(ids)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[name + "_GBps"] = round(nbytes / dt / 1e9, 2)
    print(json.dumps(res), flush=True)
    del model
    torch.cuda.empty_cache()
