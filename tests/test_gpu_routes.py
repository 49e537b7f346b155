"""Kernel routing is decided ONCE per deployment and identically for every replica.

Two replica processes start AT THE SAME TIME on the one GPU of the box against one checkpoint whose projection widths are in
no measured table (hidden 5120 = Llama-2-13B's width: 40 q / 10 kv heads of 128, FFN 13824, one layer): `load_weights()` of
the first one to take the lock file measures every (shape, 32-token bucket) class of 65..256 tokens and writes
`<model_path>/swiftllm_amd_routes.json`; the other one waits and READS it. Both then decode the same batch of 96 sequences:
the logits must be BIT-identical (the two candidate kernels sum K in different orders, so replicas that routed differently
would not be), and nothing may be measured inside a forward (`route_tune.decide` is a pure lookup).
Reference semantics of the routed operator: swiftllm/worker/kernels/linear.py:3-12 (`F.linear`).
"""
import json
import os
import subprocess
import sys

import pytest

from oracle import synth

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys, torch
from swiftllm_amd import EngineConfig, LlamaModel
from swiftllm_amd.worker.kernels import route_tune
path, batch = sys.argv[1], int(sys.argv[2])
m = LlamaModel(EngineConfig(model_path=path, use_dummy=False, block_size=16, gpu_mem_utilization=0.45, num_cpu_blocks=0,
                            max_seqs_in_block_table=batch + 8, max_blocks_per_seq=8, max_batch_size=batch,
                            max_tokens_in_batch=batch * 24, dtype="bfloat16", use_hip_graph=False))
m.load_weights()
table = dict(route_tune._table)
m.init_kvcache_and_swap(batch * 2 + 8)
g = torch.Generator().manual_seed(4)
prompts = [torch.randint(0, 512, (5 + i % 7,), generator=g).tolist() for i in range(batch)]
ids = list(range(batch))
m.post_layer.logits_tap = []
toks = m.forward(prompts, ids, [])
def boom(*a, **k): raise AssertionError("a forward must never time kernels")
route_tune.time_us = boom
toks = m.forward([[t] for t in toks], ids, [len(p) + 1 for p in prompts])
lg = m.post_layer.logits_tap[-1].float().cpu().contiguous()
print(json.dumps(dict(sha=hashlib.sha256(lg.numpy().tobytes()).hexdigest(), tokens=toks, table=table)))
"""


def test_two_replicas_of_one_deployment_route_alike_and_return_identical_logits(tmp_path):
    cfg = synth.make_config(num_hidden_layers=1, hidden_size=5120, num_attention_heads=40, num_key_value_heads=10,
                            intermediate_size=13824, vocab_size=512, max_position_embeddings=512)
    import torch
    sd = synth.make_state_dict(cfg, seed=21, dtype=torch.bfloat16)
    synth.write_model_dir(str(tmp_path), cfg, sd)
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SWIFTLLM_ROUTE_TUNE", None)
    env.pop("SWIFTLLM_ROUTE_CACHE", None)
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, str(tmp_path), "96"], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append((json.loads(out.strip().splitlines()[-1]), err))
    a, b = outs[0][0], outs[1][0]
    assert a["table"] == b["table"] and len(a["table"]) >= 4 * 6       # qkv, o, up/gate (+ SiLU form), down x 6 buckets
    assert a["sha"] == b["sha"] and a["tokens"] == b["tokens"]
    path = tmp_path / "swiftllm_amd_routes.json"
    assert path.exists() and not (tmp_path / "swiftllm_amd_routes.json.lock").exists()
    with open(path, encoding="utf-8") as f:
        disk = json.load(f)
    (dev_table,) = disk.values()
    assert {k: bool(v) for k, v in dev_table.items()} == a["table"]
    measured = sum("classes measured" in e and " 0 (shape" not in e for _, e in outs)
    print("\n[routes] replicas that measured:", measured, "table entries:", len(a["table"]),
          "hand-written kernel chosen for:", sorted(k for k, v in a["table"].items() if v))
