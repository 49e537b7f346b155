"""Host-side logic of the product, checked on CPU against the oracle's restatement of the reference:
batch planning, split-K heuristic, the host-mirrored block allocator, configs, weight loading, and
the "no GPU => loud failure" contract."""
import itertools
import json
import random
import types

import numpy as np
import os

import pytest
import torch

from oracle import eager_ops as ops
from oracle import synth
from oracle.ref_model import RefBlockManager
from swiftllm_amd import _hip
from swiftllm_amd.engine_config import EngineConfig
from swiftllm_amd.model_config import LlamaModelConfig
from swiftllm_amd.worker.batch_plan import plan_batch, select_seq_block_size
from swiftllm_amd.worker.block_manager import BlockAllocatorHost
from swiftllm_amd.worker.weight import load_weights


def _reference_plan(input_ids_list, decoding_seq_lens_list):
    """The arithmetic of reference model.py:268-297 and post_layer.py:24-29, with torch on CPU."""
    num_prefill = len(input_ids_list) - len(decoding_seq_lens_list)
    flat = list(itertools.chain(*input_ids_list))
    plens = [len(s) for s in input_ids_list[:num_prefill]]
    pl = torch.tensor(plens, dtype=torch.int32)
    starts = torch.cumsum(pl, 0, dtype=torch.int32) - pl
    dl = torch.tensor(decoding_seq_lens_list, dtype=torch.int32)
    pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in plens] + [dl - 1]) if flat else dl
    num_prefill_tokens = len(flat) - len(decoding_seq_lens_list)
    last = torch.cat((starts + pl - 1, torch.arange(num_prefill_tokens, len(flat), dtype=torch.int32)))
    return flat, plens, starts, pos, last, num_prefill_tokens


@pytest.mark.parametrize("seed", range(8))
def test_plan_batch_matches_reference_arithmetic(seed):
    rng = random.Random(seed)
    n_prefill, n_dec = rng.randint(0, 5), rng.randint(0, 6)
    if n_prefill + n_dec == 0:
        n_dec = 1
    prompts = [[rng.randrange(1000) for _ in range(rng.randint(1, 70))] for _ in range(n_prefill)]
    dec_lens = [rng.randint(1, 3000) for _ in range(n_dec)]
    ids = prompts + [[rng.randrange(1000)] for _ in range(n_dec)]
    seq_ids = rng.sample(range(100), n_prefill + n_dec)
    plan = plan_batch(ids, seq_ids, dec_lens, num_kv_heads=8)
    flat, plens, starts, pos, last, npt = _reference_plan(ids, dec_lens)
    assert plan.input_ids.tolist() == flat
    assert plan.seq_ids.tolist() == seq_ids
    assert plan.seq_lengths.tolist() == plens + dec_lens
    assert plan.prefill_seq_lens.tolist() == plens
    assert plan.prefill_start_locs_with_end[:-1].tolist() == starts.tolist()
    # the reference ends this array with num_tokens (model.py:340-343); the value every consumer
    # needs — and ours — is the number of PREFILL tokens (identical on pure-prefill batches)
    assert plan.prefill_start_locs_with_end[-1] == npt
    assert plan.position_indices.tolist() == pos.tolist()
    assert plan.last_token_indices.tolist() == last.tolist()
    assert plan.num_prefill_tokens == npt and plan.num_tokens == len(flat)
    assert plan.max_prefill_len == (max(plens) if plens else 0)
    assert plan.max_decoding_len == (max(dec_lens) if dec_lens else 0)
    assert plan.seq_block_size == select_seq_block_size(dec_lens, 8) and plan.seq_block_size % 16 == 0
    assert plan.num_seq_blocks == -(-plan.max_decoding_len // plan.seq_block_size)
    # packing round trip
    layout, total = plan.packed_layout()
    buf = np.zeros(total, dtype=np.int32)
    plan.pack_into(buf)
    for name, off, n in layout:
        assert off % 4 == 0
        assert buf[off:off + n].tolist() == getattr(plan, name).tolist()


def test_plan_batch_rejects_inconsistent_batches():
    with pytest.raises(ValueError):
        plan_batch([[1, 2]], [0, 1], [], 8)
    with pytest.raises(ValueError):
        plan_batch([[1, 2]], [0], [5], 8)       # a decoding sequence must bring exactly one token


@pytest.mark.parametrize("lens,kvh", [([1024], 8), ([1088] * 32, 8), ([16384] * 4, 32), ([1], 8),
                                      ([131072], 8), ([100, 5000, 70000], 2), ([1088] * 16, 8),
                                      ([1088] * 128, 8), ([], 8)])
def test_seq_block_size_properties(lens, kvh):
    """The split-K width is an internal choice (results are invariant to it up to fp32
    reassociation, checked on the GPU); what must hold: the reference kernel contract
    (`seq_block_size % block_size == 0`, paged_attn.py:167), a bounded split count, and the MI355X
    sizing rule: about one workgroup per CU, a single split when one round already covers the batch."""
    slots = 256
    got = select_seq_block_size(lens, kvh, slots)
    assert got % 64 == 0 and got >= 64
    if not lens:
        return
    assert -(-max(lens) // got) <= 128
    wgs = kvh * sum(-(-n // got) for n in lens)
    if kvh * len(lens) >= slots:
        assert got >= max(lens)                     # enough (sequence, kv-head) pairs: never split
    elif kvh * sum(lens) >= slots * 64 * 4:
        assert slots // 2 <= wgs <= 2 * slots       # otherwise: roughly one workgroup per CU
    # worked examples (BASELINE.json configs 1, 2, 3)
    if lens == [1024]:
        assert got == 64
    if lens == [1088] * 32:
        assert got == 1280 and wgs == 256
    if lens == [16384] * 4:
        assert got == 8192 and wgs == 256


def test_seq_block_size_is_stable_while_sequences_grow():
    # one value (=> one captured hipGraph) across the whole 1024-in / 128-out generation
    assert {select_seq_block_size([n] * 32, 8) for n in range(1025, 1153)} == {1280}


def _check_same(host: BlockAllocatorHost, ref: RefBlockManager):
    assert host.num_free_blocks == ref.num_free_blocks
    assert host.is_free.tolist() == ref.is_block_free.tolist()
    for s in range(ref.num_seq_allocated_blocks.numel()):
        n = int(ref.num_seq_allocated_blocks[s])
        assert host.num_allocated(s) == n
        assert host.seq_blocks.get(s, [])[:n] == ref.block_table[s, :n].tolist()


def test_block_allocator_replays_reference_trace(golden):
    g = golden("kvcache_blocks.pt")["block_manager_trace"]
    host = BlockAllocatorHost("GPU", g["num_blocks"], g["max_seqs"], g["mbps"], g["block_size"])
    for step in g["trace"]:
        if step["op"] == "alloc":
            _, picked = host.plan_allocation(step["ids"], step["lens"])
            assert picked.tolist() == step["ret"].tolist()
        else:
            freed = host.release(step["ids"])
            if step["op"] == "gather":
                assert freed == step["ret"].tolist()
        assert host.num_free_blocks == step["num_free"]
        assert host.is_free.tolist() == step["is_free"].tolist()
        for s in range(g["max_seqs"]):
            n = int(step["num_alloc"][s])
            assert host.seq_blocks.get(s, [])[:n] == step["block_table"][s, :n].tolist()


@pytest.mark.parametrize("seed", range(6))
def test_block_allocator_random_walk_equals_reference_manager(seed):
    rng = random.Random(seed)
    nb, ms, mb, bs = 64, 10, 16, 16
    host = BlockAllocatorHost("GPU", nb, ms, mb, bs)
    ref = RefBlockManager("GPU", nb, ms, mb, bs)
    lens = {}
    for _ in range(120):
        op = rng.choice(["alloc", "alloc", "grow", "free", "gather"])
        if op in ("alloc", "grow"):
            ids = rng.sample(range(ms), rng.randint(1, 4))
            targets = [max(lens.get(s, 0), min(mb * bs, lens.get(s, 0) + rng.randint(0, 40))) for s in ids]
            need = sum(-(-t // bs) - host.num_allocated(s) for s, t in zip(ids, targets))
            if need > host.num_free_blocks:
                with pytest.raises(RuntimeError):
                    host.plan_allocation(ids, targets)
                continue
            _, picked = host.plan_allocation(ids, targets)
            got = ref.allocate_blocks_for_seqs(torch.tensor(ids, dtype=torch.int32),
                                               torch.tensor(targets, dtype=torch.int32))
            assert picked.tolist() == got.tolist()      # lowest ids first, batch order
            lens.update(zip(ids, targets))
        else:
            ids = rng.sample(range(ms), rng.randint(1, 3))
            t = torch.tensor(ids, dtype=torch.int32)
            freed = host.release(ids)
            if op == "free":
                ref.free_blocks_for_seqs(t)
            else:
                assert freed == ref.gather_allocated_blocks_and_free(t).tolist()
            for s in ids:
                lens.pop(s, None)
        _check_same(host, ref)


def test_block_allocator_errors():
    host = BlockAllocatorHost("GPU", 4, 4, 2, 16)
    host.plan_allocation([0], [32])
    with pytest.raises(AssertionError):
        host.plan_allocation([0], [16])         # would have to shrink: the reference's logic error
    with pytest.raises(RuntimeError, match="No enough free blocks"):
        host.plan_allocation([1, 2], [32, 32])
    with pytest.raises(RuntimeError):
        host.plan_allocation([3], [48])         # > max_blocks_per_seq
    with pytest.raises(RuntimeError):
        host.plan_allocation([9], [1])          # sequence id outside the table


def test_engine_config_is_field_compatible_with_the_reference():
    # keyword construction exactly as examples/offline.py:22-33 does it
    ec = EngineConfig(model_path="/x", use_dummy=False, block_size=16, gpu_mem_utilization=0.99,
                      num_cpu_blocks=0, max_seqs_in_block_table=128, max_blocks_per_seq=2048,
                      max_batch_size=16, max_tokens_in_batch=2048 * 16)
    assert ec.dtype == "float16" and ec.fuse_qkv is True and ec.use_hip_graph is True
    import argparse
    p = argparse.ArgumentParser()
    EngineConfig.add_cli_args(p)
    args = p.parse_args(["--model-path", "/m"])
    ec2 = EngineConfig(**vars(args))            # reference api_server.py:97
    assert (ec2.block_size, ec2.gpu_mem_utilization, ec2.max_seqs_in_block_table,
            ec2.max_blocks_per_seq, ec2.max_batch_size, ec2.max_tokens_in_batch) == \
        (16, 0.97, 4096, 32768, 512, 32768)


def test_model_config_fields_and_kvslot_size():
    mc = LlamaModelConfig(synth.make_config(**synth.LLAMA3_8B))
    assert (mc.num_layers, mc.num_q_heads, mc.num_kv_heads, mc.head_dim) == (32, 32, 8, 128)
    assert mc.rope_scaling == 1.0 and mc.rope_theta == 500000.0
    assert mc.get_kvslot_size() == 131072               # 128 KiB per token
    assert mc.get_kvslot_size(torch.bfloat16) == 131072
    with pytest.raises(AssertionError):
        LlamaModelConfig(dict(synth.make_config(), model_type="gpt2"))


@pytest.mark.parametrize("fmt", ["safetensors", "safetensors_per_tensor", "bin"])
@pytest.mark.parametrize("fuse_qkv", [False, True])
def test_weight_loading_on_cpu(tmp_path, fmt, fuse_qkv):
    cfg = synth.make_config()
    sd = synth.make_state_dict(cfg, seed=3)
    synth.write_model_dir(str(tmp_path), cfg, sd, fmt=fmt.split("_")[0])
    mc = LlamaModelConfig.load_from_model_path(str(tmp_path))
    w = load_weights(mc, torch.float16, str(tmp_path), device="cpu", fuse_qkv=fuse_qkv,
                     streaming=fmt != "safetensors_per_tensor")
    assert torch.equal(w.wte, sd["model.embed_tokens.weight"])
    assert torch.equal(w.lm_head, sd["lm_head.weight"])
    l1 = w.layers[1]
    up, gate = sd["model.layers.1.mlp.up_proj.weight"], sd["model.layers.1.mlp.gate_proj.weight"]
    assert torch.equal(l1.up_gate_proj, torch.cat((up, gate)))      # [up ; gate], weight.py:133
    if fuse_qkv:
        assert l1.qkv_proj.shape == (128 + 2 * 64, 128) and not hasattr(l1, "q_proj")
        assert torch.equal(l1.qkv_proj, torch.cat([sd[f"model.layers.1.self_attn.{n}_proj.weight"] for n in "qkv"]))
    else:
        assert torch.equal(l1.k_proj, sd["model.layers.1.self_attn.k_proj.weight"])
    assert torch.equal(l1.down_proj, sd["model.layers.1.mlp.down_proj.weight"])
    assert torch.equal(w.final_norm, sd["model.norm.weight"])


def test_streaming_loader_converts_dtype_through_the_per_tensor_path_and_checks_shapes(tmp_path):
    cfg = synth.make_config()
    sd = synth.make_state_dict(cfg, seed=4)
    synth.write_model_dir(str(tmp_path), cfg, sd)
    mc = LlamaModelConfig.load_from_model_path(str(tmp_path))
    w = load_weights(mc, torch.bfloat16, str(tmp_path), device="cpu", fuse_qkv=True)      # fp16 file, bf16 run
    assert w.wte.dtype == torch.bfloat16 and torch.equal(w.wte, sd["model.embed_tokens.weight"].to(torch.bfloat16))
    bad = dict(sd)
    bad["model.layers.0.mlp.down_proj.weight"] = bad["model.layers.0.mlp.down_proj.weight"][:, :-1].contiguous()
    synth.write_model_dir(str(tmp_path), cfg, bad)
    with pytest.raises(AssertionError, match="does not match"):
        load_weights(mc, torch.float16, str(tmp_path), device="cpu")


def test_dummy_weights_and_tied_head(tmp_path):
    cfg = synth.make_config(rope_scaling=dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                              original_max_position_embeddings=128))
    synth.write_model_dir(str(tmp_path), cfg)
    mc = LlamaModelConfig.load_from_model_path(str(tmp_path))
    w = load_weights(mc, torch.bfloat16, str(tmp_path), use_dummy=True, device="cpu")
    assert w.model_version == "llama3.2" and w.lm_head is w.wte      # weight.py:157-163, 199-213
    assert w.layers[0].down_proj.dtype == torch.bfloat16
    assert float(w.layers[0].down_proj.abs().max()) <= 1e-3


def test_product_fails_loudly_without_a_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    cfg = synth.make_config()
    synth.write_model_dir(str(tmp_path), cfg, synth.make_state_dict(cfg))
    from swiftllm_amd import LlamaModel
    ec = EngineConfig(model_path=str(tmp_path), use_dummy=False, block_size=16, gpu_mem_utilization=0.9,
                      num_cpu_blocks=4, max_seqs_in_block_table=8, max_blocks_per_seq=8,
                      max_batch_size=4, max_tokens_in_batch=64)
    model = LlamaModel(ec)                      # reading config.json needs no device
    with pytest.raises(_hip.HipLibraryError):
        model.load_weights()
    from swiftllm_amd.worker.kernels import rmsnorm_inplace
    with pytest.raises(_hip.HipLibraryError):
        rmsnorm_inplace(torch.zeros(2, 8, dtype=torch.float16), torch.ones(8, dtype=torch.float16), 1e-5)


def test_product_never_imports_the_oracle():
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "swiftllm_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)


def test_swiftllm_alias_runs_modules_with_dash_m():
    """`python -m swiftllm.server.api_server` (the reference's launch line) goes through the alias."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "swiftllm.server.api_server", "--help"], cwd=root, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0 and "--model-path" in r.stdout, r.stderr[-500:]


def test_swiftllm_alias_resolves_to_this_implementation():
    import swiftllm
    import swiftllm_amd
    import swiftllm.worker.model as aliased
    import swiftllm_amd.worker.model as real
    from swiftllm.worker.kernels.paged_attn import paged_attention
    assert aliased is real and swiftllm.LlamaModel is swiftllm_amd.LlamaModel
    assert swiftllm.EngineConfig is swiftllm_amd.EngineConfig
    assert paged_attention.__module__ == "swiftllm_amd.worker.kernels.paged_attn"
    with pytest.raises(ImportError):
        import swiftllm.no_such_module  # noqa: F401


def test_streaming_loader_reads_sharded_checkpoints(tmp_path):
    """model.safetensors.index.json + several shard files (what HuggingFace writes for 8B+ models)."""
    import json
    from safetensors.torch import save_file
    cfg = synth.make_config()
    sd = synth.make_state_dict(cfg, seed=6)
    synth.write_model_dir(str(tmp_path), cfg)            # config.json only
    keys = sorted(sd)
    shards = {"model-00001-of-00003.safetensors": keys[0::3], "model-00002-of-00003.safetensors": keys[1::3],
              "model-00003-of-00003.safetensors": keys[2::3]}
    for name, ks in shards.items():
        save_file({k: sd[k].contiguous() for k in ks}, str(tmp_path / name))
    with open(tmp_path / "model.safetensors.index.json", "w", encoding="utf-8") as f:
        json.dump({"weight_map": {k: name for name, ks in shards.items() for k in ks}}, f)
    mc = LlamaModelConfig.load_from_model_path(str(tmp_path))
    for streaming in (True, False):
        w = load_weights(mc, torch.float16, str(tmp_path), device="cpu", fuse_qkv=True, streaming=streaming)
        assert torch.equal(w.wte, sd["model.embed_tokens.weight"])
        l0 = w.layers[0]
        assert torch.equal(l0.qkv_proj, torch.cat([sd[f"model.layers.0.self_attn.{n}_proj.weight"] for n in "qkv"]))
        assert torch.equal(l0.up_gate_proj, torch.cat((sd["model.layers.0.mlp.up_proj.weight"],
                                                       sd["model.layers.0.mlp.gate_proj.weight"])))
    (tmp_path / "model.safetensors.index.json").unlink()
    with pytest.raises(AssertionError, match="index.json not found"):
        load_weights(mc, torch.float16, str(tmp_path), device="cpu")


def test_engine_config_rejects_unsupported_block_size():
    """block_size is a compile-time tile of the HIP kernels (16); the reference takes it as a Triton constexpr
    (paged_attn.py:27, engine_config.py:37-42). Anything else must be refused on the host with a clear message."""
    from swiftllm_amd.engine_config import EngineConfig
    base = dict(model_path="", use_dummy=True, gpu_mem_utilization=0.5, num_cpu_blocks=4,
                max_seqs_in_block_table=8, max_blocks_per_seq=8, max_batch_size=4, max_tokens_in_batch=64)
    EngineConfig(block_size=16, **base)
    with pytest.raises(ValueError, match="block_size=32"):
        EngineConfig(block_size=32, **base)
    with pytest.raises(ValueError, match="dtype"):
        EngineConfig(block_size=16, dtype="float32", **base)


def test_graph_bucket_keeps_splits_balanced_and_few():
    """hipGraph replay quantises the flash-decoding split width so that a growing batch reuses its captured graph; the
    quantised width must stay within a few percent of the planner's (balanced splits: the longest one sets the kernel
    time), cover the longest sequence with the captured split count, and take few distinct values per octave."""
    import types
    from swiftllm_amd.worker.model import LlamaModel
    from swiftllm_amd.worker.batch_plan import select_seq_block_size
    widths = set()
    for batch, kvh, n0 in ((4, 32, 16372), (32, 8, 1024), (1, 8, 1024), (8, 8, 40000), (2, 32, 131000)):
        for step in range(0, 600, 7):
            lens = [n0 + step] * batch
            sbs = select_seq_block_size(lens, kvh, 256)
            plan = types.SimpleNamespace(seq_block_size=sbs, num_seq_blocks=-(-max(lens) // sbs), max_decoding_len=max(lens))
            q, cap = LlamaModel._graph_bucket(None, plan)
            widths.add((batch, kvh, n0, q, cap))
            assert cap * q >= max(lens) + 1
            if plan.num_seq_blocks > 1:
                assert sbs <= q <= sbs * 1.07 + 64, (lens[0], sbs, q)
                # the longest split of the replayed geometry is within 7 % of the planner's
                assert q <= 1.07 * sbs + 64
    assert len(widths) <= 5 * 4, len(widths)


def test_decode_batch_buckets_and_inert_rows():
    """hipGraph replay keys on the batch rounded up to a bucket (worker/model.py: _decode_batch_bucket) and fills the
    surplus rows with inert sequences: few distinct buckets over 1..256 (14 since r06b), never smaller than the batch, at
    most 7 / 15 / 31 rows of padding up to 32 / 64 / 256 sequences, thresholds of the launch paths (2 / 32 / 64 / 256) are bucket boundaries; the padded plan has the layout
    of an exact plan of the bucket size (a captured graph finds its metadata at fixed addresses), length 0 / position -1
    / token 0 in the inert rows, and the planner's split width does not see them."""
    import types
    from swiftllm_amd.worker.model import LlamaModel
    from swiftllm_amd.worker.batch_plan import plan_batch
    m = LlamaModel.__new__(LlamaModel)
    m.model_config = types.SimpleNamespace(num_kv_heads=8)
    m._num_slots = 256
    buckets = {b: m._decode_batch_bucket(b) for b in range(1, 257)}
    assert all(v >= b and v - b <= (7 if b <= 32 else 15 if b <= 64 else 31) for b, v in buckets.items())
    assert buckets[1] == 1 and buckets[2] == 2 and buckets[3] == 8 and buckets[32] == 32 and buckets[33] == 48
    assert buckets[64] == 64 and buckets[65] == 96 and buckets[129] == 160 and buckets[250] == 256 and buckets[256] == 256
    assert sorted(set(buckets.values())) == [1, 2, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256]
    for limit in (32, 64, 128, 256):     # a batch on one side of a launch-path threshold is never padded across it
        assert all(v <= limit for b, v in buckets.items() if b <= limit)
    lens = [1100, 37, 5]
    plan = m._plan_decode([7, 3, 9], lens, [[11], [12], [13]], True)
    exact = plan_batch([[0]] * 8, [0] * 8, [1] * 8, 8, 256)
    assert plan.batch_size == 8 and plan.real_seqs == 3 and plan.packed_layout() == exact.packed_layout()
    assert plan.input_ids.tolist() == [11, 12, 13, 0, 0, 0, 0, 0]
    assert plan.decoding_seq_lens.tolist() == lens + [0] * 5 and plan.position_indices.tolist() == [1099, 36, 4] + [-1] * 5
    assert plan.seq_ids.tolist()[:3] == [7, 3, 9] and plan.seq_lengths_list[:3] == lens
    unpadded = m._plan_decode([7, 3, 9], lens, [[11], [12], [13]], False)
    assert unpadded.batch_size == 3 and unpadded.real_seqs == 3
    assert (plan.seq_block_size, plan.num_seq_blocks, plan.max_decoding_len) == \
        (unpadded.seq_block_size, unpadded.num_seq_blocks, unpadded.max_decoding_len)
    assert plan_batch([[1]], [0], [5], 8).real_seqs == 1     # (a plan nobody padded)


def test_wide_kernel_routing_is_decided_once_per_deployment(tmp_path, monkeypatch):
    """kernels/route_tune.py: `decide` is a pure lookup — the r04 table for the classes it was measured on (Llama-3-8B widths,
    bfloat16), the deployment's table for any other, the library when nobody measured; the measurement happens in `prepare`
    (load time), is written next to the checkpoint, and a second replica READS it instead of measuring (so two replicas of
    one deployment route — and therefore round — alike); SWIFTLLM_ROUTE_TUNE=table / off pin the policy."""
    import torch
    from swiftllm_amd.worker.kernels import route_tune as R
    monkeypatch.delenv("SWIFTLLM_ROUTE_CACHE", raising=False)
    monkeypatch.delenv("SWIFTLLM_ROUTE_TUNE", raising=False)
    monkeypatch.setattr(R, "_table", {})
    monkeypatch.setattr(R, "_device_key_cache", {"cpu": "test-device|hip x|torch y"})
    bf16, f16 = torch.bfloat16, torch.float16
    assert R.decide(128, 4096, 14336, bf16, False) is True         # down: the table
    assert R.decide(224, 4096, 4096, bf16, False) is False         # o_proj at 224: library
    assert R.decide(192, 28672, 4096, bf16, True) is False         # SiLU-gate form > 128 tokens
    assert R.decide(100, 5120, 5120, bf16, False) is False         # nobody measured: the library, deterministically
    assert R.decide(100, 4096, 4096, f16, False) is False          # float16 at a measured (N, K) is not the measured class

    model_dir = tmp_path / "ckpt"
    model_dir.mkdir()
    calls = []

    def measure(n, k, silu, bucket):
        calls.append((n, k, silu, bucket))
        return bucket <= 128 and not silu
    classes = [(5120, 5120, False), (27648, 5120, True), (4096, 14336, False)]      # the last one: r04 table, never measured
    st = R.prepare(classes, bf16, "cpu", str(model_dir), measure)
    assert st["measured"] == 2 * len(R.BUCKETS) and st["read"] == 0
    assert st["path"] == str(model_dir / "swiftllm_amd_routes.json") and not os.path.exists(st["path"] + ".lock")
    assert sorted(set(c[:3] for c in calls)) == [(5120, 5120, False), (27648, 5120, True)]
    assert R.decide(100, 5120, 5120, bf16, False) is True and R.decide(128, 5120, 5120, bf16, False) is True
    assert R.decide(129, 5120, 5120, bf16, False) is False and R.decide(100, 27648, 5120, bf16, True) is False
    # a second replica (new process): reads the file, measures nothing, answers alike
    monkeypatch.setattr(R, "_table", {})
    boom = lambda *a: (_ for _ in ()).throw(AssertionError("the table must be read, not re-measured"))   # noqa: E731
    st2 = R.prepare(classes, bf16, "cpu", str(model_dir), boom)
    assert st2["read"] == 2 * len(R.BUCKETS) and st2["measured"] == 0
    assert R.decide(100, 5120, 5120, bf16, False) is True and R.decide(160, 5120, 5120, bf16, False) is False
    # a class the file does not cover yet is measured and merged into the same file; a refusing kernel = library
    st3 = R.prepare([(7168, 7168, False)], bf16, "cpu", str(model_dir), lambda *a: (_ for _ in ()).throw(RuntimeError("no")))
    assert st3["measured"] == len(R.BUCKETS) and R.decide(100, 7168, 7168, bf16, False) is False
    with open(st["path"], encoding="utf-8") as f:
        on_disk = json.load(f)["test-device|hip x|torch y"]
    assert len(on_disk) == 3 * len(R.BUCKETS)
    # a stale lock (the measurer died): wait, then measure ourselves
    monkeypatch.setattr(R, "_LOCK_WAIT_S", 0.3)
    other = tmp_path / "ckpt2"
    other.mkdir()
    (other / "swiftllm_amd_routes.json.lock").write_text("")
    st4 = R.prepare([(5120, 5120, False)], bf16, "cpu", str(other), lambda *a: True)
    assert st4["measured"] == len(R.BUCKETS)
    monkeypatch.setenv("SWIFTLLM_ROUTE_TUNE", "off")
    monkeypatch.setattr(R, "_table", {})
    assert R.prepare([(9216, 9216, False)], bf16, "cpu", str(model_dir), boom)["measured"] == 0
    assert R.decide(100, 9216, 9216, bf16, False) is False
    monkeypatch.setenv("SWIFTLLM_ROUTE_TUNE", "table")
    assert R.decide(100, 4096, 11008, f16, False) is True          # K >= 2N by the table


def test_tiny_batch_projection_policy():
    """kernels/linear.py: when may a projection consume the previous projection's slabs / the attention partials itself?
    (pure host logic: shapes, packed twin, K-chunk that fits LDS; the layer additionally limits the batch to 2)."""
    import torch
    import importlib
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")
    assert L._TINY_POLICY_M <= L._TINY_MAX_M == 4

    def weight(n, k, packed=True):
        w = torch.empty(n, k, dtype=torch.bfloat16)
        if packed:
            w._swl_packed = torch.empty(0)
        return w

    def partials(m, k, dtype=torch.bfloat16):
        return L.SplitKPartials(torch.empty(0), 8, m, k, dtype)

    wq, wug = weight(6144, 4096), weight(28672, 4096)
    assert L.tiny_from_splitk_ok(partials(1, 4096), wq) and L.tiny_from_splitk_ok(partials(4, 4096), wq)
    assert not L.tiny_from_splitk_ok(partials(5, 4096), wq)                        # more than 4 tokens
    assert not L.tiny_from_splitk_ok(partials(1, 4096), weight(6144, 4096, packed=False))
    assert not L.tiny_from_splitk_ok(partials(1, 4096, torch.float16), wq)         # dtype mismatch
    assert not L.tiny_from_splitk_ok(torch.empty(1, 4096), wq)                     # a tensor, not slabs
    assert L.tiny_from_splitk_ok(partials(2, 4096), wug, silu=True)
    assert not L.tiny_from_splitk_ok(partials(2, 8192), weight(28672, 8192), silu=True)   # whole K must fit LDS
    assert L.attn_partials_ok(1, 32, 128, weight(4096, 4096)) and not L.attn_partials_ok(5, 32, 128, weight(4096, 4096))
    assert not L.attn_partials_ok(1, 32, 128, weight(4096, 2048))                  # K != H * D
    assert not L.attn_partials_ok(1, 4, 32, weight(128, 128))                      # o_proj not split over K: nothing to fuse


def test_windowed_lowest_free_scan_equals_the_full_scan():
    """BlockAllocatorHost._lowest_free scans in windows (a 288 GB pool of a small model has tens of millions of blocks);
    it must hand out exactly the ids the reference's full scan does (torch.nonzero(is_block_free)[:n],
    block_manager.py:50) under random allocate / free traffic, holes and a nearly full pool included."""
    import numpy as np
    from swiftllm_amd.worker.block_manager import BlockAllocatorHost
    rng = np.random.default_rng(5)
    host = BlockAllocatorHost("GPU", 50_000, 64, 40_000, 16)
    lens = {}
    for step in range(300):
        sid = int(rng.integers(0, 64))
        if sid in lens and rng.random() < 0.4:
            host.release([sid])
            del lens[sid]
            continue
        grow = int(rng.choice([1, 16, 17, 500, 9000, 70_000]))
        want = lens.get(sid, 0) + grow
        need = -(-want // 16) - host.num_allocated(sid)
        if need > host.num_free_blocks or -(-want // 16) > 40_000:
            continue
        expect = np.flatnonzero(host.is_free)[:need]
        _, picked = host.plan_allocation([sid], [want])
        assert np.array_equal(picked, expect), step
        lens[sid] = want
    assert host.num_free_blocks == int(host.is_free.sum())


def test_engine_config_tuning_switches():
    """The internal decode-path switches are not constructor arguments: they default on, `tuning` overrides them, and a
    typo is an error instead of a silently ignored key."""
    kw = dict(model_path="/x", use_dummy=True, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0,
              max_seqs_in_block_table=8, max_blocks_per_seq=8, max_batch_size=2, max_tokens_in_batch=64)
    ec = EngineConfig(**kw)
    assert all(getattr(ec, k) is v for k, v in EngineConfig.TUNING_DEFAULTS.items())
    assert all(v is True for k, v in EngineConfig.TUNING_DEFAULTS.items() if k != "decode_engine")   # (opt-in: measured slower)
    assert EngineConfig(**kw, tuning=dict(decode_engine=True)).decode_engine is True
    ec = EngineConfig(**kw, tuning=dict(defer_rmsnorm=False))
    assert ec.defer_rmsnorm is False and ec.tiny_decode_batches is True
    with pytest.raises(ValueError, match="unknown tuning"):
        EngineConfig(**kw, tuning=dict(defer_rmsnrom=False))
    with pytest.raises(TypeError):
        EngineConfig(**kw, defer_rmsnorm=False)     # not a constructor argument


def test_decode_lookahead_is_taken_only_by_the_exact_continuation():
    """LlamaModel._take_lookahead (host logic only): the step prepared behind the running graph is used by the call that
    continues the same sequences with every length + 1 and exactly the tokens returned — and dropped by anything else
    (other tokens, other lengths, other sequences, a profile run, hipGraph off, look-ahead off, tokens not yet known)."""
    import types
    from swiftllm_amd.worker.model import LlamaModel, _DecodeLookahead

    def fresh(tokens=(7, 8, 9)):
        m = LlamaModel.__new__(LlamaModel)
        m.engine_config = types.SimpleNamespace(use_hip_graph=True)
        m._decode_lookahead = True
        la = _DecodeLookahead()
        la.seq_ids, la.lens, la.tokens, la.plan, la.dev = [4, 0, 2], [11, 21, 31], list(tokens) if tokens else None, "plan", "dev"
        m._lookahead = la
        return m, la
    ids = [[7], [8], [9]]
    m, la = fresh()
    assert m._take_lookahead(ids, [4, 0, 2], [11, 21, 31], False) is la and m._lookahead is None
    for args in (([[7], [8], [1]], [4, 0, 2], [11, 21, 31], False),        # another token
                 (ids, [4, 0, 2], [11, 21, 32], False),                    # another length
                 (ids, [4, 2, 0], [11, 21, 31], False),                    # other sequences / order
                 (ids[:2], [4, 0], [11, 21], False),                       # a smaller batch
                 ([[7], [8], [9, 9]], [4, 0, 2], [11, 21, 31], False),     # not one token each
                 (ids, [4, 0, 2], [11, 21, 31], True)):                    # a profile run (ignore_kvcache)
        m, _ = fresh()
        assert m._take_lookahead(*args) is None and m._lookahead is None, args
    m, _ = fresh()
    m.engine_config.use_hip_graph = False
    assert m._take_lookahead(ids, [4, 0, 2], [11, 21, 31], False) is None
    m, _ = fresh()
    m._decode_lookahead = False
    assert m._take_lookahead(ids, [4, 0, 2], [11, 21, 31], False) is None
    m, _ = fresh(tokens=None)
    assert m._take_lookahead(ids, [4, 0, 2], [11, 21, 31], False) is None


def test_blas_row_blocks_cover_every_row_once():
    """kernels/linear.py: prompt-sized products are taken in row blocks of the sizes hipBLASLt is good at (8192, 4096,
    remainder); here with toy block sizes on the CPU: the blocked product is the plain one (to fp32 summation order), for token counts around
    every block boundary, and small inputs are left to a single call."""
    import torch
    import torch.nn.functional as F
    import importlib
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")     # (the package re-exports the function)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(24, 16, generator=g)
    for m in (1, 4, 5, 8, 9, 12, 13, 16, 17, 20, 21, 29):
        a = torch.randn(m, 16, generator=g)
        got = L._blas_linear(a, w, row_blocks=(8, 4), plateau=16)
        assert got.shape == (m, 24)
        # (same products and sums per row; a BLAS may order a row's K-sum differently for another block height)
        torch.testing.assert_close(got, F.linear(a, w), rtol=1e-5, atol=1e-5)
    calls = []
    real = torch.mm
    try:
        torch.mm = lambda x, y, out=None: (calls.append(x.shape[0]), real(x, y, out=out))[1]
        L._blas_linear(torch.randn(15, 16, generator=g), w, row_blocks=(8, 4), plateau=16)
        assert calls == [8, 4, 3]
        calls.clear()
        L._blas_linear(torch.randn(29, 16, generator=g), w, row_blocks=(8, 4), plateau=16)
        assert calls == [28, 1]             # past the plateau: the whole multiple of the smallest block at once
        calls.clear()
        for m in (4, 16, 32):               # at or below the smallest block / on the plateau's grid: one plain call
            L._blas_linear(torch.randn(m, 16, generator=g), w, row_blocks=(8, 4), plateau=16)
        assert calls == []
    finally:
        torch.mm = real
    assert L._BLAS_ROW_BLOCKS == (8192, 4096) and L._BLAS_PLATEAU_ROWS == 16384


def test_block_planner_arithmetic_and_fixed_rule_mirror():
    """tools/blas_block_planner.py: `plan` finds the cheapest cover of M rows from a cost table (checked against brute
    force on a small grid, cliffs included), and `fixed_rule` is exactly the split kernels/linear.py ships."""
    import importlib
    import itertools
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    P = importlib.import_module("blas_block_planner")
    cost = {4: 10.0, 8: 14.0, 12: 40.0, 16: 26.0, 20: 60.0}         # a cliff at 12 and at 20
    for m in range(1, 41):
        pred, blocks = P.plan(cost, m, 4)
        assert sum(blocks) == m and all(b > 0 for b in blocks)
        units = -(-m // 4)
        brute = min(sum(cost[b] for b in combo)
                    for r in range(1, units + 1)
                    for combo in itertools.combinations_with_replacement(sorted(cost), r)
                    if sum(combo) == units * 4)
        assert abs(pred - brute) < 1e-9, (m, pred, brute)
    assert P.plan(cost, 12, 4) == (24.0, [8, 4])                     # steps around the cliff
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")
    w = torch.zeros(4, 8)
    real, calls = torch.mm, []
    try:
        torch.mm = lambda x, y, out=None: (calls.append(x.shape[0]), real(x, y, out=out))[1]
        for m in (1, 4096, 4097, 4124, 8191, 8192, 8193, 12288, 13000, 16383, 16384, 16385, 17000, 32768, 32796):
            calls.clear()
            L._blas_linear(torch.zeros(m, 8), w)
            assert (calls or [m]) == P.fixed_rule(m), m
    finally:
        torch.mm = real


def test_reference_harness_plans_keep_the_scripts_sequence_order():
    """oracle/ref_triton.py: planned_step — the reference-vs-itself control runs a step as several forward() calls and / or
    in reverse order; whatever the plan, tokens and logits must come back in the script's order, prefill sequences must
    precede decoding ones in every call and every decoding length must travel with its sequence."""
    import torch
    from oracle.ref_triton import planned_step

    class Fake:
        def __init__(self):
            self.calls = []

        def forward(self, ids, seq_ids, dec_lens):
            n_pre = len(ids) - len(dec_lens)
            assert all(len(x) > 1 for x in ids[:n_pre]) and all(len(x) == 1 for x in ids[n_pre:])
            self.calls.append((list(seq_ids), list(dec_lens)))
            for sid, dl in zip(seq_ids[n_pre:], dec_lens):
                assert dl == 100 + sid          # the length that belongs to this sequence
            log.append(torch.tensor([[float(s), float(x[-1])] for s, x in zip(seq_ids, ids)]))
            return [10 * s + x[-1] for s, x in zip(seq_ids, ids)]

    ids = [[1, 2, 3], [4, 5], [6], [7], [8]]
    seq_ids = [40, 41, 0, 1, 2]
    dec_lens = [100, 101, 102]
    want_t = [10 * s + x[-1] for s, x in zip(seq_ids, ids)]
    want_l = torch.tensor([[float(s), float(x[-1])] for s, x in zip(seq_ids, ids)])
    for split, reverse in ((1, False), (2, False), (1, True), (2, True), (3, False)):
        log, fake = [], Fake()
        t, lg = planned_step(fake, log, ids, seq_ids, dec_lens, split, reverse)
        assert t == want_t and torch.equal(lg, want_l), (split, reverse)
        assert len(fake.calls) == min(split, 3)
        assert sorted(s for c in fake.calls for s in c[0]) == sorted(seq_ids)


def test_affinity_plan_says_which_rule_it_applied(monkeypatch):
    """dp.affinity_plan (r04): eight replicas never share cores — NUMA-local cores when the KFD topology is
    readable, an even contiguous split of the allowed cores when it is not — and the reason travels with the core list
    (bench.py prints it as config.cpu_affinity)."""
    from swiftllm_amd import dp
    allowed = list(range(64))
    monkeypatch.setattr(dp, "_gpu_numa_nodes", lambda: [])
    seen = []
    for r in range(8):
        cpus, how = dp.affinity_plan(r, 8, allowed)
        assert len(cpus) == 8 and "even split" in how and "unreadable" in how
        seen += cpus
    assert sorted(seen) == allowed
    monkeypatch.setattr(dp, "_gpu_numa_nodes", lambda: [-1] * 8)
    assert "reported as" in dp.affinity_plan(3, 8, allowed)[1]
    cpus, how = dp.affinity_plan(0, 1, allowed)
    assert cpus == allowed and "single replica" in how


def test_spawn_local_ranks_kills_a_rank_that_ignores_sigterm(monkeypatch):
    """a surviving rank stuck in a HIP call (here: ignoring SIGTERM) must not make the launcher spin for ever —
    terminate, a grace period, then kill; the timeout is a monotonic deadline."""
    import sys
    import time
    from swiftllm_amd import dp
    monkeypatch.setattr(dp, "TERMINATE_GRACE_S", 0.5)
    stubborn = ("import os, signal, sys, time\n"
                "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                "sys.exit(3) if os.environ['RANK'] == '0' else time.sleep(60)\n")
    t0 = time.monotonic()
    rc = dp.spawn_local_ranks([sys.executable, "-c", stubborn], 2, visible_devices=["0", "0"])
    assert rc == 3 and time.monotonic() - t0 < 20
    sleeper = "import signal, time\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\ntime.sleep(60)\n"
    t0 = time.monotonic()
    rc = dp.spawn_local_ranks([sys.executable, "-c", sleeper], 2, timeout_s=1.0, visible_devices=["0", "0"])
    assert rc == 124 and time.monotonic() - t0 < 20


def test_wide_gemm_routing_is_the_measured_table():
    """kernels/linear.py: _wide_wins / _wide_silu_wins encode profiles/r04c_/r04d_gemm_wide_micro.jsonl and, for qkv / o, the
    r06d re-measurement profiles/r06d_gemm_wide_routing_remeasure.jsonl (Llama-3-8B widths)."""
    from swiftllm_amd.worker.kernels.linear import _wide_wins, _wide_silu_wins
    qkv, o, up_gate, down, lm_head = (6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128256, 4096)
    for m in (65, 96, 128, 160, 192, 224, 256):
        assert _wide_wins(m, *down)
        assert not _wide_wins(m, *up_gate) and not _wide_wins(m, *lm_head)
    assert [_wide_wins(m, *qkv) for m in (96, 128, 160, 192, 200, 224, 256)] == [True, True, True, True, True, True, True]
    assert [_wide_wins(m, *o) for m in (96, 128, 160, 192, 200, 224, 256)] == [True, True, True, True, False, False, False]
    assert [_wide_silu_wins(m) for m in (96, 128, 129, 256)] == [True, True, False, False]
    # a threshold inside a replay bucket would route a batch differently in eager launches and in its padded graph
    from swiftllm_amd.worker.model import LlamaModel
    for shape in (qkv, o, up_gate, down, lm_head, (8192, 2048), (3072, 2048)):
        for m in range(65, 257):
            assert _wide_wins(m, *shape) == _wide_wins(LlamaModel._decode_batch_bucket(None, m), *shape), (m, shape)
            assert _wide_silu_wins(m) == _wide_silu_wins(LlamaModel._decode_batch_bucket(None, m)), m


def test_bench_prefill_flops_and_full_depth_cpu_baseline():
    """bench.py: the flop count behind `prefill_roofline` (SURVEY.md §8d arithmetic: projections on every token, causal
    attention, lm_head on the last tokens) and the full-depth CPU baseline leg (all layers timed, nothing extrapolated)."""
    import bench
    cfg = bench.model_config_dict("llama3-8b")
    gemm, attn = bench.prefill_flops(cfg, [1024] * 32)
    L, h, I, V, kvd = 32, 4096, 14336, 128256, 1024
    assert gemm == 2 * L * (2 * h * h + 2 * kvd * h + 3 * I * h) * 32768 + 2 * 32 * V * h
    assert attn == L * 32 * 2 * 1024 * 1024 * 128 * 32
    assert gemm + attn == 466226551717888                      # the figure in profiles/r04*_bench_*.json
    tiny = bench.model_config_dict("tiny")
    out = bench.cpu_baseline(tiny, 2, 40, "bfloat16", steps=1)
    assert out["kind"] == "port" and out["extrapolated"] is False and out["value"] > 0
    assert "2 layers, all timed" in out["sample"]


def test_decode_engine_stream_packing_matches_the_kernel_addressing():
    """worker/decode_engine.py packs what csrc/decode_engine.hip reads: per CU the slots of its rows in (row group of 8,
    k-chunk of 1024) order; in a slot, 1 KiB piece p, lane l, element e = W[row 8g + l // 8][1024 j + 64 p + 8 (l % 8) + e];
    a layer is qkv | o | up rows | the matching gate rows | down, and its slot count is the library's."""
    from swiftllm_amd.worker import decode_engine as de
    n, k = 2 * 2048, 2 * 1024
    w = torch.arange(n * k, dtype=torch.int32).reshape(n, k)
    packed = de._pack_rows(w)
    r, kj = n // 256, k // 1024
    assert packed.shape == (256, (r // 8) * kj * 8192)
    rng = random.Random(0)
    for _ in range(2000):
        cu, g, j, p, lane, e = (rng.randrange(256), rng.randrange(r // 8), rng.randrange(kj), rng.randrange(16),
                                rng.randrange(64), rng.randrange(8))
        slot = g * kj + j
        got = int(packed[cu, slot * 8192 + p * 512 + lane * 8 + e])
        assert got == int(w[cu * r + 8 * g + lane // 8, 1024 * j + 64 * p + 8 * (lane % 8) + e])
    # layer layout and slot count (Llama-3-8B geometry scaled down to what fits a CPU test: hidden 2048, 16/8 heads, FFN 2048)
    hidden, heads, kv_heads, ffn = 2048, 16, 8, 2048
    lib = _hip.load()
    spl = lib.swl_decode_engine_slots_per_layer(hidden, heads, kv_heads, ffn)
    qkv = torch.zeros(((heads + 2 * kv_heads) * 128, hidden), dtype=torch.int16) + 1
    o = torch.zeros((hidden, hidden), dtype=torch.int16) + 2
    up_gate = torch.cat((torch.zeros((ffn, hidden), dtype=torch.int16) + 3, torch.zeros((ffn, hidden), dtype=torch.int16) + 4))
    down = torch.zeros((hidden, ffn), dtype=torch.int16) + 5
    layer = de.pack_engine_layer(qkv, o, up_gate, down)
    assert layer.shape == (256, spl * 8192)
    counts = [int((layer[7] == v).sum()) // 8192 for v in (1, 2, 3, 4, 5)]
    assert counts == [2 * 2, 1 * 2, 1 * 2, 1 * 2, 1 * 2] and sum(counts) == spl
    order = layer[7, ::8192].tolist()
    assert order == sorted(order)                      # qkv | o | up | gate | down
    assert lib.swl_decode_engine_supported(4096, 32, 8, 128, 14336, 256) == 1      # Llama-3-8B on an MI355X
    assert lib.swl_decode_engine_slots_per_layer(4096, 32, 8, 14336) == 104        # 436 MB per layer / 256 CUs / 16 KiB
    assert lib.swl_decode_engine_supported(4096, 32, 32, 128, 11008, 256) == 0     # Llama-2-7B: FFN rows do not divide
    assert lib.swl_decode_engine_supported(4096, 32, 8, 128, 14336, 304) == 0


def test_gemm_wide_ablation_tool_still_matches_the_kernel_source():
    """tools/make_gemm_wide_ablations.py edits csrc/gemm_wide.hip by exact text: every pattern must occur exactly once in the
    shipped source, or the timing-only variants of DESIGN.md section 4.7 can no longer be rebuilt."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_gemm_wide_ablations", os.path.join(root, "tools", "make_gemm_wide_ablations.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    src = open(tool.SRC).read()
    for name, edits in tool.EDITS.items():
        for old, _new in edits:
            assert src.count(old) == 1, (name, old[:60])
