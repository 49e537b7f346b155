#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): Llama-3-8B, bf16, random-init weights, batch 32 per GPU,
1024-token synthetic prompts, 128 generated tokens. One "step" = one decode forward of the whole
batch (`LlamaModel.forward`, the hot path) with everything resident in HBM. The prompt phase (one
32x1024-token prefill forward) runs before the timed region and is reported separately as
`prefill_tok_s`. `value` = decode tokens/s of the whole job = n_gpus * batch * K / max-over-ranks
time of the K timed steps (request-sharded replicas, weak scaling, no collective on the data path).

Extra objects on the JSON line:
  roofline      the dominant hand-written kernel of the decode step (paged-attention phase 1):
                algorithmic bytes per launch / its mean launch duration, measured live with HIP events
                on the launch stream over launches that cycle through all layers' KV (4.6 GB, far
                past the 256 MiB Infinity Cache), at the decode loop's exact shapes and geometry.
  step_roofline the whole decode step against HBM: (weights + KV read + KV write) / step time.
  cpu_baseline  rank 0, N=1 only: the CPU oracle's forward (oracle/ref_model.py — the reference has
                no CPU path of its own) on the host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU")
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama2-7b", "tiny"])
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"])
    ap.add_argument("--no-hip-graph", action="store_true")
    ap.add_argument("--no-fuse-qkv", dest="fuse_qkv", action="store_false")
    ap.add_argument("--no-skinny-gemm", dest="skinny_gemm", action="store_false")
    ap.add_argument("--no-splitk-fusion", dest="splitk_fusion", action="store_false")
    ap.add_argument("--layer-fusion", dest="layer_fusion", action="store_true",
                    help="fold norm/rotary/residual hand-offs into the GEMMs (experimental, slower: DESIGN.md §4.4)")
    ap.add_argument("--no-packed-weights", dest="packed_weights", action="store_false")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=256)
    return ap.parse_args()


MODEL_DIMS = {
    "llama3-8b": dict(num_hidden_layers=32, hidden_size=4096, num_attention_heads=32, num_key_value_heads=8,
                      intermediate_size=14336, vocab_size=128256, max_position_embeddings=8192,
                      rope_theta=500000.0),
    "llama2-7b": dict(num_hidden_layers=32, hidden_size=4096, num_attention_heads=32, num_key_value_heads=32,
                      intermediate_size=11008, vocab_size=32000, max_position_embeddings=4096,
                      rope_theta=10000.0),
    "tiny": dict(num_hidden_layers=2, hidden_size=512, num_attention_heads=4, num_key_value_heads=1,
                 intermediate_size=1024, vocab_size=512, max_position_embeddings=2048, rope_theta=10000.0),
}


def model_config_dict(name):
    cfg = dict(model_type="llama", hidden_act="silu", rms_norm_eps=1e-5, rope_scaling=None,
               tie_word_embeddings=False)
    cfg.update(MODEL_DIMS[name])
    return cfg


def ensure_positions(cfg, needed):
    """Long-context runs on a short-context architecture (BASELINE configs[3]: Llama-2-7B at 16k): linear
    rope scaling so the rotary table covers the run, as long-context Llama-2 derivatives do."""
    have = cfg["max_position_embeddings"]
    if needed + 128 > have:
        cfg["rope_scaling"] = float(-(-(needed + 128) // have))
    return cfg


def build_model(args, cfg, num_blocks):
    from swiftllm_amd import EngineConfig, LlamaModel
    path = tempfile.mkdtemp(prefix="swl_bench_")
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f)
    ec = EngineConfig(model_path=path, use_dummy=True, block_size=16, gpu_mem_utilization=0.97,
                      num_cpu_blocks=0, max_seqs_in_block_table=max(64, args.batch),
                      max_blocks_per_seq=max(256, (args.prompt_len + args.steps + args.warmup) // 16 + 8),
                      max_batch_size=args.batch, max_tokens_in_batch=args.batch * args.prompt_len,
                      dtype=args.dtype, fuse_qkv=args.fuse_qkv, use_hip_graph=not args.no_hip_graph,
                      use_skinny_gemm=args.skinny_gemm, fuse_splitk_consumers=args.splitk_fusion,
                      fuse_decode_layer=getattr(args, "layer_fusion", False),
                      pack_decode_weights=getattr(args, "packed_weights", True))
    model = LlamaModel(ec)
    model.load_weights()
    # random-init weights of the named architecture: N(0, 0.02^2) matrices, norm weights 1 + N(0, 0.02^2)
    # (the reference's dummy U(-1e-3, 1e-3) makes every logit ~0; bench on realistic value ranges)
    g = torch.Generator(device="cuda").manual_seed(1234 + int(os.environ.get("RANK", "0")))
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
    model.repack_decode_weights()       # the packed decode copies follow the re-initialised weights
    model.init_kvcache_and_swap(num_blocks)
    return model


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def weight_bytes(cfg, e):
    L, h, I, V = cfg["num_hidden_layers"], cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    kvd = cfg["num_key_value_heads"] * (h // cfg["num_attention_heads"])
    # W = e*(L*(2h^2 + 2*KVH*D*h + 3*I*h + 2h) + V*h + h); the embedding is gathered
    return e * (L * (2 * h * h + 2 * kvd * h + 3 * I * h + 2 * h) + V * h + h)


def kernel_roofline(model, lens, iters):
    """Mean duration of paged-attention phase 1 (the dominant hand-written kernel of a decode step)
    measured with HIP events on the launch stream, at the decode loop's shapes and launch geometry."""
    from swiftllm_amd import _hip
    from swiftllm_amd.worker.batch_plan import plan_batch
    mc, ecfg = model.model_config, model.engine_config
    B, H, KVH, D, L = len(lens), mc.num_q_heads, mc.num_kv_heads, mc.head_dim, mc.num_layers
    plan = plan_batch([[0]] * B, list(range(B)), lens, KVH, model._num_slots)
    sbs, nsb = model._graph_bucket(plan) if ecfg.use_hip_graph else (plan.seq_block_size, plan.num_seq_blocks)
    dev = model.device
    q = torch.randn(B, H, D, device=dev, dtype=torch.float32).to(model.dtype)
    o = torch.empty_like(q)
    seq_ids = torch.arange(B, dtype=torch.int32, device=dev)
    d_lens = torch.tensor(lens, dtype=torch.int32, device=dev)
    mid_o = torch.empty(B * H * nsb * D, dtype=torch.float32, device=dev)
    mid_lse = torch.empty(B * H * nsb, dtype=torch.float32, device=dev)
    bt = model.gpu_block_manager.block_table
    code, scale = _hip.dtype_code(model.dtype), D ** -0.5

    def launch(layer):
        _hip.call("swl_paged_attn_phase1", o.data_ptr(), q.data_ptr(), model.k_cache.data_ptr(),
                  model.v_cache.data_ptr(), bt.data_ptr(), seq_ids.data_ptr(), d_lens.data_ptr(),
                  mid_o.data_ptr(), mid_lse.data_ptr(), scale, B, H, KVH, D, L, ecfg.block_size, layer,
                  bt.shape[1], sbs, nsb, H * D, H * D, code, _hip.stream())

    for i in range(min(iters, 2 * L)):
        launch(i % L)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for i in range(iters):
        launch(i % L)
    stop.record()
    stop.synchronize()
    us = start.elapsed_time(stop) * 1e3 / iters
    e = model.dtype.itemsize
    kv_bytes = sum(lens) * 2 * KVH * D * e
    splits = sum(-(-n // sbs) for n in lens)
    part_bytes = (splits * H * (D + 1) * 4) if nsb > 1 else B * H * D * e
    alg_bytes = kv_bytes + B * H * D * e + part_bytes
    gbs = alg_bytes / (us * 1e-6) / 1e9
    # HBM traffic cannot be read from inside this process: it comes from the committed rocprofv3 PMC
    # passes on this kernel (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/r01_paged_attn_pmc.*),
    # scaled by bytes to this launch; null for a kernel specialisation that was not profiled.
    traffic, src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r01_paged_attn_pmc.json")
    if os.path.exists(pmc_path) and (H, KVH, D) == (32, 8, 128) and nsb == 1:
        with open(pmc_path, encoding="utf-8") as f:
            pmc = json.load(f)
        traffic = int(alg_bytes * pmc["traffic_over_algorithmic"])
        src = "profiles/r01_paged_attn_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, scaled by bytes)"
    return dict(bound="hbm", kernel="paged_attn_phase1_kernel", achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), frac_of_measured_copy=round(gbs / HBM_COPY_GBS, 4),
                traffic=traffic, traffic_source=src, bytes_per_launch=int(alg_bytes), us_per_launch=round(us, 2),
                seq_block_size=sbs, num_seq_blocks=nsb, launches=iters)


def gemm_roofline(model, batch, iters):
    """Mean duration of the up/gate projection + SiLU-gate (the kernel with the largest share of a decode step:
    54 % of the weight bytes), HIP events on the launch stream, cycling through all layers' weights
    (7.5 GB footprint >> the 256 MiB Infinity Cache). None when the decode path does not use it."""
    from swiftllm_amd import _hip
    mc, ecfg = model.model_config, model.engine_config
    M, K, I = batch, mc.hidden_size, mc.ffn_inter_dim
    if not getattr(ecfg, "use_skinny_gemm", False) or M > 32 or I % 32 or K % 128:
        return None
    layers = model.weight.layers
    x = torch.randn(M, K, device=model.device, dtype=torch.float32).to(model.dtype)
    out = torch.empty(M, I, device=model.device, dtype=model.dtype)
    code = _hip.dtype_code(model.dtype)

    packed = all(getattr(l.up_gate_proj, "_swl_packed", None) is not None for l in layers)
    fn = "swl_gemm_skinny_packed_silu_gate" if packed else "swl_gemm_skinny_silu_gate"
    srcs = [(l.up_gate_proj._swl_packed if packed else l.up_gate_proj) for l in layers]

    def launch(i):
        _hip.call(fn, out.data_ptr(), x.data_ptr(), srcs[i % len(srcs)].data_ptr(), M, I, K, K, I, code, _hip.stream())
    for i in range(min(iters, len(layers))):
        launch(i)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for i in range(iters):
        launch(i)
    stop.record()
    stop.synchronize()
    us = start.elapsed_time(stop) * 1e3 / iters
    e = model.dtype.itemsize
    alg_bytes = 2 * I * K * e + M * K * e + M * I * e
    gbs = alg_bytes / (us * 1e-6) / 1e9
    traffic, src = None, None
    pmc_name = "r01f_gemm_silu_packed_pmc.json" if packed else "r01e_gemm_silu_pmc.json"
    pmc_path = os.path.join(ROOT, "profiles", pmc_name)
    if os.path.exists(pmc_path) and (I, K) == (14336, 4096) and model.dtype == torch.bfloat16:
        with open(pmc_path, encoding="utf-8") as f:
            pmc = json.load(f)
        traffic = int(alg_bytes * pmc["traffic_over_algorithmic"])
        src = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, scaled by bytes)"
    return dict(bound="hbm", kernel="gemm_skinny_ring_kernel<SiluGate%s> (up/gate projection + SiLU-gate)" % (", packed W" if packed else ""),
                achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                frac_of_measured_copy=round(gbs / HBM_COPY_GBS, 4), traffic=traffic, traffic_source=src,
                bytes_per_launch=int(alg_bytes), us_per_launch=round(us, 2), launches=iters)


def cpu_baseline(cfg, batch, context, dtype):
    """The CPU oracle (a port: the reference has no CPU forward) on the host cores: same architecture
    and batch, truncated to 2 of the layers, short context, a couple of decode steps (~10-30 s incl.
    building the random weights); per-layer and head costs are timed separately and recombined for
    the full depth. This is the ONLY part of bench.py that touches oracle/."""
    from oracle import eager_ops, synth
    from oracle.ref_model import RefLlamaModel
    from swiftllm_amd import EngineConfig, LlamaModelConfig
    small = dict(cfg, num_hidden_layers=2)
    tdtype = torch.bfloat16 if dtype == "bfloat16" else torch.float16
    sd = synth.make_state_dict(small, seed=0, dtype=tdtype)
    ec = EngineConfig(model_path="", use_dummy=True, block_size=16, gpu_mem_utilization=0.9,
                      num_cpu_blocks=0, max_seqs_in_block_table=batch, max_blocks_per_seq=context // 16 + 4,
                      max_batch_size=batch, max_tokens_in_batch=batch * context)
    eager_ops.linear = lambda a, w: torch.nn.functional.linear(a, w)    # native 16-bit CPU GEMM
    ref = RefLlamaModel(LlamaModelConfig(small), ec, sd, tdtype)
    ref.init_kvcache_and_swap(batch * (context // 16 + 2))
    layer_s = [0.0]
    orig_layer = ref._layer

    def timed_layer(*a):
        t0 = time.perf_counter()
        r = orig_layer(*a)
        layer_s[0] += time.perf_counter() - t0
        return r
    ref._layer = timed_layer
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(0, cfg["vocab_size"], (context,), generator=g).tolist() for _ in range(batch)]
    toks = ref.forward(prompts, list(range(batch)), [])
    steps, total, layers = 2, 0.0, 0.0
    for s in range(steps):
        layer_s[0] = 0.0
        t0 = time.perf_counter()
        toks = ref.forward([[t] for t in toks], list(range(batch)), [context + 1 + s] * batch)
        total += time.perf_counter() - t0
        layers += layer_s[0]
    per_layer = layers / steps / 2
    rest = (total - layers) / steps
    full_step = per_layer * cfg["num_hidden_layers"] + rest
    return dict(value=round(batch / full_step, 3), unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=(f"oracle/ref_model.py decode step, batch {batch}, context {context}, 2 of "
                        f"{cfg['num_hidden_layers']} layers timed ({per_layer * 1e3:.1f} ms/layer) + embedding/"
                        f"lm_head ({rest * 1e3:.1f} ms), recombined for {cfg['num_hidden_layers']} layers; "
                        f"{steps} steps, native 16-bit CPU GEMM"))


def main():
    args = parse_args()
    from swiftllm_amd import dp
    rank, local_rank, world = dp.env_rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    dp.init_control_group()

    cfg = model_config_dict(args.model)
    B, S = args.batch, args.prompt_len
    gen_total = args.warmup + args.steps
    ensure_positions(cfg, S + gen_total)
    blocks_per_seq = (S + gen_total + 1 + 15) // 16
    num_blocks = int(B * blocks_per_seq * 1.25) + 8
    model = build_model(args, cfg, num_blocks)
    e = model.dtype.itemsize

    # every rank serves its own shard of the requests: `batch` sequences per GPU
    g = torch.Generator().manual_seed(1 + rank)
    prompts = [torch.randint(0, cfg["vocab_size"], (S,), generator=g).tolist() for _ in range(B)]
    seq_ids = list(range(B))

    # ---- prompt phase (reported, outside the K timed steps) ------------------------------------------------
    model.forward(prompts, seq_ids, [])                 # untimed: GEMM heuristics, allocator pools
    model.free_seqs_resources(seq_ids)
    dp.barrier()
    toks, prefill_s = timed(lambda: model.forward(prompts, seq_ids, []))
    prefill_units, prefill_max_s = dp.reduce_job(B * S, prefill_s)

    # ---- decode: W warm-up steps, then exactly K timed steps --------------------------------------------------
    lens = [S] * B

    def step():
        nonlocal toks, lens
        lens = [n + 1 for n in lens]
        toks = model.forward([[t] for t in toks], seq_ids, lens)

    for _ in range(args.warmup):
        step()
    first_ctx = lens[0] + 1
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    local_s = time.perf_counter() - t0
    dp.barrier()
    units, max_s = dp.reduce_job(B * args.steps, local_s)
    last_ctx = lens[0]

    if rank != 0:
        return
    ms_per_step = max_s / args.steps * 1e3
    mean_ctx = (first_ctx + last_ctx) / 2
    W = weight_bytes(cfg, e)
    kv_token = 2 * cfg["num_hidden_layers"] * cfg["num_key_value_heads"] * (cfg["hidden_size"] // cfg["num_attention_heads"]) * e
    step_bytes = W + B * mean_ctx * kv_token + B * kv_token
    step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
    result = {
        "metric": f"decode tok/s ({args.model} {args.dtype}, batch {B}/GPU, {S}-in/{gen_total}-out; prefill tok/s alongside)",
        "value": round(units / max_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bfloat16" else "f16",
        "data": "synthetic (random-init weights, uniform random prompt ids)",
        "config": {"workload": f"BASELINE.json configs[2]: {args.model}, batch {B} per GPU, {S}-token prompts, "
                               f"{gen_total} generated tokens (prefill forward, then decode forwards; "
                               f"context {first_ctx}..{last_ctx} in the timed steps)",
                   "global_batch": B * world, "prompt_len": S, "gen_len": gen_total,
                   "parallelism": f"request-sharded dp{world} (independent replicas, no collective)",
                   "hip_graph": not args.no_hip_graph, "fuse_qkv": args.fuse_qkv,
                   "skinny_gemm": args.skinny_gemm, "packed_decode_weights": getattr(args, "packed_weights", True),
                   "kv_blocks": num_blocks, "decode_graphs_captured": len(getattr(model, "_decode_graphs", {}))},
        "prefill_tok_s": round(prefill_units / prefill_max_s, 1),
        "prefill_ms": round(prefill_max_s * 1e3, 2),
        "step_roofline": {"bound": "hbm", "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(step_gbs / HBM_PEAK_GBS, 4),
                          "frac_of_measured_copy": round(step_gbs / HBM_COPY_GBS, 4),
                          "bytes_per_step": int(step_bytes), "weights_bytes": int(W),
                          "kv_bytes": int(step_bytes - W)},
    }
    # `roofline` = the kernel with the largest share of the step; the other hand-written heavyweight next to it
    attn = kernel_roofline(model, lens, args.kernel_iters)
    gemm = gemm_roofline(model, B, args.kernel_iters)
    if gemm is not None and gemm["us_per_launch"] > attn["us_per_launch"]:
        result["roofline"], result["roofline_paged_attention"] = gemm, attn
    else:
        result["roofline"] = attn
        if gemm is not None:
            result["roofline_up_gate_gemm"] = gemm
    if world == 1 and not args.no_cpu_baseline:
        del model
        torch.cuda.empty_cache()
        result["cpu_baseline"] = cpu_baseline(cfg, B, 16, args.dtype)
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
