#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N and no launcher environment (WORLD_SIZE unset) the script starts the N ranks itself, one process per GPU
(HIP_VISIBLE_DEVICES=i), and rank 0 prints the line.

The KV pool is the real one: `LlamaModel.profile_num_blocks()` at gpu_mem_utilization 0.97 (~125 k blocks = 262 GB for
Llama-3-8B on a 288 GB MI355X; `config.kv_pool_blocks` / `kv_pool_gb`), and the run's sequences live in the HIGHEST
block ids of it (filler sequences hold the low ones): every pool offset the timed kernels form is far beyond 2^31
elements.

Workload (BASELINE.json configs[2]): Llama-3-8B, bf16, random-init weights, batch 32 per GPU, 1024-token synthetic
prompts, 128 generated tokens. One "step" = one decode forward of the whole batch (`LlamaModel.forward`, the hot
path) with everything resident in HBM. The K timed steps always sit in the MIDDLE of the 128-token generation —
contexts 1025 + (128-K)/2 ... whatever --steps is, so their mean context is the mean context of a full
1024-in/128-out run (1088.5) and `value` is on the workload the config names, not on a lighter one: after the real
prefill the sequences are advanced to the first warm-up context directly (their KV slots for the skipped positions
hold the N(0,1) data the pool was filled with). The prompt phase (one 32x1024-token prefill forward) runs before
the timed region and is reported separately as `prefill_tok_s`. `value` = decode tokens/s of the whole job =
n_gpus * batch * K / max-over-ranks time of the K timed steps (request-sharded replicas, weak scaling, no
collective on the data path). Each rank pins itself to the NUMA-local cores of its GPU.

Extra objects on the JSON line:
  roofline        the kernel with the largest share of the decode step, timed live with HIP events on the launch
                  stream over launches that cycle through all layers' weights / KV (far past the 256 MiB Infinity
                  Cache), at the decode loop's exact entry point, shapes and launch geometry: algorithmic bytes per
                  launch / mean launch duration. The other heavyweight rides along (roofline_paged_attention /
                  roofline_up_gate_gemm). The attention entry is swl_paged_attn_decode_qkv fed by real split-K slabs —
                  the variant the step runs.
  step_roofline   the whole decode step against HBM: (weights + KV read + KV write) / step time.
  eager           the same K steps with hipGraph replay off (--no-hip-graph; replay is the product default since r03).
  reference_triton rank 0, N=1 only: the REFERENCE's own LlamaModel.forward (its Triton kernels compiled for gfx950 +
                  F.linear; oracle/ref_triton.py on the staged oracle/_ref copy) timed in a child process on the same
                  box, same batch, same timed contexts, after this process has released the GPU. Skipped when
                  oracle/_ref is absent.
  configs1_batch1 / configs3_llama2_7b_4x16k   (rank 0, N=1) short driver-measured runs of BASELINE configs[1]
                  and configs[3] (KV pre-filled, no 16k prefill).
  cpu_baseline    rank 0, N=1 only: the CPU oracle's decode forward (oracle/ref_model.py — the reference has no CPU path
                  of its own) on the host cores at FULL depth (all layers timed, nothing extrapolated):
                  one warm-up + 2 timed steps at the bench's batch and mean timed context (~8 s per step on 128 threads).
  prefill_roofline / prefill_attention_roofline   the prompt forward against the dense MFMA peak (executed flops / wall
                  time; ~90 % of it is hipBLASLt, the reference's own call) and the hand-written prefill attention kernel
                  alone (HIP events on its launch stream, causal flops / mean launch).
  decode_batch128 / decode_batch256   (rank 0, N=1) decode-only steps at the batch sizes the 264 GB pool is sized for
                  (projections on csrc/gemm_wide.hip where it beats the library), with step_roofline.
  reference_triton also carries the reference's prompt pass (prefill_tok_s) from the same child process.
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
GEN_LEN = 128               # BASELINE.json configs[2]: 1024-in / 128-out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU")
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--gen-len", type=int, default=GEN_LEN)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama2-7b", "tiny"])
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"])
    ap.add_argument("--no-hip-graph", action="store_true")
    ap.add_argument("--no-fuse-qkv", dest="fuse_qkv", action="store_false")
    ap.add_argument("--no-skinny-gemm", dest="skinny_gemm", action="store_false")
    ap.add_argument("--no-splitk-fusion", dest="splitk_fusion", action="store_false")
    ap.add_argument("--no-packed-weights", dest="packed_weights", action="store_false")
    ap.add_argument("--no-rows-decode", dest="rows_decode", action="store_false",
                    help="A/B: split-K + consumer launches for o_proj / down_proj instead of the row-owned kernels (r05)")
    ap.add_argument("--decode-engine", dest="decode_engine", action="store_true",
                    help="one-sequence steps through the persistent decode engine (csrc/decode_engine.hip) instead of the multi-launch path (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1] / configs[3] / eager side runs")
    ap.add_argument("--skip-prefill", action="store_true", help="fill the KV pool directly instead of running the prompt")
    ap.add_argument("--kernel-iters", type=int, default=256)
    ap.add_argument("--kv-blocks", type=int, default=0,
                    help="KV pool size in blocks; 0 = profile_num_blocks() at gpu_mem_utilization 0.97 (the product's sizing)")
    ap.add_argument("--kv-placement", default="top", choices=["top", "bottom"],
                    help="top: filler sequences occupy the low block ids, the run's sequences get the highest ones")
    ap.add_argument("--no-reference", action="store_true", help="skip the reference-Triton child run")
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


FILLER_BLOCKS_PER_SEQ = 8192    # block-table columns: a filler sequence holds up to this many blocks


def build_model(args, cfg, min_blocks, batch, max_len, hip_graph, max_tokens=0):
    """The model with its KV pool. Pool size: --kv-blocks, else what the product's own sizing gives on this GPU
    (LlamaModel.profile_num_blocks at gpu_mem_utilization 0.97: reference model.py:94-131), never less than the run
    needs. With --kv-placement top, filler sequences (ids batch, batch+1, ...) take the low block ids so that the run's
    sequences are allocated — lowest free id first, as always — at the very top of the pool."""
    import torch
    from swiftllm_amd import EngineConfig, LlamaModel
    path = tempfile.mkdtemp(prefix="swl_bench_")
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f)
    ec = EngineConfig(model_path=path, use_dummy=True, block_size=16, gpu_mem_utilization=0.97,
                      num_cpu_blocks=int(getattr(args, "num_cpu_blocks", 0) or 0), max_seqs_in_block_table=max(64, batch) + 64,
                      max_blocks_per_seq=max(FILLER_BLOCKS_PER_SEQ, max_len // 16 + 8),
                      max_batch_size=batch, max_tokens_in_batch=max_tokens or batch * min(max_len, 8192),
                      dtype=args.dtype, fuse_qkv=args.fuse_qkv, use_hip_graph=hip_graph,
                      use_skinny_gemm=args.skinny_gemm, tuning=dict(fuse_splitk_consumers=args.splitk_fusion, rows_decode=getattr(args, "rows_decode", True),
                                  decode_engine=getattr(args, "decode_engine", False)),
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
    num_blocks = int(getattr(args, "kv_blocks", 0) or 0)
    if num_blocks <= 0:
        num_blocks = model.profile_num_blocks()
    num_blocks = max(num_blocks, min_blocks)
    model.init_kvcache_and_swap(num_blocks)
    with torch.inference_mode():        # positions the bench skips over must hold data, not zeros (DVFS: §5.4 rule 25)
        chunk = 4096                    # (in slices: one 130 GB normal_() would need a 64-bit element counter per launch)
        for b0 in range(0, num_blocks, chunk):
            model.k_cache[b0:b0 + chunk].normal_(0.0, 1.0, generator=g)
            model.v_cache[b0:b0 + chunk].normal_(0.0, 1.0, generator=g)
    model.bench_filler_ids = []
    if getattr(args, "kv_placement", "top") == "top":
        spare = num_blocks - min_blocks
        sid = max(64, batch)
        while spare > 0 and sid < ec.max_seqs_in_block_table:
            n = min(spare, FILLER_BLOCKS_PER_SEQ)
            model.gpu_block_manager.allocate_blocks_for_seqs([sid], [n * 16])
            model.bench_filler_ids.append(sid)
            spare -= n
            sid += 1
    return model


def pool_report(model):
    """What the pool is and where the run's blocks sit in it (for the JSON line)."""
    host = model.gpu_block_manager.host
    used = [b for sid, blocks in host.seq_blocks.items() if sid not in set(model.bench_filler_ids) for b in blocks]
    block_elems = model.k_cache[0].numel()
    return {"kv_pool_blocks": int(model.num_blocks),
            "kv_pool_gb": round(2 * model.k_cache.numel() * model.k_cache.element_size() / 1e9, 1),
            "kv_filler_blocks": int(sum(len(host.seq_blocks.get(s, ())) for s in model.bench_filler_ids)),
            "kv_run_block_ids": [int(min(used)), int(max(used))] if used else None,
            "kv_run_min_element_offset": int(min(used)) * block_elems if used else None}


def timed(fn):
    import torch
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


def kv_token_bytes(cfg, e):
    return 2 * cfg["num_hidden_layers"] * cfg["num_key_value_heads"] * (cfg["hidden_size"] // cfg["num_attention_heads"]) * e


def step_roofline(cfg, e, batch, mean_ctx, ms_per_step):
    W = weight_bytes(cfg, e)
    kvt = kv_token_bytes(cfg, e)
    step_bytes = W + batch * mean_ctx * kvt + batch * kvt
    gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
    # the other roof (at 128 / 256 tokens the projections approach the ridge): 2 flops per weight per
    # token + QK^T and PV over the context; time at the dense MFMA peak next to time at the HBM peak
    h, L = cfg["hidden_size"], cfg["num_hidden_layers"]
    flops = 2.0 * batch * (W / e) + 4.0 * batch * mean_ctx * h * L
    t_hbm_ms, t_mfma_ms = step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, flops / (MFMA_PEAK_TFLOPS * 1e12) * 1e3
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(gbs / HBM_COPY_GBS, 4),
            "bytes_per_step": int(step_bytes), "weights_bytes": int(W), "kv_bytes": int(step_bytes - W),
            "mfma_roof": {"flops_per_step": int(flops), "achieved_tflops": round(flops / (ms_per_step * 1e-3) / 1e12, 1),
                          "peak_tflops": MFMA_PEAK_TFLOPS, "frac": round(t_mfma_ms / ms_per_step, 4),
                          "ms_at_mfma_peak": round(t_mfma_ms, 3), "ms_at_hbm_peak": round(t_hbm_ms, 3)}}


class DecodeRun:
    """`batch` sequences decoding in lock step on one model; contexts can be set directly (block allocation follows)."""

    def __init__(self, model, batch, vocab, seed):
        import torch
        self.model, self.batch = model, batch
        self.seq_ids = list(range(batch))
        g = torch.Generator().manual_seed(seed)
        self.toks = torch.randint(0, vocab, (batch,), generator=g).tolist()
        self.lens = [0] * batch

    def prefill(self, prompts):
        self.toks = self.model.forward(prompts, self.seq_ids, [])
        self.lens = [len(p) for p in prompts]
        return self.toks

    def jump_to(self, context):
        """Next step decodes at `context` (= length including the new token)."""
        assert all(context - 1 >= n for n in self.lens)
        self.lens = [context - 1] * self.batch

    def step(self):
        self.lens = [n + 1 for n in self.lens]
        self.toks = self.model.forward([[t] for t in self.toks], self.seq_ids, self.lens)

    def timed_steps(self, warmup, steps, barrier=lambda: None):
        """EXACTLY `steps` forwards between two barrier + synchronize pairs. Per-step wall times and the interpreter's
        garbage collections inside the region are recorded on the side (self.region): a 20-step region is 80 ms, and one
        generation-2 collection of the interpreter (tens of ms with the prompt lists and 8 G parameters' worth of tensor
        objects alive) would be a third of it — so the collector is run BEFORE the region and paused inside it, as a
        serving loop would do around its latency-critical section. The forwards themselves are untouched."""
        import gc
        import torch
        for _ in range(warmup):
            self.step()
        first = self.lens[0] + 1
        gc_log = []
        def on_gc(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gc_log.append((info.get("generation"), (time.perf_counter() - _t[0]) * 1e3))
        gc.collect()
        gc_was_enabled = gc.isenabled()
        if os.environ.get("SWL_BENCH_KEEP_GC") != "1":
            gc.disable()
        gc.callbacks.append(on_gc)
        per_step = []
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            self.step()
            per_step.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        gc.callbacks.remove(on_gc)
        if gc_was_enabled:
            gc.enable()
        ps = sorted(per_step)
        self.region = {"step_ms_min": round(ps[0] * 1e3, 4), "step_ms_median": round(ps[len(ps) // 2] * 1e3, 4),
                       "step_ms_max": round(ps[-1] * 1e3, 4), "slowest_step_index": per_step.index(ps[-1]),
                       "gc_collections_in_region": len(gc_log), "gc_ms_in_region": round(sum(t for _, t in gc_log), 3),
                       "gc_paused": not gc.isenabled() or os.environ.get("SWL_BENCH_KEEP_GC") != "1"}
        return dt, first, self.lens[0]

    def release(self):
        self.model.free_seqs_resources(self.seq_ids)
        self.lens = [0] * self.batch


def _event_time(launch, iters, warm):
    import torch
    for i in range(warm):
        launch(i)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for i in range(iters):
        launch(i)
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) * 1e3 / iters


def _pmc_traffic(name, alg_bytes, ok):
    """HBM traffic cannot be read from inside this process: it comes from the committed rocprofv3 PMC passes on this
    kernel (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/*.json), scaled by bytes to this launch; null for
    a kernel specialisation that was not profiled."""
    path = os.path.join(ROOT, "profiles", name)
    if not ok or not os.path.exists(path):
        return None, None
    with open(path, encoding="utf-8") as f:
        pmc = json.load(f)
    return (int(alg_bytes * pmc["traffic_over_algorithmic"]),
            f"COMMITTED measurement, not taken in this run: profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE in separate "
            f"passes over this kernel at this shape), scaled by bytes")


def attention_roofline(model, lens, iters):
    """Mean duration of the decode step's attention launch: swl_paged_attn_decode_qkv (rotary + KV store in its
    prologue) fed by the split-K slabs of a real fused-qkv projection, at the decode loop's shapes and launch
    geometry, HIP events on the launch stream, cycling through all layers' KV. Falls back to the plain
    swl_paged_attn_decode entry when the decode path does not use the slab-fed variant."""
    import torch
    from swiftllm_amd import _hip
    from swiftllm_amd.worker.batch_plan import plan_batch
    from swiftllm_amd.worker.kernels.linear import SplitKPartials, linear_splitk
    from swiftllm_amd.worker.kernels.paged_attn import paged_attention, paged_attention_from_qkv_splitk
    import types
    mc, ecfg = model.model_config, model.engine_config
    B, H, KVH, D, L = len(lens), mc.num_q_heads, mc.num_kv_heads, mc.head_dim, mc.num_layers
    plan = plan_batch([[0]] * B, list(range(B)), lens, KVH, model._num_slots)
    sbs, nsb = model._graph_bucket(plan) if ecfg.use_hip_graph else (plan.seq_block_size, plan.num_seq_blocks)
    dev = model.device
    bt = model.gpu_block_manager.block_table
    st = types.SimpleNamespace(
        num_decoding_seqs=B, num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb, softmax_scale=D ** -0.5,
        decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=dev),
        seq_ids=torch.arange(B, dtype=torch.int32, device=dev), position_cos=model._cos_cached,
        position_sin=model._sin_cached, position_indices=torch.tensor([n - 1 for n in lens], dtype=torch.int32, device=dev),
        paged_attn_scratch=torch.empty(_hip.scratch_bytes(B, H, D, nsb) // 4 + 4, dtype=torch.float32, device=dev))
    o = torch.empty(B, H, D, device=dev, dtype=model.dtype)
    x = torch.randn(B, mc.hidden_size, device=dev, dtype=torch.float32).to(model.dtype)
    w_qkv = getattr(model.weight.layers[0], "qkv_proj", None)
    slab_fed = (w_qkv is not None and getattr(ecfg, "use_skinny_gemm", False) and getattr(ecfg, "fuse_splitk_consumers", True)
                and getattr(ecfg, "fuse_rope_into_attention", True) and D in (32, 64, 128))
    part = linear_splitk(x, w_qkv, always=True) if slab_fed else None
    if isinstance(part, SplitKPartials):
        part = SplitKPartials(part.slabs[:part.k_splits * B * part.shape[1]].clone(), part.k_splits, B, part.shape[1], part.dtype)
        entry = "swl_paged_attn_decode_qkv (paged_attn_phase1_kernel<QKV>: rotary + KV store + attention)"

        def launch(i):
            paged_attention_from_qkv_splitk(part, model.k_cache, model.v_cache, bt, mc, ecfg, st, i % L, o)
    else:
        q = torch.randn(B, H, D, device=dev, dtype=torch.float32).to(model.dtype)
        entry = "swl_paged_attn_decode (paged_attn_phase1_kernel)"

        def launch(i):
            paged_attention(q, model.k_cache, model.v_cache, bt, mc, ecfg, st, i % L, o)
    us = _event_time(launch, iters, min(iters, 2 * L))
    e = model.dtype.itemsize
    kv_bytes = sum(lens) * 2 * KVH * D * e
    splits = sum(-(-n // sbs) for n in lens)
    part_bytes = (2 * splits * H * (D + 1) * 4) if nsb > 1 else 0
    slab_bytes = part.k_splits * B * part.shape[1] * 4 if isinstance(part, SplitKPartials) else B * H * D * e
    alg_bytes = kv_bytes + slab_bytes + B * H * D * e + part_bytes
    gbs = alg_bytes / (us * 1e-6) / 1e9
    pmc_name = "r04_paged_attn_qkv_pmc.json" if isinstance(part, SplitKPartials) else "r01_paged_attn_pmc.json"
    traffic, src = _pmc_traffic(pmc_name, alg_bytes, (H, KVH, D) == (32, 8, 128) and nsb == 1)
    return dict(bound="hbm", kernel=entry, achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(gbs / HBM_PEAK_GBS, 4), frac_of_measured_copy=round(gbs / HBM_COPY_GBS, 4),
                traffic=traffic, traffic_source=src, bytes_per_launch=int(alg_bytes), us_per_launch=round(us, 2),
                seq_block_size=sbs, num_seq_blocks=nsb, launches=iters)


def gemm_roofline(model, batch, iters):
    """Mean duration of the up/gate projection + SiLU-gate (54 % of the weight bytes of a decode step), HIP events on
    the launch stream, cycling through all layers' weights (7.5 GB footprint >> the 256 MiB Infinity Cache). None
    when the decode path does not use it."""
    import torch
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
    # the entry the decode step runs: since r05 (rows_decode, bfloat16) the SiLU-gate GEMM normalises the raw residual rows
    # on the fly (csrc/gemm_skinny.hip NF); else the plain packed / row-major forms
    nf = packed and getattr(ecfg, "rows_decode", False) and model.dtype == torch.bfloat16
    fn = ("swl_gemm_skinny_packed_silu_gate_nf" if nf else "swl_gemm_skinny_packed_silu_gate") if packed else "swl_gemm_skinny_silu_gate"
    srcs = [(l.up_gate_proj._swl_packed if packed else l.up_gate_proj) for l in layers]
    norms = [l.ffn_norm for l in layers]

    def launch(i):
        j = i % len(srcs)
        if nf:
            _hip.call(fn, out.data_ptr(), x.data_ptr(), norms[j].data_ptr(), mc.rms_norm_eps, srcs[j].data_ptr(), M, I, K, K, I,
                      code, _hip.stream())
        else:
            _hip.call(fn, out.data_ptr(), x.data_ptr(), srcs[j].data_ptr(), M, I, K, K, I, code, _hip.stream())
    us = _event_time(launch, iters, min(iters, len(layers)))
    e = model.dtype.itemsize
    alg_bytes = 2 * I * K * e + M * K * e + M * I * e
    gbs = alg_bytes / (us * 1e-6) / 1e9
    pmc_name = ("r06b_gemm_silu_nf_pmc.json" if nf else "r04_gemm_silu_packed_pmc.json") if packed else "r01e_gemm_silu_pmc.json"
    traffic, src = _pmc_traffic(pmc_name, alg_bytes, (I, K) == (14336, 4096) and model.dtype == torch.bfloat16)
    return dict(bound="hbm", kernel="%s (gemm_skinny_ring_kernel<SiluGate%s%s>: up/gate projection + SiLU-gate)" % (
                    fn, ", packed W" if packed else "", ", norm on the fly" if nf else ""),
                achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                frac_of_measured_copy=round(gbs / HBM_COPY_GBS, 4), traffic=traffic, traffic_source=src,
                bytes_per_launch=int(alg_bytes), us_per_launch=round(us, 2), launches=iters)


MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16/fp16 dense MFMA ~2.5 PF (measured 2495 TF with 32x32x16)


def prefill_flops(cfg, lens):
    """Arithmetic of one prompt pass as executed: the projections on every token, causal attention (half the S x S
    products), lm_head on the last token of each sequence only (post_layer.py:18-40)."""
    L, h, I, V = cfg["num_hidden_layers"], cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    H = cfg["num_attention_heads"]
    D = h // H
    kvd = cfg["num_key_value_heads"] * D
    gemm = 2 * L * (2 * h * h + 2 * kvd * h + 3 * I * h) * sum(lens) + 2 * len(lens) * V * h
    attn = L * sum(2 * n * n * D * H for n in lens)
    return gemm, attn


def prefill_attention_roofline(model, lens, iters=24):
    """The one hand-written COMPUTE-bound kernel of the path — csrc/prefill_attn.hip, varlen causal flash attention — at
    the prompt pass's shapes, HIP events on the launch stream: causal flops (4 * sum(len^2) * D * H / 2) per launch / mean
    launch duration against the dense MFMA peak."""
    import types
    import torch
    from swiftllm_amd.worker.kernels.prefill_attn import prefill_attention
    mc = model.model_config
    H, KVH, D = mc.num_q_heads, mc.num_kv_heads, mc.head_dim
    P = sum(lens)
    dev = model.device
    q = torch.randn(P, H, D, device=dev, dtype=torch.float32).to(model.dtype)
    k = torch.randn(P, KVH, D, device=dev, dtype=torch.float32).to(model.dtype)
    v = torch.randn(P, KVH, D, device=dev, dtype=torch.float32).to(model.dtype)
    o = torch.empty_like(q)
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int32), 0)
    st = types.SimpleNamespace(num_prefill_seqs=len(lens), max_prefill_len=max(lens), softmax_scale=D ** -0.5,
                               prefill_seq_start_locs_with_end=cu.to(dev))
    us = _event_time(lambda i: prefill_attention(q, k, v, o, mc, None, st), iters, 3)
    flop = sum(2 * n * n * D * H for n in lens)
    tf = flop / (us * 1e-6) / 1e12
    return dict(bound="mfma", kernel="swl_prefill_attn_varlen (prefill_attn_kernel: varlen causal GQA flash attention, MFMA 32x32x16)",
                achieved=round(tf, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_PEAK_TFLOPS, 4),
                flops_per_launch=int(flop), us_per_launch=round(us, 1), launches=iters, traffic=None)


def cpu_baseline(cfg, batch, context, dtype, steps=2):
    """The CPU oracle (a port: the reference has no CPU forward) on the host cores: the FULL-DEPTH decode forward of the same
    architecture at the bench's batch and MEAN TIMED CONTEXT — all layers timed, nothing extrapolated (r03 timed one layer
    and multiplied). KV pool filled directly with N(0,1) data, no prompt pass; one untimed warm-up step, then `steps` timed
    decode steps (~10 s each on a 128-core host). Layer 0's weights are drawn at random and layers 1.. are rolled copies of
    them (distinct memory — nothing is served from cache that a real checkpoint would stream — without minutes of CPU
    randn for 8 G parameters). This is the ONLY part of bench.py that touches oracle/."""
    import torch
    from oracle import eager_ops, synth
    from oracle.ref_model import RefLlamaModel
    from swiftllm_amd import EngineConfig, LlamaModelConfig
    L = cfg["num_hidden_layers"]
    tdtype = torch.bfloat16 if dtype == "bfloat16" else torch.float16
    sd = synth.make_state_dict(dict(cfg, num_hidden_layers=1), seed=0, dtype=tdtype)
    for i in range(1, L):
        for k in [k for k in sd if k.startswith("model.layers.0.")]:
            sd[k.replace("model.layers.0.", f"model.layers.{i}.")] = torch.roll(sd[k], shifts=i, dims=0)
    blocks_per_seq = (context + steps + 1) // 16 + 2
    ec = EngineConfig(model_path="", use_dummy=True, block_size=16, gpu_mem_utilization=0.9,
                      num_cpu_blocks=0, max_seqs_in_block_table=batch, max_blocks_per_seq=blocks_per_seq + 2,
                      max_batch_size=batch, max_tokens_in_batch=batch * 16)
    eager_ops.linear = lambda a, w: torch.nn.functional.linear(a, w)    # native 16-bit CPU GEMM
    ref = RefLlamaModel(LlamaModelConfig(cfg), ec, sd, tdtype, dense_decode_attention=True)
    ref.init_kvcache_and_swap(batch * blocks_per_seq)
    g = torch.Generator().manual_seed(1)
    one = torch.randn(ref.k_cache[:, :1].shape, generator=g).to(tdtype)          # one layer's worth, shared by all layers
    ref.k_cache.copy_(one.expand_as(ref.k_cache))
    ref.v_cache.copy_(one.flip(0).expand_as(ref.v_cache))
    del one
    toks = torch.randint(0, cfg["vocab_size"], (batch,), generator=g).tolist()
    first = context - steps // 2
    total = 0.0
    for s in range(-1, steps):      # step -1: untimed warm-up (allocates the blocks, touches the weights)
        t0 = time.perf_counter()
        toks = ref.forward([[t] for t in toks], list(range(batch)), [first + s] * batch)
        if s >= 0:
            total += time.perf_counter() - t0
    step_s = total / steps
    return dict(value=round(batch / step_s, 3), unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                extrapolated=False,
                sample=(f"oracle/ref_model.py decode forward at full depth ({L} layers, all timed), batch {batch}, contexts "
                        f"{first}..{first + steps - 1} (KV pool filled directly; attention over the full context as one "
                        f"dense softmax per sequence), {steps} steps of {step_s:.2f} s after one warm-up step, native 16-bit "
                        f"CPU GEMM on {torch.get_num_threads()} threads"))


def reference_triton_leg(batch, first_ctx, steps, warmup, prompt_len):
    """The reference's own data plane on this GPU (oracle/ref_triton.py: its LlamaModel.forward, its Triton kernels
    compiled by Triton's gfx950 backend, F.linear; fp16 — the only precision the reference has) timed in a child
    process at the SAME batch and the SAME timed contexts as `value`. Runs after this process has released its model
    (the reference builds its own 16 GB of weights). None when oracle/_ref is not staged or the child fails: a side
    measurement must never take the bench line down."""
    import subprocess
    if not os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py")):
        return None
    cmd = [sys.executable, "-m", "oracle.ref_triton", "bench", "--config", "c2", "--batch", str(batch),
           "--first-context", str(first_ctx), "--steps", str(steps), "--warmup", str(warmup),
           "--prefill-len", str(prompt_len)]
    env = dict(os.environ)
    env.pop("TRITON_INTERPRET", None)
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return dict(value=d["decode_tok_s"], unit="tokens/s", ms_per_step=d["ms_per_step"], dtype=d["dtype"],
                    contexts=[d["context_first"], d["context_last"]], steps=d["steps"], warmup=d["warmup"],
                    prefill_tok_s=d.get("prefill_tok_s"), prefill_ms=d.get("prefill_ms"),
                    path=d["path"], how="child process: " + " ".join(cmd[1:]))
    except Exception as e:     # noqa: BLE001
        print(f"[bench] reference Triton leg failed ({type(e).__name__}: {e})", file=sys.stderr)
        return None


def reference_prefill_leg(batch, prompt_len):
    """The reference's prompt pass at a long-prompt shape (its Triton flash-attention kernel + F.linear, fp16) in a child
    process; None when oracle/_ref is not staged or the child fails."""
    import subprocess
    if not os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py")):
        return None
    cmd = [sys.executable, "-m", "oracle.ref_triton", "bench", "--config", "c2", "--batch", str(batch),
           "--first-context", str(prompt_len + 8), "--steps", "1", "--warmup", "1", "--prefill-len", str(prompt_len)]
    env = dict(os.environ)
    env.pop("TRITON_INTERPRET", None)
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return dict(prefill_tok_s=d.get("prefill_tok_s"), prefill_ms=d.get("prefill_ms"), dtype=d["dtype"], path=d["path"])
    except Exception as e:     # noqa: BLE001
        print(f"[bench] reference long-prompt leg failed ({type(e).__name__}: {e})", file=sys.stderr)
        return None


def prefill_long_leg(args, shapes=((4, 16384), (1, 32768))):
    """The regime the reference publishes (README.md:93-101: one forward from (128, 128) up to (1, 131072) input tokens):
    ONE real LlamaModel.forward prompt pass per shape at Llama-3-8B dims — projections in row blocks through hipBLASLt,
    rotary + KV store, the hand-written flash-attention kernel (csrc/prefill_attn.hip), lm_head on the last tokens. Here
    attention is 20-40 % of the flops (4 % at 32 x 1024). Reports tok/s, the fraction of the dense MFMA peak, and the
    attention kernel's own time (HIP events at the same shape) as a share of the pass."""
    import torch
    longest = max(n for _, n in shapes)
    tokens = max(b * n for b, n in shapes)
    cfg = ensure_positions(model_config_dict("llama3-8b"), longest + 64)
    need = max(b * (-(-(n + 8) // 16) + 1) for b, n in shapes) + 8
    big_args = argparse.Namespace(**dict(vars(args), kv_blocks=need + 64, kv_placement="bottom"))
    model = build_model(big_args, cfg, need, max(b for b, _ in shapes), longest + 64, False, max_tokens=tokens)
    g = torch.Generator().manual_seed(5)
    out = {}
    for b, n in shapes:
        prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for _ in range(b)]
        ids = list(range(b))
        model.forward(prompts, ids, [])                      # untimed: GEMM heuristics, allocator pools
        model.free_seqs_resources(ids)
        _, sec = timed(lambda: model.forward(prompts, ids, []))
        model.free_seqs_resources(ids)
        gemm_f, attn_f = prefill_flops(cfg, [n] * b)
        tf = (gemm_f + attn_f) / sec / 1e12
        leg = dict(batch=b, prompt_len=n, prefill_ms=round(sec * 1e3, 2), prefill_tok_s=round(b * n / sec, 1),
                   roofline={"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "flops_per_pass": int(gemm_f + attn_f),
                             "attention_share_of_flops": round(attn_f / (gemm_f + attn_f), 4)})
        try:
            attn = prefill_attention_roofline(model, [n] * b, iters=4)
            leg["attention_kernel"] = {k: attn[k] for k in ("achieved", "frac", "us_per_launch", "unit")}
            leg["attention_share_of_time"] = round(attn["us_per_launch"] * cfg["num_hidden_layers"] / (sec * 1e6), 4)
        except Exception as exc:     # noqa: BLE001
            print(f"[bench] long-prompt attention timing failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        out[f"{b}x{n}"] = leg
    del model
    torch.cuda.empty_cache()
    return out


def side_run(args, model_name, batch, context, steps, warmup, label, model=None):
    """A short decode-only measurement of another BASELINE config: KV for `context - warmup - 1` positions is taken
    as it lies in the (N(0,1)-filled) pool, `warmup` + `steps` decode forwards run from there."""
    import torch
    cfg = ensure_positions(model_config_dict(model_name), context + steps + warmup + 2)
    e = 2
    own = model is None
    if own:
        max_len = context + steps + warmup + 2
        num_blocks = batch * (-(-max_len // 16) + 1) + 8
        model = build_model(args, cfg, num_blocks, batch, max_len, not args.no_hip_graph)
    run = DecodeRun(model, batch, cfg["vocab_size"], seed=7)
    run.jump_to(context - (steps // 2) - warmup)
    dt, first, last = run.timed_steps(warmup, steps)
    run.release()
    ms = dt / steps * 1e3
    out = dict(workload=label, model=model_name, batch=batch, context_first=first, context_last=last, steps=steps,
               warmup=warmup, ms_per_step=round(ms, 4), decode_tok_s=round(batch * steps / dt, 1),
               hip_graph=bool(model.engine_config.use_hip_graph),
               step_roofline=step_roofline(cfg, e, batch, (first + last) / 2, ms))
    if own:
        del run, model
        torch.cuda.empty_cache()
    return out


def mixed_step_run(model, cfg, batch, prompt_len, context, n_prefill=4, reps=5):
    """The batch shape BASELINE configs[2] names ("piggybacked prefill+decode"): `n_prefill` fresh prompt_len-token
    prompts and `batch - n_prefill` sequences decoding at ~`context` in ONE forward — what the scheduler emits with
    piggyback=True; the data plane runs the decodes' paged attention on a side stream next to the prompts' flash attention
    (transformer_layer.py) — against the same prompts alone and the same decodes alone. Medians of `reps` forwards each,
    host-synchronised (mixed and prompt-only forwards are eager launches; the decode-only one replays its hipGraph)."""
    import statistics
    import torch
    g = torch.Generator().manual_seed(11)
    vocab = cfg["vocab_size"]
    pre_ids, dec_ids = list(range(n_prefill)), list(range(n_prefill, batch))
    prompts = [torch.randint(0, vocab, (prompt_len,), generator=g).tolist() for _ in pre_ids]
    toks = torch.randint(0, vocab, (len(dec_ids),), generator=g).tolist()
    lens = [context - 2 * reps - 4] * len(dec_ids)        # KV as it lies in the (N(0,1)-filled) pool, as side_run
    t = {"decode_only": [], "prefill_only": [], "mixed": []}
    for rep in range(reps + 1):                            # (rep 0: warm-up — graph capture, GEMM heuristics)
        lens = [n + 1 for n in lens]
        toks, dt = timed(lambda: model.forward([[x] for x in toks], dec_ids, lens))
        t["decode_only"].append(dt)
        _, dt = timed(lambda: model.forward(prompts, pre_ids, []))
        t["prefill_only"].append(dt)
        model.free_seqs_resources(pre_ids)
        lens = [n + 1 for n in lens]
        out, dt = timed(lambda: model.forward(prompts + [[x] for x in toks], pre_ids + dec_ids, lens))
        t["mixed"].append(dt)
        toks = out[n_prefill:]
        model.free_seqs_resources(pre_ids)
    model.free_seqs_resources(dec_ids)
    ms = {k: statistics.median(v[1:]) * 1e3 for k, v in t.items()}
    return dict(workload=f"{n_prefill} x {prompt_len}-token prompts + {len(dec_ids)} decodes at context ~{context} in one forward",
                mixed_ms=round(ms["mixed"], 3), prefill_only_ms=round(ms["prefill_only"], 3),
                decode_only_ms=round(ms["decode_only"], 3),
                mixed_over_separate=round(ms["mixed"] / (ms["prefill_only"] + ms["decode_only"]), 4),
                tokens_per_s=round((n_prefill * prompt_len + len(dec_ids)) / ms["mixed"] * 1e3, 1), reps=reps)


def side_run_fresh_process(args, label, model_name="llama2-7b", batch=4, prompt_len=16384, gen_len=64, steps=24, warmup=4,
                           extra=(), timeout=420):
    """A side measurement as its own `bench.py --model M --batch B ... --skip-prefill --no-extras` run in a child process
    (this process has released its model). Two users: configs[3] — the 47 GB a step streams then sit in a freshly mapped
    address space (measured as the last side run of THIS process, after a 30 GB model was built and freed, the same code
    ran 15 % slower: 9.7 vs 8.4 ms/step, profiles/r02q_*) — and the persistent decode engine, whose third copy of the
    layer weights wants its own pool sizing and whose failure must not reach the allocator / capture state of this
    process (round-6 driver-form run: an engine built on a model that had already replayed graphs tripped a
    HIPCachingAllocator assert and took the float16 and long-prompt legs down with it). Returns None when the child fails."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--model", model_name, "--batch", str(batch), "--prompt-len",
           str(prompt_len), "--gen-len", str(gen_len), "--skip-prefill", "--steps", str(steps), "--warmup", str(warmup),
           "--no-extras", "--no-cpu-baseline", "--dtype", args.dtype, *extra]
    if args.no_hip_graph:
        cmd.append("--no-hip-graph")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        first, last = d["config"]["timed_contexts"]
        return dict(workload=label, model=model_name, batch=batch, context_first=first, context_last=last, steps=d["steps"],
                    warmup=d["warmup"], ms_per_step=d["ms_per_step"], decode_tok_s=d["value"],
                    hip_graph=bool(d["config"].get("hip_graph")), step_roofline=d["step_roofline"],
                    how="child process: " + " ".join(cmd[1:]),
                    **{k: d["config"][k] for k in ("decode_engine_active", "decode_engine_fallbacks") if k in d["config"]})
    except Exception as e:     # noqa: BLE001 — a side measurement must never take the bench line down
        tail = getattr(e, "stderr", "") or ""
        print(f"[bench] child run failed ({label[:40]}...: {type(e).__name__}: {e}) {tail[-400:]}", file=sys.stderr)
        return None


def main():
    args = parse_args()
    from swiftllm_amd import dp
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves, one process per GPU; rank 0 prints the JSON line
        raise SystemExit(dp.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    # stdout carries exactly ONE line — the JSON result of rank 0. Everything the libraries print on the way
    # ([Model.profile] ..., gloo's connection banner from C++, engine banners) goes to stderr: file descriptor 1 itself
    # points at stderr while the benchmark runs.
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        with contextlib.redirect_stdout(sys.stderr):
            result = _run(args)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    if result is not None:
        print(json.dumps(result), flush=True)


def _run(args):
    from swiftllm_amd import dp
    rank, local_rank, world = dp.env_rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # one GPU per process, seen as device 0, unless the launcher already restricted visibility
    vis = [k for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(k)]
    if world > 1 and not vis:
        os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(0 if (world > 1 and not vis and ndev == 1) else local_rank % ndev)
    pinned = dp.pin_to_local_cores(local_rank, local_world) if world > 1 else []
    dp.init_control_group()

    cfg = model_config_dict(args.model)
    B, S, GEN = args.batch, args.prompt_len, args.gen_len
    K, Wm = args.steps, args.warmup
    # timed contexts: K consecutive steps centred in the GEN-token generation (all of it and beyond when K >= GEN)
    first_timed = S + 1 + max(0, (GEN - K) // 2)
    last_needed = max(S + GEN, first_timed + K)
    ensure_positions(cfg, last_needed + 2)
    blocks_per_seq = -(-(last_needed + 2) // 16)
    num_blocks = int(B * blocks_per_seq * 1.25) + 8
    model = build_model(args, cfg, num_blocks, B, last_needed + 2, not args.no_hip_graph)
    e = model.dtype.itemsize
    pool = None

    # every rank serves its own shard of the requests: `batch` sequences per GPU
    g = torch.Generator().manual_seed(1 + rank)
    prompts = [torch.randint(0, cfg["vocab_size"], (S,), generator=g).tolist() for _ in range(B)]
    run = DecodeRun(model, B, cfg["vocab_size"], seed=1 + rank)

    # ---- prompt phase (reported, outside the K timed steps) ------------------------------------------------
    prefill_units = prefill_max_s = None
    if not args.skip_prefill:
        run.prefill(prompts)                            # untimed: GEMM heuristics, allocator pools
        run.release()
        dp.barrier()
        _, prefill_s = timed(lambda: run.prefill(prompts))
        prefill_units, prefill_max_s = dp.reduce_job(B * S, prefill_s)
    else:
        run.lens = [S] * B

    # ---- decode: W warm-up steps, then exactly K timed steps, centred in the generation --------------------------
    run.jump_to(max(S + 1, first_timed - Wm))
    local_s, first_ctx, last_ctx = run.timed_steps(Wm, K, dp.barrier)
    pool = pool_report(model)
    units, max_s = dp.reduce_job(B * K, local_s)
    graphs = len(getattr(model, "_decode_graphs", {}))
    engine_state = ({"decode_engine_active": getattr(model, "_engine", None) is not None,
                     "decode_engine_fallbacks": int(getattr(model, "engine_fallbacks", 0))}
                    if getattr(args, "decode_engine", False) else {})

    if rank != 0:
        return None
    ms_per_step = max_s / K * 1e3
    mean_ctx = (first_ctx + last_ctx) / 2
    result = {
        "metric": f"decode tok/s ({args.model} {args.dtype}, batch {B}/GPU, {S}-in/{GEN}-out; prefill tok/s alongside)",
        "value": round(units / max_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": K,
        "warmup": Wm, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bfloat16" else "f16",
        "data": "synthetic (random-init weights, uniform random prompt ids)",
        "config": {"workload": f"BASELINE.json configs[2]: {args.model}, batch {B} per GPU, {S}-token prompts, "
                               f"{GEN} generated tokens; the {K} timed decode forwards run at contexts "
                               f"{first_ctx}..{last_ctx} (centred in the generation: mean context {mean_ctx:.1f} vs "
                               f"{S + (GEN + 1) / 2:.1f} for all {GEN} tokens)",
                   "global_batch": B * world, "prompt_len": S, "gen_len": GEN,
                   "timed_contexts": [first_ctx, last_ctx],
                   "parallelism": f"request-sharded dp{world} (independent replicas, no collective)",
                   "hip_graph": not args.no_hip_graph, "fuse_qkv": args.fuse_qkv,
                   "skinny_gemm": args.skinny_gemm, "packed_decode_weights": getattr(args, "packed_weights", True),
                   **pool, "decode_graphs_captured": graphs, **engine_state,
                   "cpu_affinity_cores": len(pinned) if pinned else len(os.sched_getaffinity(0)),
                   "cpu_affinity": dict(dp.last_affinity) if world > 1 else dict(
                       cores=len(os.sched_getaffinity(0)), how="single rank on this host: not pinned, all allowed cores")},
        "step_roofline": step_roofline(cfg, e, B, mean_ctx, ms_per_step),
        # rank 0's view of the timed region: per-forward wall times and interpreter garbage collections inside it
        "timed_region": getattr(run, "region", None),
    }
    if prefill_units is not None:
        result["prefill_tok_s"] = round(prefill_units / prefill_max_s, 1)
        result["prefill_ms"] = round(prefill_max_s * 1e3, 2)
        gemm_f, attn_f = prefill_flops(cfg, [S] * B)
        tf = world * (gemm_f + attn_f) / prefill_max_s / 1e12
        result["prefill_roofline"] = {
            "bound": "mfma", "achieved": round(tf / world, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / world / MFMA_PEAK_TFLOPS, 4), "flops_per_pass": int(gemm_f + attn_f),
            "gemm_share_of_flops": round(gemm_f / (gemm_f + attn_f), 4),
            "what": "whole prompt forward per GPU: executed flops (projections on every token, causal attention, lm_head on "
                    "the last tokens) / wall time of the forward",
            "dominant_kernel": "hipBLASLt Cijk_* MT256x256x64 (F.linear, the reference's own call, kernels/linear.py:3-12): "
                               "~90 % of the pass — profiles/r03_kernel_trace_configs2.md; the hand-written attention kernel "
                               "is priced separately in prefill_attention_roofline"}
    # `roofline` = the kernel with the largest share of the step; the other hand-written heavyweight next to it
    attn = attention_roofline(model, run.lens, args.kernel_iters)
    gemm = gemm_roofline(model, B, args.kernel_iters)
    if gemm is not None and gemm["us_per_launch"] > attn["us_per_launch"]:
        result["roofline"], result["roofline_paged_attention"] = gemm, attn
    else:
        result["roofline"] = attn
        if gemm is not None:
            result["roofline_up_gate_gemm"] = gemm
    if world == 1 and not args.skip_prefill:
        try:
            result["prefill_attention_roofline"] = prefill_attention_roofline(model, [S] * B)
        except Exception as exc:     # noqa: BLE001 — a side measurement must never take the bench line down
            print(f"[bench] prefill attention roofline failed ({type(exc).__name__}: {exc})", file=sys.stderr)
    if world == 1 and not args.no_extras:
        # the same K steps with hipGraph replay off (one HIP launch per kernel from Python), same contexts
        if not args.no_hip_graph:
            model.engine_config.use_hip_graph = False
            run.release()
            run.lens = [0] * B
            run.jump_to(max(S + 1, first_timed - Wm))
            dt, f2, l2 = run.timed_steps(Wm, K)
            model.engine_config.use_hip_graph = True
            result["eager"] = dict(ms_per_step=round(dt / K * 1e3, 4), value=round(B * K / dt, 1), contexts=[f2, l2],
                                   note="hipGraph replay off: one HIP launch per kernel from Python")
        run.release()
        if B > 4:
            try:
                result["piggyback_mixed_step"] = mixed_step_run(model, cfg, B, S, S + GEN // 2)
            except Exception as exc:     # noqa: BLE001 — a side measurement must never take the bench line down
                print(f"[bench] mixed-step side run failed ({type(exc).__name__}: {exc})", file=sys.stderr)
                model.free_seqs_resources(list(range(B)))
        if args.model == "llama3-8b":
            result["configs1_batch1"] = side_run(args, "llama3-8b", 1, 1024 + GEN // 2, 48, 8,
                                                 "BASELINE.json configs[1]: batch 1 decode-only", model=model)
    del run, model
    torch.cuda.empty_cache()
    if world == 1 and not args.no_extras and args.model == "llama3-8b":
        # contexts 16404..16427: just past 16k, and clear of the 16384/16385 boundary where the flash-decoding split
        # width changes bucket — a hipGraph re-capture (~3 steps of time) inside a 24-step timed region is an artefact
        # of where the window sits, not steady-state decode
        label = "BASELINE.json configs[3]: Llama-2-7B dims, batch 4, 16k context"
        result["configs3_llama2_7b_4x16k"] = (side_run_fresh_process(args, label)
                                              or side_run(args, "llama2-7b", 4, 16384 + 32, 24, 4, label))
    if world == 1 and not args.no_extras and args.model == "llama3-8b":
        # the batch sizes the 264 GB pool is sized for: decode-only at 1024-in / mid-generation contexts, one model for both
        try:
            need = 256 * (-(-(S + GEN + 64) // 16) + 1) + 8
            big_args = argparse.Namespace(**dict(vars(args), kv_blocks=need + 1024, kv_placement="bottom"))  # (no profile pass)
            big = build_model(big_args, ensure_positions(model_config_dict("llama3-8b"), S + GEN + 64), need, 256,
                              S + GEN + 64, not args.no_hip_graph)
            for nb in (128, 256):
                result[f"decode_batch{nb}"] = side_run(
                    args, "llama3-8b", nb, S + GEN // 2, 24, 6,
                    f"llama3-8b decode-only, batch {nb} at context ~{S + GEN // 2} (projections: swl_gemm_packed_wide, "
                    f"csrc/gemm_wide.hip)", model=big)
            del big
            torch.cuda.empty_cache()
        except Exception as exc:     # noqa: BLE001 — a side measurement must never take the bench line down
            print(f"[bench] large-batch side runs failed ({type(exc).__name__}: {exc})", file=sys.stderr)
    if world == 1 and not args.no_extras and args.model == "llama3-8b":
        # BASELINE configs[1] through the persistent one-sequence decode engine (opt-in: tuning decode_engine), in its own
        # process: the pool is sized around the engine's third copy of the layer weights there
        leg = side_run_fresh_process(
            args, "BASELINE.json configs[1] through csrc/decode_engine.hip: ONE persistent launch for the 32 layers of a "
                  "one-sequence step (opt-in: measured slower than the 5-6 launches per layer it replaces, see configs1_batch1)",
            model_name="llama3-8b", batch=1, prompt_len=1024, gen_len=GEN, steps=48, warmup=8,
            extra=("--decode-engine", "--kv-blocks", "8192", "--kv-placement", "bottom"))
        if leg is not None:
            result["configs1_batch1_engine"] = leg
    if world == 1 and not args.no_extras and args.model == "llama3-8b":
        try:        # the reference's own precision on the same batch and contexts (its Triton path is float16-only)
            if args.dtype != "float16":
                f16_args = argparse.Namespace(**dict(vars(args), dtype="float16", kv_blocks=B * (-(-(S + GEN + 64) // 16) + 1) + 64,
                                                     kv_placement="bottom"))
                result["configs2_fp16"] = side_run(f16_args, "llama3-8b", B, S + GEN // 2, 24, 6,
                                                   f"llama3-8b float16 decode-only, batch {B} at context ~{S + GEN // 2}: the reference's "
                                                   f"precision AND rounding points — the exact norm on the fly of r06c: o_proj adds itself into the residual and "
                                                   f"leaves the rows' sums of squares, the SiLU-gate GEMM stages round(r * rstd * w); 6 launches per layer")
        except Exception as exc:     # noqa: BLE001 — a side measurement must never take the bench line down
            print(f"[bench] float16 side run failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        try:
            result["prefill_long"] = prefill_long_leg(args)
        except Exception as exc:     # noqa: BLE001
            print(f"[bench] long-prompt legs failed ({type(exc).__name__}: {exc})", file=sys.stderr)
    if world == 1 and not args.no_extras and not args.no_reference and args.model == "llama3-8b":
        ref = reference_triton_leg(B, first_ctx, K, Wm, S)
        if ref is not None:
            result["reference_triton"] = ref
        if "prefill_long" in result and "4x16384" in result["prefill_long"]:
            ref_long = reference_prefill_leg(4, 16384)
            if ref_long is not None:
                result["prefill_long"]["4x16384"]["reference_triton"] = ref_long
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, B, int(round(mean_ctx)), args.dtype)
    # the gates of the side measurements as flat numbers INSIDE `roofline` (a reader that keeps only the contract keys of
    # this line still sees them); the objects they come from stay where they were
    def frac(*path):
        d = result
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return d if isinstance(d, (int, float)) else None
    summary = {"step_frac": frac("step_roofline", "frac"),
               "paged_attention_us": frac("roofline_paged_attention", "us_per_launch"),
               "prefill_frac": frac("prefill_roofline", "frac"),
               "prefill_attention_frac": frac("prefill_attention_roofline", "frac"),
               "configs1_batch1_step_frac": frac("configs1_batch1", "step_roofline", "frac"),
               "configs1_batch1_ms": frac("configs1_batch1", "ms_per_step"),
               "configs1_batch1_engine_ms": frac("configs1_batch1_engine", "ms_per_step"),
               "configs3_step_frac": frac("configs3_llama2_7b_4x16k", "step_roofline", "frac"),
               "decode_batch128_step_frac": frac("decode_batch128", "step_roofline", "frac"),
               "decode_batch256_step_frac": frac("decode_batch256", "step_roofline", "frac"),
               "configs2_fp16_step_frac": frac("configs2_fp16", "step_roofline", "frac"),
               "prefill_long_4x16384_frac": frac("prefill_long", "4x16384", "roofline", "frac"),
               "prefill_long_4x16384_attention_frac": frac("prefill_long", "4x16384", "attention_kernel", "frac"),
               "prefill_long_1x32768_frac": frac("prefill_long", "1x32768", "roofline", "frac"),
               "reference_decode_tok_s": frac("reference_triton", "value")}
    if isinstance(result.get("roofline"), dict):
        result["roofline"].update({k: v for k, v in summary.items() if v is not None})
    return result


if __name__ == "__main__":
    main()
