#!/usr/bin/env python3
"""serve_bench.py — the control plane over the real data plane, on one GPU: R requests (S-token prompts,
G generated tokens) through Engine (scheduler + continuous batching), all at once (--rate 0) or as Poisson
arrivals at --rate req/s. Reports output tokens/s, time-to-first-token and per-output-token latency,
with and without piggybacked decodes (BASELINE.json configs[2] / configs[4], one replica).

    python tools/serve_bench.py [--model llama3-8b] [--requests 256] [--prompt-len 1024] [--gen-len 128]
"""
import argparse, asyncio, json, os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (model construction with random-init weights)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--requests", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--gen-len", type=int, default=128)
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--max-tokens", type=int, default=8192)
    ap.add_argument("--rate", type=float, default=0.0, help="Poisson arrival rate (req/s); 0 = all at once")
    ap.add_argument("--modes", default="plain,piggyback")
    ap.add_argument("--kv-blocks", type=int, default=0, help="KV pool size in blocks (0 = profile_num_blocks at 0.97)")
    ap.add_argument("--passes", type=int, default=1,
                    help="run every mode this many times in the same process: pass 1 pays the hipGraph captures of the batch / "
                         "split-geometry buckets it meets, later passes show the steady state (graph_captures should be 0)")
    return ap.parse_args()


async def run(model, a, piggyback):
    from swiftllm_amd import Engine, RawRequest
    eng = Engine(model.engine_config, model=model, piggyback=piggyback)
    await eng.initialize()
    loops = asyncio.ensure_future(eng.start_all_event_loops())
    rng = random.Random(7)
    vocab = model.model_config.vocab_size
    prompts = [[rng.randrange(vocab) for _ in range(a.prompt_len)] for _ in range(a.requests)]
    ttft, tpot = [], []

    async def one(p, delay):
        await asyncio.sleep(delay)
        t0 = time.perf_counter(); first = None; n = 0
        async for _ in eng.add_request_and_stream(RawRequest("", a.gen_len, p)):
            n += 1
            if first is None:
                first = time.perf_counter()
        t1 = time.perf_counter()
        ttft.append(first - t0)
        if n > 1:
            tpot.append((t1 - first) / (n - 1))
    delays, t = [], 0.0
    for _ in prompts:
        delays.append(t)
        if a.rate > 0:
            t += rng.expovariate(a.rate)
    caps0, fwd0 = getattr(model, "graph_captures", 0), eng.num_forwards
    t0 = time.perf_counter()
    await asyncio.gather(*(one(p, d) for p, d in zip(prompts, delays)))
    dt = time.perf_counter() - t0
    loops.cancel()
    caps = getattr(model, "graph_captures", 0) - caps0
    fwds = max(1, eng.num_forwards - fwd0)
    ttft.sort(); tpot.sort()
    return {"piggyback": piggyback, "requests": a.requests, "prompt_len": a.prompt_len, "gen_len": a.gen_len,
            "rate_req_s": a.rate, "wall_s": round(dt, 3), "output_tok_s": round(a.requests * a.gen_len / dt, 1),
            "total_tok_s": round(a.requests * (a.gen_len + a.prompt_len) / dt, 1), "forwards": eng.num_forwards,
            # every capture = one eager warm-up forward + one capture (worker/model.py: _forward_decode_graph); keyed on batch
            # BUCKETS and split-geometry buckets since r05
            "graph_captures": caps, "graph_captures_per_1000_forwards": round(1000.0 * caps / fwds, 2),
            "graphs_cached": len(getattr(model, "_decode_graphs", {}) or {}),
            "ttft_ms_p50": round(ttft[len(ttft) // 2] * 1e3, 1), "ttft_ms_max": round(ttft[-1] * 1e3, 1),
            "tpot_ms_p50": round(tpot[len(tpot) // 2] * 1e3, 2) if tpot else None,
            "tpot_ms_p99": round(tpot[int(len(tpot) * 0.99)] * 1e3, 2) if tpot else None}


def main():
    a = parse()
    cfg = bench.model_config_dict(a.model)
    # the KV pool is what the product's own sizing gives (profile_num_blocks at gpu_mem_utilization 0.97: ~125 k blocks for
    # Llama-3-8B on a 288 GB MI355X) unless --kv-blocks says otherwise; no filler sequences: the engine hands out sequence ids
    ns = argparse.Namespace(batch=2 * a.max_batch, prompt_len=a.prompt_len, steps=a.gen_len, warmup=0, dtype=a.dtype,
                            fuse_qkv=True, no_hip_graph=False, skinny_gemm=True, splitk_fusion=True,
                            kv_blocks=a.kv_blocks, kv_placement="bottom")
    blocks_per_seq = (a.prompt_len + a.gen_len + 16) // 16
    model = bench.build_model(ns, cfg, int(a.max_batch * blocks_per_seq * 1.1) + 8, 2 * a.max_batch,
                              a.prompt_len + a.gen_len + 16, True)
    # scheduler limits (the block table was built for 2 x max_batch ids: running + swapped-out requests)
    model.engine_config.max_tokens_in_batch = a.max_tokens
    model.engine_config.max_batch_size = a.max_batch
    for mode in a.modes.split(","):
        for p in range(a.passes):
            res = asyncio.run(run(model, a, mode == "piggyback"))
            res["pass"] = p + 1
            res["model"] = a.model
            res["max_batch"] = a.max_batch
            res["kv_pool_blocks"] = int(model.num_blocks)
            res["kv_pool_gb"] = round(2 * model.k_cache.numel() * model.k_cache.element_size() / 1e9, 1)
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
