#!/usr/bin/env python3
"""serve_bench.py — the control plane over the real data plane, on one GPU: R requests (S-token prompts,
G generated tokens) through Engine (scheduler + continuous batching), all at once (--rate 0) or as Poisson
arrivals at --rate req/s. Reports output tokens/s, time-to-first-token and per-output-token latency,
with and without piggybacked decodes (BASELINE.json configs[2] / configs[4], one replica).

    python tools/serve_bench.py [--model llama3-8b] [--requests 256] [--prompt-len 1024] [--gen-len 128]

--sweep r1,r2,... : the online benchmark the reference publishes (README.md:105-113: prompts sampled from ShareGPT, Poisson
arrivals at increasing rates, latency against the rate up to saturation) on one replica: every rate runs `--sweep-seconds`
of Poisson arrivals through the SAME engine and pool, lengths drawn from `--lengths` (fixed | sharegpt = a synthetic
stand-in for the dataset, which is not on the box: log-normal prompt / output lengths, medians ~150 / ~150 tokens, clipped
to [4, 2048] / [4, 1024]); one JSON line per rate plus a last line naming the knee (the first rate the engine no longer
keeps up with: mean normalised latency — end-to-end seconds per output token — above 2 x its low-load value).
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
    ap.add_argument("--cpu-blocks", type=int, default=0, help="host swap pool in blocks (with a small --kv-blocks: swapping under load)")
    ap.add_argument("--sweep", default="", help="comma-separated Poisson rates (req/s): the online sweep to the knee")
    ap.add_argument("--sweep-seconds", type=float, default=10.0, help="seconds of arrivals per rate of the sweep")
    ap.add_argument("--lengths", default="fixed", choices=["fixed", "sharegpt"],
                    help="fixed: --prompt-len / --gen-len for every request; sharegpt: synthetic log-normal lengths")
    ap.add_argument("--passes", type=int, default=1,
                    help="run every mode this many times in the same process: pass 1 pays the hipGraph captures of the batch / "
                         "split-geometry buckets it meets, later passes show the steady state (graph_captures should be 0)")
    return ap.parse_args()


def draw_lengths(a, rng, n):
    """(prompt_len, gen_len) per request. sharegpt: a synthetic stand-in for the reference's dataset (README.md:107) —
    log-normal lengths (mu 5.0, sigma 1.0 / 0.9), clipped; the medians are ~150 tokens each, the means ~240 / ~215."""
    if a.lengths == "fixed":
        return [(a.prompt_len, a.gen_len)] * n
    out = []
    for _ in range(n):
        pl = int(min(2048, max(4, rng.lognormvariate(5.0, 1.0))))
        gl = int(min(1024, max(4, rng.lognormvariate(5.0, 0.9))))
        out.append((pl, gl))
    return out


async def run(model, a, piggyback, eng=None, seed=7):
    from swiftllm_amd import Engine, RawRequest
    own = eng is None
    if own:
        eng = Engine(model.engine_config, model=model, piggyback=piggyback)
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())
    rng = random.Random(seed)
    vocab = model.model_config.vocab_size
    lens = draw_lengths(a, rng, a.requests)
    prompts = [[rng.randrange(vocab) for _ in range(pl)] for pl, _ in lens]
    ttft, tpot, norm = [], [], []

    async def one(p, gl, delay):
        await asyncio.sleep(delay)
        t0 = time.perf_counter(); first = None; n = 0
        async for _ in eng.add_request_and_stream(RawRequest("", gl, p)):
            n += 1
            if first is None:
                first = time.perf_counter()
        t1 = time.perf_counter()
        if first is None:           # the engine ended the stream without a token (its model thread died: see stderr)
            raise RuntimeError("a request came back without tokens")
        ttft.append(first - t0)
        norm.append((t1 - t0) / max(1, n))      # end-to-end latency per output token of this request
        if n > 1:
            tpot.append((t1 - first) / (n - 1))
    delays, t = [], 0.0
    for _ in prompts:
        delays.append(t)
        if a.rate > 0:
            t += rng.expovariate(a.rate)
    caps0, fwd0, cap_s0 = getattr(model, "graph_captures", 0), eng.num_forwards, getattr(model, "graph_capture_s", 0.0)
    so0, si0 = eng.num_swapped_out, eng.num_swapped_in
    t0 = time.perf_counter()
    await asyncio.gather(*(one(p, gl, d) for p, (_, gl), d in zip(prompts, lens, delays)))
    dt = time.perf_counter() - t0
    if own:
        loops.cancel()
    caps = getattr(model, "graph_captures", 0) - caps0
    fwds = max(1, eng.num_forwards - fwd0)
    ttft.sort(); tpot.sort(); norm.sort()
    out_toks, in_toks = sum(gl for _, gl in lens), sum(pl for pl, _ in lens)
    return {"piggyback": piggyback, "requests": a.requests, "prompt_len": a.prompt_len, "gen_len": a.gen_len,
            "lengths": a.lengths, "mean_prompt_len": round(in_toks / a.requests, 1), "mean_gen_len": round(out_toks / a.requests, 1),
            "rate_req_s": a.rate, "wall_s": round(dt, 3), "completed_req_s": round(a.requests / dt, 2),
            "arrival_span_s": round(delays[-1], 3),
            "output_tok_s": round(out_toks / dt, 1),
            "total_tok_s": round((out_toks + in_toks) / dt, 1), "forwards": eng.num_forwards - fwd0,
            "norm_latency_ms_per_tok_mean": round(sum(norm) / len(norm) * 1e3, 2),
            "norm_latency_ms_per_tok_p99": round(norm[int(len(norm) * 0.99)] * 1e3, 2),
            "ttft_ms_p99": round(ttft[int(len(ttft) * 0.99)] * 1e3, 1),
            # every capture = one eager warm-up forward + one capture (worker/model.py: _forward_decode_graph); keyed on batch
            # BUCKETS and split-geometry buckets since r05
            "seqs_swapped_out": eng.num_swapped_out - so0, "seqs_swapped_in": eng.num_swapped_in - si0,
            "graph_captures": caps, "graph_captures_per_1000_forwards": round(1000.0 * caps / fwds, 2),
            "graph_capture_s": round(getattr(model, "graph_capture_s", 0.0) - cap_s0, 3),
            "graphs_cached": len(getattr(model, "_decode_graphs", {}) or {}),
            "ttft_ms_p50": round(ttft[len(ttft) // 2] * 1e3, 1), "ttft_ms_max": round(ttft[-1] * 1e3, 1),
            "tpot_ms_p50": round(tpot[len(tpot) // 2] * 1e3, 2) if tpot else None,
            "tpot_ms_p99": round(tpot[int(len(tpot) * 0.99)] * 1e3, 2) if tpot else None}


def sweep(model, a):
    """The online sweep: one engine, one pool, rates in ascending order; each rate = --sweep-seconds of Poisson arrivals and
    the drain of what they left. A first short pass at a low rate pays the hipGraph captures of the batch buckets (and is
    printed as `warmup`)."""
    from swiftllm_amd import Engine
    rates = [float(r) for r in a.sweep.split(",")]
    piggyback = "piggyback" in a.modes.split(",")

    async def go():
        eng = Engine(model.engine_config, model=model, piggyback=piggyback)
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())
        rows = []
        for i, rate in enumerate([rates[len(rates) // 2]] + rates):
            b = argparse.Namespace(**vars(a))
            b.rate = rate
            b.requests = max(32, int(rate * (a.sweep_seconds if i else a.sweep_seconds / 2)))
            res = await run(model, b, piggyback, eng=eng, seed=11 + i)
            res.update(model=a.model, max_batch=a.max_batch, kv_pool_blocks=int(model.num_blocks), phase="warmup" if i == 0 else "sweep")
            print(json.dumps(res), flush=True)
            if i:
                rows.append(res)
        loops.cancel()
        return rows
    rows = asyncio.run(go())
    base = rows[0]["norm_latency_ms_per_tok_mean"]
    knee = next((r["rate_req_s"] for r in rows if r["norm_latency_ms_per_tok_mean"] > 2.0 * base), None)
    best = max(rows, key=lambda r: r["output_tok_s"])
    print(json.dumps({"summary": "online sweep", "model": a.model, "lengths": a.lengths, "piggyback": piggyback,
                      "max_batch": a.max_batch, "rates": rates, "knee_rate_req_s": knee,
                      "low_load_norm_latency_ms_per_tok": base,
                      "norm_latency_ms_per_tok_by_rate": {str(r["rate_req_s"]): r["norm_latency_ms_per_tok_mean"] for r in rows},
                      "drain_s_by_rate": {str(r["rate_req_s"]): round(r["wall_s"] - r["arrival_span_s"], 2) for r in rows},
                      "peak_output_tok_s": best["output_tok_s"], "peak_at_rate": best["rate_req_s"],
                      "knee_rule": "first rate whose mean normalised latency (end-to-end seconds per output token, averaged "
                                   "over requests) exceeds 2 x the lowest rate's; None = not reached in this sweep"}), flush=True)


def main():
    a = parse()
    if a.sweep and a.lengths == "sharegpt":
        a.prompt_len, a.gen_len = 2048, 1024        # the pool / block-table sizing below uses the clip limits
    cfg = bench.model_config_dict(a.model)
    # the KV pool is what the product's own sizing gives (profile_num_blocks at gpu_mem_utilization 0.97: ~125 k blocks for
    # Llama-3-8B on a 288 GB MI355X) unless --kv-blocks says otherwise; no filler sequences: the engine hands out sequence ids
    ns = argparse.Namespace(batch=2 * a.max_batch, prompt_len=a.prompt_len, steps=a.gen_len, warmup=0, dtype=a.dtype,
                            fuse_qkv=True, no_hip_graph=False, skinny_gemm=True, splitk_fusion=True,
                            kv_blocks=a.kv_blocks, kv_placement="bottom", num_cpu_blocks=a.cpu_blocks)
    blocks_per_seq = (a.prompt_len + a.gen_len + 16) // 16
    min_blocks = int(a.max_batch * blocks_per_seq * 1.1) + 8 if not a.kv_blocks else min(a.kv_blocks, blocks_per_seq + 8)
    model = bench.build_model(ns, cfg, min_blocks, 2 * a.max_batch,
                              a.prompt_len + a.gen_len + 16, True)
    # scheduler limits (the block table was built for 2 x max_batch ids: running + swapped-out requests)
    model.engine_config.max_tokens_in_batch = a.max_tokens
    model.engine_config.max_batch_size = a.max_batch
    if a.sweep:
        return sweep(model, a)
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
