#!/usr/bin/env python3
"""mixed_step_probe.py — what a serving step at the knee costs, host against GPU (GPU): P prompt tokens (one fresh request) + Bd
decoding sequences at context C in ONE forward (the scheduler's piggybacked step, eager launches), against the same decodes
alone (hipGraph replay) and the prompt alone. Wall time per forward (host-synchronised) and GPU time (HIP events on the
forward's stream) side by side: wall >> GPU means the step is bound by the Python launch path, not by the kernels."""
import argparse, json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--decodes", default="64,128,192,256")
    ap.add_argument("--prompt", type=int, default=240)
    ap.add_argument("--context", type=int, default=350)
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    cfg = bench.model_config_dict("llama3-8b")
    nmax = max(int(x) for x in a.decodes.split(","))
    ns = argparse.Namespace(batch=nmax + 8, prompt_len=a.prompt, steps=64, warmup=0, dtype="bfloat16", fuse_qkv=True,
                            no_hip_graph=False, skinny_gemm=True, splitk_fusion=True, kv_blocks=(nmax + 8) * 40, kv_placement="bottom")
    model = bench.build_model(ns, cfg, (nmax + 8) * 40, nmax + 8, 640, True, max_tokens=16384)
    vocab = cfg["vocab_size"]
    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(0, vocab, (a.prompt,), generator=g).tolist()

    def timed(fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.record()
        out = fn()
        e.record()
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) * 1e3, s.elapsed_time(e)

    for bd in [int(x) for x in a.decodes.split(",")]:
        dec_ids = list(range(bd))
        pid = bd
        lens = [a.context - 2 * a.reps - 4] * bd
        toks = torch.randint(0, vocab, (bd,), generator=g).tolist()
        rows = {"decode_only": [], "prompt_only": [], "mixed": []}
        for rep in range(a.reps + 1):
            lens = [n + 1 for n in lens]
            toks, w, gms = timed(lambda: model.forward([[x] for x in toks], dec_ids, lens))
            rows["decode_only"].append((w, gms))
            _, w, gms = timed(lambda: model.forward([prompt], [pid], []))
            rows["prompt_only"].append((w, gms))
            model.free_seqs_resources([pid])
            lens = [n + 1 for n in lens]
            out, w, gms = timed(lambda: model.forward([prompt] + [[x] for x in toks], [pid] + dec_ids, lens))
            rows["mixed"].append((w, gms))
            toks = out[1:]
            model.free_seqs_resources([pid])
        model.free_seqs_resources(dec_ids)
        res = {"decodes": bd, "prompt_tokens": a.prompt, "context": a.context}
        for k, v in rows.items():
            res[k + "_wall_ms"] = round(statistics.median(x[0] for x in v[1:]), 3)
            res[k + "_gpu_ms"] = round(statistics.median(x[1] for x in v[1:]), 3)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
