#!/usr/bin/env python3
"""prefill_host_profile.py — cProfile of the eager launch path of a SMALL prompt pass (GPU): where the host's ~20 us per
operator go when the kernels of a 240-token prompt are shorter than the Python that launches them."""
import argparse, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=240)
    ap.add_argument("--decodes", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    cfg = bench.model_config_dict("llama3-8b")
    nb = (a.decodes + 8) * 40
    ns = argparse.Namespace(batch=a.decodes + 8, prompt_len=a.prompt, steps=64, warmup=0, dtype="bfloat16", fuse_qkv=True,
                            no_hip_graph=False, skinny_gemm=True, splitk_fusion=True, kv_blocks=nb, kv_placement="bottom")
    model = bench.build_model(ns, cfg, nb, a.decodes + 8, 640, True, max_tokens=16384)
    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(0, cfg["vocab_size"], (a.prompt,), generator=g).tolist()
    dec_ids = list(range(a.decodes))
    pid = a.decodes
    lens = [300] * a.decodes
    toks = [1] * a.decodes

    def step():
        nonlocal lens, toks
        lens = [n + 1 for n in lens]
        out = model.forward([prompt] + [[t] for t in toks], [pid] + dec_ids, lens)
        toks = out[1:]
        model.free_seqs_resources([pid])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    print(f"wall per forward: {(time.perf_counter() - t0) / a.reps * 1e3:.3f} ms")
    # host-only time: enqueue without waiting (the forward syncs on its tokens, so measure with the profiler instead)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.reps):
        step()
    pr.disable()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(a.top)
        print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
